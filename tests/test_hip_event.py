"""-m gpu: the tracker's camera iteration with the event term (TrackerIteration.optimize_cam_in_batch: pose -> rays ->
HIP render of the rescaled image and of the random pixel batch -> U-Net -> event + RGB-D losses -> pose gradient)
against tests/golden/tiny_event_iter.npz (the reference's own statements, functions and network)."""
import types

import numpy as np
import pytest
import torch

from tests.util import load, rel_err

pytestmark = pytest.mark.gpu


class _RecordingOptimizer:
    """stands in for the camera Adam: keeps the gradient the iteration hands to step()"""

    def __init__(self, p):
        self.p, self.grad, self.steps = p, None, 0

    def zero_grad(self):
        self.p.grad = None

    def step(self):
        self.grad = None if self.p.grad is None else self.p.grad.detach().clone()
        self.steps += 1


def _setup(fx, handle_dynamic=True, blur=True, activate=True):
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, cfg_like, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    for p in model.parameters():
        p.requires_grad_(False)
    H, W, fxx, fy, cx, cy = [float(x) for x in fx['cam']]
    cfg = cfg_like()
    cfg['tracking'] = {'device': DEV, 'w_color_loss': float(fx['w_color_loss']), 'ignore_edge_W': int(fx['edge'][1]),
                       'ignore_edge_H': int(fx['edge'][0]), 'handle_dynamic': handle_dynamic, 'use_color_in_tracking': True}
    cfg['event'] = {'activate_events': activate, 'blur': blur, 'kernel_sizes': [int(k) for k in fx['kernel_sizes']],
                    'kernel_weights': [float(k) for k in fx['kernel_weights']],
                    'unblurred_weight': float(fx['unblurred_weight']), 'balancer': float(fx['balancer'])}
    torch.manual_seed(int(fx['unet_seed']))
    net = E.event.UNet_2heads(6, 2, 2)                 # built on the CPU under the fixture's seed, then moved
    for p in net.parameters():
        p.requires_grad_(False)
    net = net.to(DEV).eval()
    slam = types.SimpleNamespace(nice=True, bound=bound, renderer=renderer, event_net=net, H=int(H), W=int(W), fx=fxx,
                                 fy=fy, cx=cx, cy=cy, low_gpu_mem=False)
    trk = E.tracker.TrackerIteration(cfg, None, slam)
    trk.c, trk.decoders = grids, model
    img = {k: torch.from_numpy(fx[k]).to(DEV) for k in ('gt_depth', 'gt_color', 'pre_gt_color', 'gt_event', 'gt_mask')}
    return trk, img, DEV


def test_event_iteration_matches_reference_fixture(monkeypatch):
    fx = load("tiny_event_iter")
    trk, img, DEV = _setup(fx)
    idx = torch.from_numpy(fx['idx']).to(DEV)
    monkeypatch.setattr(torch, 'randint', lambda *a, **k: idx)        # the reference's pixel draw (RNG streams differ)
    ct = torch.from_numpy(fx['camera_tensor']).to(DEV).requires_grad_(True)
    opt = _RecordingOptimizer(ct)
    ret = trk.optimize_cam_in_batch(ct, None, img['gt_color'], img['gt_depth'], img['gt_event'], img['gt_mask'],
                                    int(fx['batch_size']), opt, 0, 0, img['pre_gt_color'], rgbd=True, event=True,
                                    scale_factor=float(fx['scale_factor']))
    assert len(ret) == 10 and opt.steps == 1
    loss_rgbd, loss_event, loss_mask, gt_event, full_event, gts, preds, terms, gt_mask, p_event = ret
    assert np.array_equal(gt_event.cpu().numpy(), fx['gt_event_s'])
    assert np.array_equal(gt_mask.cpu().numpy(), fx['gt_mask_s'])
    assert rel_err(full_event.detach().cpu().numpy(), fx['full_event']) <= 1e-4
    assert rel_err(p_event.detach().cpu().numpy(), fx['event_prob'][0, 1]) <= 1e-4
    assert rel_err(preds[0].detach().cpu().numpy(), fx['pred_event_blur']) <= 1e-4
    assert rel_err(gts[0].cpu().numpy(), fx['gt_event_blur']) <= 1e-5
    assert abs(loss_rgbd - float(fx['loss_rgbd'])) <= 1e-4 * abs(float(fx['loss_rgbd']))
    assert abs(loss_event - float(fx['loss_event'])) <= 1e-4 * abs(float(fx['loss_event']))
    assert abs(loss_mask - float(fx['loss_mask'])) <= 1e-4 * abs(float(fx['loss_mask']))
    assert abs(terms[1] - float(fx['loss_terms'][1])) <= 1e-4 * float(fx['loss_terms'][1])
    assert rel_err(opt.grad.cpu().numpy(), fx['g_total']) <= 1e-3
    assert ct.grad is None                                            # the iteration ends with zero_grad (:236)

    # event term alone (rgbd=False): its own pose gradient, 100x smaller than the RGB-D one
    opt2 = _RecordingOptimizer(ct)
    ret = trk.optimize_cam_in_batch(ct, None, img['gt_color'], img['gt_depth'], img['gt_event'], img['gt_mask'],
                                    int(fx['batch_size']), opt2, 0, 0, img['pre_gt_color'], rgbd=False, event=True,
                                    scale_factor=float(fx['scale_factor']))
    assert ret[0] is None and abs(ret[1] - float(fx['loss_event'])) <= 1e-4 * abs(float(fx['loss_event']))
    assert rel_err(opt2.grad.cpu().numpy(), fx['g_event']) <= 1e-3

    # RGB-D alone with the dynamic-object mask (:180-182)
    opt3 = _RecordingOptimizer(ct)
    ret = trk.optimize_cam_in_batch(ct, None, img['gt_color'], img['gt_depth'], None, None, int(fx['batch_size']), opt3,
                                    0, 0, None, rgbd=True, event=False)
    assert ret[1] is None and ret[2] is None and ret[4] is None
    assert abs(ret[0] - float(fx['loss_rgbd'])) <= 1e-4 * abs(float(fx['loss_rgbd']))
    assert rel_err(opt3.grad.cpu().numpy(), fx['g_rgbd']) <= 1e-3


def test_event_iteration_switches(monkeypatch):
    """activate_events off: the event loss is reported but not back-propagated (:231); blur off: 7-tuple."""
    fx = load("tiny_event_iter")
    trk, img, DEV = _setup(fx, blur=False, activate=False)
    idx = torch.from_numpy(fx['idx']).to(DEV)
    monkeypatch.setattr(torch, 'randint', lambda *a, **k: idx)
    ct = torch.from_numpy(fx['camera_tensor']).to(DEV).requires_grad_(True)
    opt = _RecordingOptimizer(ct)
    ret = trk.optimize_cam_in_batch(ct, None, img['gt_color'], img['gt_depth'], img['gt_event'], img['gt_mask'],
                                    int(fx['batch_size']), opt, 0, 0, img['pre_gt_color'], rgbd=True, event=True,
                                    scale_factor=float(fx['scale_factor']))
    assert len(ret) == 7
    l2 = float(((fx['gt_event_s'].astype(np.float64) - fx['full_event']) ** 2).sum()) * float(fx['balancer'])
    assert abs(ret[1] - l2) <= 1e-4 * l2
    assert rel_err(opt.grad.cpu().numpy(), fx['g_rgbd']) <= 1e-3
    trk.event_net = None
    with pytest.raises(RuntimeError):
        trk.optimize_cam_in_batch(ct, None, img['gt_color'], img['gt_depth'], img['gt_event'], img['gt_mask'],
                                  int(fx['batch_size']), opt, 0, 0, img['pre_gt_color'], rgbd=True, event=True)


def test_static_shape_and_graphed_iteration_match_fixture(monkeypatch):
    """The capture-friendly formulation (no boolean indexing: dropped rays rendered without loss, sampler maxima over
    the kept rays, masked median) and its hipGraph replay reproduce the reference fixture."""
    import gc
    from evennicer_slam_amd.mapper import FusedAdam
    fx = load("tiny_event_iter")
    trk, img, DEV = _setup(fx)
    idx = torch.from_numpy(fx['idx']).to(DEV)
    monkeypatch.setattr(torch, 'randint', lambda *a, **k: idx)
    sf = float(fx['scale_factor'])
    ct = torch.from_numpy(fx['camera_tensor']).to(DEV).requires_grad_(True)
    frame = trk.prepare_event_frame(img['gt_event'], img['gt_mask'], img['pre_gt_color'], sf)
    o = trk.iteration_losses(ct, img['gt_color'], img['gt_depth'], frame, int(fx['batch_size']), True, True, sf,
                             static_shapes=True)
    assert abs(o['rgbd'].item() - float(fx['loss_rgbd'])) <= 1e-4 * abs(float(fx['loss_rgbd']))
    assert abs(o['event'].item() - float(fx['loss_event'])) <= 1e-4 * abs(float(fx['loss_event']))
    o['total'].backward()
    assert rel_err(ct.grad.cpu().numpy(), fx['g_total']) <= 1e-3
    del o
    ct.grad = None
    gc.collect()
    opt = FusedAdam([ct], lr=0.0)                       # lr 0: the replays leave the camera where the fixture has it
    git = E_tracker().GraphedCameraIteration(trk, ct, opt, img['gt_color'], img['gt_depth'], img['gt_event'], img['gt_mask'],
                                             img['pre_gt_color'], batch_size=int(fx['batch_size']), rgbd=True, event=True,
                                             scale_factor=sf)
    for _ in range(2):
        git.set_frame(img['gt_color'], img['gt_depth'], img['gt_event'], img['gt_mask'], img['pre_gt_color'])
        l_rgbd, l_event, l_mask = git.step()
        assert abs(l_rgbd.item() - float(fx['loss_rgbd'])) <= 1e-4 * abs(float(fx['loss_rgbd']))
        assert abs(l_event.item() - float(fx['loss_event'])) <= 1e-4 * abs(float(fx['loss_event']))
        assert abs(l_mask.item() - float(fx['loss_mask'])) <= 1e-4 * abs(float(fx['loss_mask']))
        assert rel_err(ct.grad.cpu().numpy(), fx['g_total']) <= 1e-3
    assert np.array_equal(ct.detach().cpu().numpy(), fx['camera_tensor'])

    # the map changes in place (Tracker.update_para_from_mapping): the captured iteration follows after refresh_map()
    with torch.no_grad():
        trk.c['grid_color'].mul_(1.25)
        trk.c['grid_fine'].add_(0.01)
        trk.decoders.color_decoder.output_linear.weight.mul_(0.9)
    stale = [t.item() for t in git.step()]
    assert git.refresh_map() >= 3
    l_rgbd, l_event, l_mask = git.step()
    o = trk.iteration_losses(ct, img['gt_color'], img['gt_depth'], frame, int(fx['batch_size']), True, True, sf,
                             static_shapes=True)
    assert abs(l_rgbd.item() - o['rgbd'].item()) <= 1e-6 * abs(o['rgbd'].item())
    assert abs(l_event.item() - o['event'].item()) <= 1e-5 * abs(o['event'].item())
    assert abs(stale[0] - o['rgbd'].item()) > 1e-4 * abs(o['rgbd'].item())      # (before the refresh it rendered the old map)
    del o

    # the reference's protocol REPLACES the map per frame (Tracker.py:247-259: `self.c[key] = val.clone()`,
    # `self.decoders = copy.deepcopy(shared_decoders)`): the graph owns its map buffers and copies the new map into them
    import copy
    captured = {k: trk.c[k] for k in ('grid_middle', 'grid_fine', 'grid_color')}
    captured_dec = trk.decoders
    with torch.no_grad():
        shared_c = {k: v.clone() for k, v in trk.c.items()}
        shared_c['grid_middle'].mul_(0.8)
        shared_c['grid_color'].add_(0.02)
        shared_dec = copy.deepcopy(trk.decoders)
        shared_dec.fine_decoder.pts_linears[1].weight.mul_(1.1)
    trk.c = {k: v.clone() for k, v in shared_c.items()}             # update_para_from_mapping
    trk.decoders = copy.deepcopy(shared_dec)
    assert git.refresh_map() >= 3
    assert all(trk.c[k] is captured[k] for k in captured) and trk.decoders is captured_dec
    assert torch.equal(trk.c['grid_middle'], shared_c['grid_middle'])
    l_rgbd, l_event, l_mask = git.step()
    # ground truth: an eager iteration on a FRESH tracker map holding the same values
    trk2_c, trk2_dec = trk.c, trk.decoders
    trk.c = {k: v.clone() for k, v in shared_c.items()}
    trk.decoders = copy.deepcopy(shared_dec)
    o = trk.iteration_losses(ct, img['gt_color'], img['gt_depth'], frame, int(fx['batch_size']), True, True, sf,
                             static_shapes=True)
    trk.c, trk.decoders = trk2_c, trk2_dec
    assert abs(l_rgbd.item() - o['rgbd'].item()) <= 1e-6 * abs(o['rgbd'].item())
    assert abs(l_event.item() - o['event'].item()) <= 1e-5 * abs(o['event'].item())
    # explicit arguments work as well, and a wrong shape is refused
    assert git.refresh_map(shared_c, shared_dec) >= 0
    bad = dict(shared_c, grid_color=shared_c['grid_color'][:, :, :-1].contiguous())
    with pytest.raises(Exception):
        git.refresh_map(bad, shared_dec)


def E_tracker():
    import evennicer_slam_amd as E
    return E.tracker


def test_render_img_rescale_full_size_matches_reference():
    """BASELINE config 3's cost centre at FULL size: Renderer.render_img_rescale on room0, 102 x 180 = 18 360 rays x 48,
    scale 0.15, against the reference's own render_img_rescale (tests/golden/room0_rescale_18360.npz): colour / depth /
    uncertainty images to 1e-4 relative, the gradient of a seeded functional of colour and depth with respect to c2w to 1e-3,
    and the tracker's fused pose -> ray formulation (TrackerIteration._render_rescaled) against the same colour image."""
    import types
    import bench
    import evennicer_slam_amd as E
    from tests.util import load, rel_err, within_rel
    DEV = 'cuda:0'
    g = load('room0_rescale_18360')
    sc = bench.build_scene_cpu('room0', seed=0)
    assert np.allclose([float(sc['grids'][k].double().sum()) for k in sc['grids']], g['grid_checksum'], rtol=0, atol=1e-9)
    assert abs(float(sc['depth_img'].double().sum()) - float(g['depth_img_checksum'])) < 1e-6
    model = sc['model'].cuda()
    bench.attach_bounds(model, sc['bound'])
    for q in model.parameters():
        q.requires_grad_(False)
    grids = {k: v.cuda() for k, v in sc['grids'].items()}
    renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
    depth_img = sc['depth_img'].cuda()
    c2w = torch.from_numpy(g['c2w']).cuda().requires_grad_(True)
    sf = float(g['scale_factor'])
    depth, unc, color = renderer.render_img_rescale(grids, model, c2w, DEV, 'color', gt_depth=depth_img, scale_factor=sf)
    assert tuple(color.shape) == (102, 180, 3) and depth.dtype == torch.float64
    # floor: entries below 10 % of the image's largest magnitude are compared against that 10 % (the colour of random-init
    # decoders is a signed float32 sum over 48 samples that cancels to near zero in places: its absolute rounding error is
    # ~5e-6 of the largest value on either implementation)
    for name, got in (('color', color), ('depth', depth), ('uncertainty', unc)):
        # (uncertainty = second moment about the rendered depth: not part of north_star's 1e-4 bar, amplifies the float32
        # rounding of the resized depth image; 3e-4 above a 1 % floor)
        ok, worst = within_rel(got.detach().cpu().numpy(), g[name], rel=3e-4 if name == 'uncertainty' else 1e-4,
                               floor={'color': 0.1, 'depth': 1e-3}.get(name, 1e-2))
        assert ok, (name, worst)
    loss = (color * torch.from_numpy(g['w_color']).cuda()).sum().double() + (depth * torch.from_numpy(g['w_depth']).cuda()).sum()
    assert abs(float(loss) - float(g['loss'])) < 1e-4 * max(abs(float(g['loss'])), 1.0)
    loss.backward()
    assert rel_err(c2w.grad.cpu().numpy(), g['g_c2w']) < 1e-3
    # the tracker's formulation: camera tensor (quaternion + translation) -> rays in one fused launch
    cfg = dict(sc['cfg'])
    cfg['tracking'] = {'device': DEV, 'w_color_loss': 0.5, 'ignore_edge_W': 100, 'ignore_edge_H': 100, 'handle_dynamic': True,
                       'use_color_in_tracking': True}
    cfg['event'] = {'activate_events': False, 'blur': False, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                    'balancer': 0.025}
    slam = types.SimpleNamespace(nice=True, bound=sc['bound'], renderer=renderer, event_net=None, low_gpu_mem=False, **bench.CAM)
    trk = E.tracker.TrackerIteration(cfg, None, slam)
    trk.c, trk.decoders = grids, model
    ct = E.common.get_tensor_from_camera(torch.from_numpy(g['c2w'])).to(DEV).float().requires_grad_(True)
    col2 = trk._render_rescaled(ct, depth_img, sf)
    ok, worst = within_rel(col2.detach().cpu().numpy(), g['color'], rel=1e-4, floor=0.1)
    assert ok, worst
    (col2 * torch.from_numpy(g['w_color']).cuda()).sum().backward()
    assert ct.grad is not None and bool(torch.isfinite(ct.grad).all()) and float(ct.grad.abs().max()) > 0
