"""-m gpu: the mapper glue on the device layout (mapper.MaskedGridOptimizer + functional.VoxelMajorGrid) against
the fixture produced by the reference's own optimize_map statements (tests/golden/tiny_mapper_iters.npz)."""
import numpy as np
import pytest
import torch

from tests.util import load

pytestmark = pytest.mark.gpu

KEYS = ('grid_middle', 'grid_fine', 'grid_color')
STAGE_LR = {'middle': (0.0, 0.1, 0.0, 0.0), 'fine': (0.0, 0.005, 0.005, 0.0), 'color': (0.005, 0.005, 0.005, 0.005)}


def _run(graph_free=True, layout='contiguous'):
    import evennicer_slam_amd as E
    from evennicer_slam_amd.mapper import MaskedGridOptimizer
    from tests.hip_util import DEV, as_layout, tiny_on_gpu
    g = load("tiny_mapper_iters")
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    grids = {k: as_layout(v, layout) for k, v in grids.items()}
    masks = {k: torch.from_numpy(g['mask_' + k]).to(DEV) for k in KEYS}
    before = {k: grids[k].clone() for k in KEYS}
    opt = MaskedGridOptimizer(grids, masks, keys=KEYS)
    from evennicer_slam_amd.mapper import FusedAdam
    dec_params = list(model.color_decoder.parameters())
    dec_opt = FusedAdam(dec_params, lr=0.0)
    lrf = float(g['lr_factor'])
    losses = []
    for it, stage in enumerate(g['stages']):
        stage = str(stage)
        lr = STAGE_LR[stage]
        dec_opt.set_lr(lr[0] * lrf)
        dec_opt.zero_grad()
        depth, var, color = renderer.render_batch_ray(opt.render_grids(), model, rays['rays_d'], rays['rays_o'], DEV,
                                                      stage, gt_depth=rays['gt_depth'])
        loss = E.losses.rgbd_loss(depth, color if stage == 'color' else None, rays['gt_depth'], rays['gt_color'],
                                  float(g['w_color_loss']))
        loss.backward()
        dec_opt.step()
        opt.step({'grid_middle': lr[1] * lrf, 'grid_fine': lr[2] * lrf, 'grid_color': lr[3] * lrf})
        losses.append(float(loss.item()))
        if it in (0, 4, 6):
            ref = g[f'grid_middle_after_{it}']
            got = opt.grid('grid_middle').cpu().numpy()
            assert np.abs(got - ref).max() <= 5e-5, it
    opt.write_back()
    return g, s, grids, before, model, losses, opt


@pytest.mark.parametrize("layout", ['contiguous', 'channels_last_3d'])
def test_mapper_iterations_match_reference_fixture(layout):
    """layout channels_last_3d: the optimiser works on the caller's tensors themselves (their storage is the device layout)"""
    g, s, grids, before, model, losses, opt = _run(layout=layout)
    assert opt.native == (set(KEYS) if layout == 'channels_last_3d' else set())
    assert np.abs(np.array(losses) - g['losses']).max() <= 1e-4 * np.abs(g['losses']).max()
    for k in KEYS:
        ref = g['final_' + k]
        got = grids[k].cpu().numpy()
        # an Adam update is at most lr per step whatever the gradient; 10 steps at lr <= 0.1
        assert np.abs(got - ref).max() <= 5e-5, k
        unmasked = ~g['mask_' + k]
        assert np.array_equal(got[0, :, unmasked], before[k].cpu().numpy()[0, :, unmasked]), k     # bit-identical
        moved = np.abs(ref - s[k]).max()
        assert moved > 1e-2, k                                                        # the test is not vacuous
    for name, ref in g.items():
        if name.startswith('final_cd_'):
            got = dict(model.color_decoder.named_parameters())[name[len('final_cd_'):]].detach().cpu().numpy()
            assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), name
    # the accumulators are clear again, the step counters follow torch.optim.Adam's (grids join with their stage)
    for k in KEYS:
        assert float(opt.grids[k].grad_vm.abs().max()) == 0.0
    assert opt.steps == [10, 5, 3]
    assert opt.step_t.tolist() == [10, 5, 3]


def test_voxel_major_grid_gradients_equal_dense_path():
    """Same render call through a dense [1,32,D,H,W] leaf and through VoxelMajorGrid: identical gradients."""
    import evennicer_slam_amd as E
    from evennicer_slam_amd.mapper import MaskedGridOptimizer
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    dense = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    d, v, c = renderer.render_batch_ray(dense, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
    (d.sum() + c.sum()).backward()
    opt = MaskedGridOptimizer(grids, None, keys=KEYS)
    d2, v2, c2 = renderer.render_batch_ray(opt.render_grids(), model, rays['rays_d'], rays['rays_o'], DEV, 'color',
                                           gt_depth=rays['gt_depth'])
    assert torch.equal(d, d2) and torch.equal(c, c2)
    (d2.sum() + c2.sum()).backward()
    for k in KEYS:
        G = opt.grids[k]
        D, H, W = G.dims
        got = G.grad_vm.reshape(D, H, W, 32).permute(3, 0, 1, 2)[None]
        ref = dense[k].grad
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), k
        assert G.has_grad


def test_fused_adam_matches_torch_adam():
    from evennicer_slam_amd.mapper import FusedAdam
    from tests.hip_util import DEV
    g = torch.Generator().manual_seed(0)
    shapes = [(32, 93), (32,), (4, 32), (1,), (32, 125), (3, 93)]
    a = [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    ref = torch.optim.Adam([{'params': b, 'lr': 0.0}])
    opt = FusedAdam(a, lr=0.0)
    for it, lr in enumerate([0.0, 0.005, 0.005, 0.1, 0.1, 0.005, 0.0, 0.005]):
        grads = [torch.randn(s, generator=g).to(DEV) * (10.0 ** ((it % 4) - 2)) for s in shapes]
        grads[1][:4] = 0.0                                   # exact zeros: update must be exactly 0 at step 1
        for t, u, gr in zip(a, b, grads):
            t.grad, u.grad = gr.clone(), gr.clone()
        ref.param_groups[0]['lr'] = lr
        ref.step()
        opt.step(lr)
    assert opt.step_t.item() == 8
    for t, u in zip(a, b):
        assert float((t - u).abs().max()) <= 2e-6 * max(1.0, float(u.abs().max()))


def test_render_with_fused_loss_matches_separate_loss():
    """Renderer.render_batch_ray_rgbd_loss (loss folded into the compositing launches) against render_batch_ray +
    losses.rgbd_loss: outputs, loss and every gradient, for the three depth-guided stages."""
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    names = ['grid_middle', 'grid_fine', 'grid_color']
    for stage in ('middle', 'fine', 'color'):
        res = []
        for fused in (False, True):
            g = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
            ro = rays['rays_o'].detach().clone().requires_grad_(True)
            rd = rays['rays_d'].detach().clone().requires_grad_(True)
            for p in model.parameters():
                p.grad = None
            if fused:
                loss, depth, var, color = renderer.render_batch_ray_rgbd_loss(g, model, rd, ro, DEV, stage, rays['gt_depth'],
                                                                              rays['gt_color'], 0.2)
                assert not depth.requires_grad and not color.requires_grad
            else:
                depth, var, color = renderer.render_batch_ray(g, model, rd, ro, DEV, stage, gt_depth=rays['gt_depth'])
                loss = E.losses.rgbd_loss(depth, color if stage == 'color' else None, rays['gt_depth'], rays['gt_color'], 0.2)
            (loss * 1.5).backward()
            grads = [g[k].grad for k in names] + [ro.grad, rd.grad] + [p.grad for p in model.parameters()]
            res.append((loss.item(), depth.detach().clone(), color.detach().clone(),
                        [None if t is None else t.detach().clone() for t in grads]))
        (l0, d0, c0, g0), (l1, d1, c1, g1) = res
        assert abs(l0 - l1) <= 1e-12 * abs(l0)
        assert torch.equal(d0, d1) and torch.equal(c0, c1)
        n_checked = 0
        for a, b in zip(g0, g1):
            assert (a is None) == (b is None)
            if a is not None:
                assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-30)
                n_checked += 1
        assert n_checked >= 4
    with pytest.raises(ValueError):
        renderer.render_batch_ray_rgbd_loss(grids, model, rays['rays_d'], rays['rays_o'], DEV, 'coarse', rays['gt_depth'],
                                            rays['gt_color'])


def test_coarse_grid_in_device_layout_matches_dense_path():
    """The coarse mapper (Mapper.py:326-328: keys = ['grid_coarse'], stage 'coarse') through MaskedGridOptimizer: same
    outputs and gradients as the dense leaf, and one Adam step equal to torch.optim.Adam on the dense grid."""
    import evennicer_slam_amd as E
    from evennicer_slam_amd.mapper import MaskedGridOptimizer
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        dense = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
        d, v, c = renderer.render_batch_ray(dense, model, rays['rays_d'], rays['rays_o'], DEV, 'coarse')
        w = torch.linspace(0.5, 1.5, d.shape[0], device=DEV, dtype=d.dtype)
        (d * w).sum().backward()
        ref = dense['grid_coarse'].grad
        assert float(ref.abs().max()) > 0
        work = {k: v.clone() for k, v in grids.items()}
        opt = MaskedGridOptimizer(work, None, keys=('grid_coarse',))
        d2, v2, c2 = renderer.render_batch_ray(opt.render_grids(), model, rays['rays_d'], rays['rays_o'], DEV, 'coarse')
        assert torch.equal(d, d2)
        (d2 * w).sum().backward()
        G = opt.grids['grid_coarse']
        D, H, W = G.dims
        got = G.grad_vm.reshape(D, H, W, 32).permute(3, 0, 1, 2)[None]
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        # one optimiser step
        tref = torch.optim.Adam([dense['grid_coarse']], lr=0.01)
        tref.step()
        opt.step({'grid_coarse': 0.01})
        opt.write_back()
        assert float((work['grid_coarse'] - dense['grid_coarse'].detach()).abs().max()) <= 2e-6
        assert float(G.grad_vm.abs().max()) == 0.0                       # accumulators cleared by the step
    finally:
        for p in model.parameters():
            p.requires_grad_(True)


def test_fused_loss_falls_back_above_the_tile_mode_limit(monkeypatch):
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    g = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
    monkeypatch.setattr(type(renderer), 'FUSED_LOSS_MAX_RAYS', 8)       # the 64-ray batch now counts as "large"
    loss, depth, var, color = renderer.render_batch_ray_rgbd_loss(g, model, rays['rays_d'], rays['rays_o'], DEV, 'color',
                                                                  rays['gt_depth'], rays['gt_color'], 0.2)
    loss.backward()
    g2 = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
    monkeypatch.setattr(type(renderer), 'FUSED_LOSS_MAX_RAYS', 32768)
    loss2, *_ = renderer.render_batch_ray_rgbd_loss(g2, model, rays['rays_d'], rays['rays_o'], DEV, 'color', rays['gt_depth'],
                                                    rays['gt_color'], 0.2)
    loss2.backward()
    assert abs(loss.item() - loss2.item()) <= 1e-12 * abs(loss2.item()) and not depth.requires_grad
    a, b = g['grid_color'].grad, g2['grid_color'].grad
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


def test_fused_loss_backward_twice_accumulates():
    """retain_graph: the unit gradients and the work list of the fused-loss forward serve every backward of the call."""
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    g = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
    loss, *_ = renderer.render_batch_ray_rgbd_loss(g, model, rays['rays_d'], rays['rays_o'], DEV, 'color', rays['gt_depth'],
                                                   rays['gt_color'], 0.2)
    loss.backward(retain_graph=True)
    once = g['grid_color'].grad.clone()
    loss.backward()
    twice = g['grid_color'].grad
    assert float(once.abs().max()) > 0
    assert float((twice - 2 * once).abs().max()) <= 1e-5 * float(once.abs().max())
    for p in model.parameters():
        p.grad = None


def test_eager_consumers_see_decoders_a_captured_optimiser_stepped():
    """A captured mapper iteration (render + fused loss + backward + FusedAdam on the colour decoder) mutates the decoder
    parameters on every replay through raw kernel writes, and packs its decoders into graph-pool memory.  An eager
    consumer afterwards (render_img, eval_points for meshing, the visualiser) must render with the CURRENT parameters:
    same result as a freshly cloned module that never saw the graph."""
    import copy
    import gc
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd.graph import GraphedStep
    from evennicer_slam_amd.mapper import FusedAdam
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    g = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
    params = list(model.color_decoder.parameters())
    opt = FusedAdam(params, lr=1e-2)
    one = {}

    def it():
        for t in g.values():
            t.grad = None
        opt.zero_grad()
        loss, *_ = renderer.render_batch_ray_rgbd_loss(g, model, rays['rays_d'], rays['rays_o'], DEV, 'color', rays['gt_depth'],
                                                       rays['gt_color'], 0.2)
        if 'g' not in one:
            one['g'] = torch.ones_like(loss)
        loss.backward(gradient=one['g'])
        opt.step()
        return loss

    before = model.color_decoder.output_linear.weight.detach().clone()
    gc.collect()
    gs = GraphedStep(it)
    versions = [p._version for p in params]
    l0 = gs.replay().item()
    for _ in range(3):
        l1 = gs.replay().item()
    assert l1 != l0                                                     # the optimiser really steps inside the graph
    assert all(p._version > v for p, v in zip(params, versions))        # ... and every replay announces its writes
    assert float((model.color_decoder.output_linear.weight - before).abs().max()) > 0
    with torch.no_grad():
        d1, u1, c1 = renderer.render_batch_ray(grids, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
        fresh = copy.deepcopy(model)
        d2, u2, c2 = renderer.render_batch_ray(grids, fresh, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
        p = rays['rays_o'][:50].double() + 0.3 * rays['rays_d'][:50].double()
        r1 = renderer.eval_points(p, model, grids, 'color', DEV)
        r2 = renderer.eval_points(p, fresh, grids, 'color', DEV)
    assert torch.equal(c1, c2) and torch.equal(d1, d2) and torch.equal(r1, r2)
    # the next replay still works and keeps stepping
    l2 = gs.replay().item()
    assert l2 != l1
    del gs
    EF.clear_caches()


# ------------------------------------------------------------------------------------------------ bundle adjustment
def _ba_setup(static_shapes, monkeypatch):
    """MapperIteration on the tiny scene with the fixture's frames, masks, camera tensors and pixel draws."""
    import evennicer_slam_amd as E
    from evennicer_slam_amd.mapper import MapperIteration
    from tests.hip_util import DEV, cfg_like, tiny_on_gpu
    g = load("tiny_mapper_ba")
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    H, W, fx, fy, cx, cy = [float(v) for v in g['cam']]
    cam = dict(H=int(H), W=int(W), fx=fx, fy=fy, cx=cx, cy=cy)
    cfg = cfg_like()
    cfg['mapping'] = {'w_color_loss': float(g['w_color_loss']), 'lr_factor': float(g['lr_factor']), 'BA': True,
                      'BA_cam_lr': float(g['BA_cam_lr']), 'middle_iter_ratio': 0.4, 'fine_iter_ratio': 0.6, 'fix_fine': True,
                      'fix_color': False, 'pixels': int(g['pixs_per_image']) * 3,
                      'stage': {'coarse': dict(decoders_lr=0.0, coarse_lr=0.001, middle_lr=0.0, fine_lr=0.0, color_lr=0.0),
                                'middle': dict(decoders_lr=0.0, coarse_lr=0.0, middle_lr=0.1, fine_lr=0.0, color_lr=0.0),
                                'fine': dict(decoders_lr=0.0, coarse_lr=0.0, middle_lr=0.005, fine_lr=0.005, color_lr=0.0),
                                'color': dict(decoders_lr=0.005, coarse_lr=0.0, middle_lr=0.005, fine_lr=0.005, color_lr=0.005)}}
    frames = [dict(depth=torch.from_numpy(g[f'depth_{f}']).to(DEV), color=torch.from_numpy(g[f'color_{f}']).to(DEV),
                   c2w=torch.from_numpy(g[f'est_c2w_{f}']).to(DEV), fixed=(f == 0)) for f in range(3)]
    cts = [None] + [torch.from_numpy(g['camera_tensors'][f]) for f in (1, 2)]
    masks = {k: torch.from_numpy(g['mask_' + k]).to(DEV) for k in KEYS}
    it = MapperIteration(cfg, renderer, grids, model, frames, cam, masks=masks, keys=KEYS, camera_tensors=cts,
                         static_shapes=static_shapes)
    draws = iter(torch.from_numpy(g['idx']).reshape(-1, g['idx'].shape[-1]).to(DEV))
    monkeypatch.setattr(torch, 'randint', lambda *a, **k: next(draws))
    return g, it, grids, model


@pytest.mark.parametrize("static_shapes", [False, True])
def test_bundle_adjustment_iterations_match_reference_fixture(monkeypatch, static_shapes):
    """8 joint iterations with BA (Mapper.py:374-407,481-490,502-547): losses, the number of rays the prefilter keeps,
    the gradients reaching the two camera tensors, the camera tensors after every Adam step, final grids and colour
    decoder -- once with the reference's dynamic-shape prefilter and once with the static-shape (capturable) form."""
    g, it, grids, model = _ba_setup(static_shapes, monkeypatch)
    n = int(g['num_joint_iters'])
    for j in range(n):
        grads_before = None
        loss = it.step(j, n)
        assert it.last['stage'] == str(g['stages'][j])
        assert int(it.last['inside'].sum()) == int(g['n_inside'][j])
        assert it.last['n_rays'] == (72 if static_shapes else int(g['n_inside'][j]))
        assert abs(loss.item() - g['losses'][j]) <= 1e-4 * abs(g['losses'][j]), (j, loss.item(), g['losses'][j])
        cams = torch.stack([t.detach() for t in it.camera_tensors if t is not None]).cpu().numpy()
        assert np.abs(cams - g['cam_after'][j]).max() <= 2e-6, j
        cg = torch.stack([t.grad for t in it.camera_tensors if t is not None]).cpu().numpy()
        ref = g['cam_grads'][j]
        assert np.abs(cg - ref).max() <= 1e-3 * np.abs(ref).max(), j
    it.finish()
    moved = np.abs(g['cam_after'][-1] - g['camera_tensors'][1:]).max()
    assert moved > 1e-3                                                  # the cameras really moved (colour stage)
    for k in KEYS:
        assert np.abs(grids[k].cpu().numpy() - g['final_' + k]).max() <= 5e-5, k
    for name, ref in g.items():
        if name.startswith('final_cd_'):
            got = dict(model.color_decoder.named_parameters())[name[len('final_cd_'):]].detach().cpu().numpy()
            assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), name


def test_graphed_mapper_iteration_draws_new_rays_every_replay():
    """The static-shape iteration captured as ONE hipGraph: every replay draws new pixels (torch.randint on the device
    generator inside the graph), filters, renders, steps -- and equals the eager static-shape iteration fed the same
    generator state."""
    import gc
    import evennicer_slam_amd as E
    from evennicer_slam_amd.mapper import MapperIteration
    from tests.hip_util import DEV, cfg_like, tiny_on_gpu
    g = load("tiny_mapper_ba")

    def build():
        s, bound, model, grids, rays, renderer = tiny_on_gpu()
        H, W, fx, fy, cx, cy = [float(v) for v in g['cam']]
        cam = dict(H=int(H), W=int(W), fx=fx, fy=fy, cx=cx, cy=cy)
        cfg = cfg_like()
        cfg['mapping'] = {'w_color_loss': 0.2, 'lr_factor': 1.0, 'BA': True, 'BA_cam_lr': 0.001, 'middle_iter_ratio': 0.4,
                          'fine_iter_ratio': 0.6, 'fix_fine': True, 'fix_color': False, 'pixels': 72,
                          'stage': {'color': dict(decoders_lr=0.005, coarse_lr=0.0, middle_lr=0.005, fine_lr=0.005, color_lr=0.005)}}
        frames = [dict(depth=torch.from_numpy(g[f'depth_{f}']).to(DEV), color=torch.from_numpy(g[f'color_{f}']).to(DEV),
                       c2w=torch.from_numpy(g[f'est_c2w_{f}']).to(DEV), fixed=(f == 0)) for f in range(3)]
        cts = [None] + [torch.from_numpy(g['camera_tensors'][f]) for f in (1, 2)]
        masks = {k: torch.from_numpy(g['mask_' + k]).to(DEV) for k in KEYS}
        return MapperIteration(cfg, renderer, grids, model, frames, cam, masks=masks, keys=KEYS, camera_tensors=cts,
                               static_shapes=True), grids

    # eager reference: 3 warm-up iterations (what GraphedStep runs before capturing) + 4 more, one RNG stream
    it_e, grids_e = build()
    torch.manual_seed(5)
    for _ in range(3):
        it_e.step(0, 1, stage='color')
    eager = [it_e.step(0, 1, stage='color').item() for _ in range(4)]
    cams_e = torch.stack([t.detach() for t in it_e.camera_tensors if t is not None]).clone()
    gc.collect()
    it_g, grids_g = build()
    torch.manual_seed(5)
    gs = it_g.graphed('color', warmup=3)
    graphed = [gs.replay().item() for _ in range(4)]
    torch.cuda.synchronize()
    assert len(set(graphed)) == 4                                        # new pixels every replay
    # same generator stream -> same draws -> same losses and the same cameras
    assert np.abs(np.array(graphed) - np.array(eager)).max() <= 1e-5 * max(eager)
    cams_g = torch.stack([t.detach() for t in it_g.camera_tensors if t is not None])
    assert float((cams_g - cams_e).abs().max()) <= 2e-5       # (float atomics reorder sums; Adam normalises tiny gradient differences)
    del gs
    gc.collect()
