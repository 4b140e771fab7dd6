"""-m gpu: one camera iteration of the tracker (pose -> rays -> HIP render -> uncertainty-weighted loss -> pose
gradient) against the fixture produced by the reference's own statements (tests/golden/tiny_tracker_iter.npz)."""
import numpy as np
import pytest
import torch

from tests.util import load, rel_err

pytestmark = pytest.mark.gpu


def test_pose_gradient_matches_reference_fixture():
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, tiny_on_gpu
    g = load("tiny_tracker_iter")
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    H, W, fx, fy, cx, cy = [float(x) for x in g['cam']]
    He, We = int(g['edge'][0]), int(g['edge'][1])
    for p in model.parameters():                       # the tracker optimises the camera only (Tracker.py:248-260)
        p.requires_grad_(False)
    try:
        ct = torch.from_numpy(g['camera_tensor']).to(DEV).requires_grad_(True)
        c2w = E.common.get_camera_from_tensor(ct)
        assert np.array_equal(c2w.detach().cpu().numpy(), g['c2w'])
        # the reference's pixel draw (the RNG streams of CPU and HIP generators differ: identical indices are fed)
        idx = torch.from_numpy(g['idx']).to(DEV)
        ww = int(W) - 2 * We
        i = (We + idx % ww).float()
        j = (He + idx // ww).float()
        ro, rd = E.common.get_rays_from_uv(i, j, c2w, int(H), int(W), fx, fy, cx, cy, DEV)
        assert np.allclose(rd.detach().cpu().numpy(), g['rays_d_all'], rtol=0, atol=1e-6)
        gd_img = torch.from_numpy(g['gt_depth']).to(DEV)
        gc_img = torch.from_numpy(g['gt_color']).to(DEV)
        pix = (j.long() * int(W) + i.long())
        gd, gc = gd_img.reshape(-1)[pix], gc_img.reshape(-1, 3)[pix]
        with torch.no_grad():                          # Tracker.py:164-170
            t = (bound.to(DEV).float().unsqueeze(0) - ro.detach().unsqueeze(-1)) / rd.detach().unsqueeze(-1)
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            inside = t >= gd
        assert np.array_equal(inside.cpu().numpy(), g['inside_mask'])
        ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
        depth, unc, color = renderer.render_batch_ray(grids, model, rd, ro, DEV, 'color', gt_depth=gd)
        assert rel_err(depth.detach().cpu().numpy(), g['depth']) <= 1e-4
        assert rel_err(color.detach().cpu().numpy(), g['color']) <= 1e-4
        unc = unc.detach()
        mask = gd > 0
        loss = (torch.abs(gd - depth) / torch.sqrt(unc + 1e-10))[mask].sum()
        loss = loss + float(g['w_color_loss']) * torch.abs(gc - color)[mask].sum()
        assert abs(loss.item() - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
        loss.backward()
        assert rel_err(ct.grad.cpu().numpy(), g['g_camera_tensor']) <= 1e-3
    finally:
        for p in model.parameters():
            p.requires_grad_(True)


def test_fused_pose_rays_and_loss_match_torch_route():
    """tracker.rays_from_camera_tensor / losses.tracker_loss against the plain-torch formulation of the same
    statements (common.get_camera_from_tensor + get_rays_from_uv; Tracker.py:187-195), values and gradients."""
    import evennicer_slam_amd as E
    from tests.hip_util import DEV
    g = torch.Generator().manual_seed(3)
    n = 300
    i = (torch.rand(n, generator=g) * 1199).floor().to(DEV)
    j = (torch.rand(n, generator=g) * 679).floor().to(DEV)
    fx, fy, cx, cy = 600.0, 600.0, 599.5, 339.5
    ct0 = torch.tensor([0.7, -0.2, 0.6, 0.1, 3.0, 1.0, -0.5])
    cot_o, cot_d = torch.randn(n, 3, generator=g).to(DEV), torch.randn(n, 3, generator=g).to(DEV)
    a = ct0.clone().to(DEV).requires_grad_(True)
    ro, rd = E.tracker.rays_from_camera_tensor(a, i, j, fx, fy, cx, cy)
    ((ro * cot_o).sum() + (rd * cot_d).sum()).backward()
    b = ct0.clone().to(DEV).requires_grad_(True)
    ro2, rd2 = E.common.get_rays_from_uv(i, j, E.common.get_camera_from_tensor(b), 680, 1200, fx, fy, cx, cy, DEV)
    ((ro2 * cot_o).sum() + (rd2 * cot_d).sum()).backward()
    assert float((rd - rd2).abs().max()) <= 2e-6 and torch.equal(ro, ro2.contiguous())
    assert float((a.grad - b.grad).abs().max()) <= 1e-4 * float(b.grad.abs().max())
    # loss
    depth = (torch.rand(n, generator=g).double() * 3).to(DEV).requires_grad_(True)
    unc = (torch.rand(n, generator=g).double() * 0.1).to(DEV)
    color = torch.rand(n, 3, generator=g).to(DEV).requires_grad_(True)
    gd = (torch.rand(n, generator=g) * 3).to(DEV)
    gd[::7] = 0.0
    gc = torch.rand(n, 3, generator=g).to(DEV)
    loss = E.losses.tracker_loss(depth, unc, color, gd, gc, 0.5)
    loss.backward()
    d2, c2 = depth.detach().clone().requires_grad_(True), color.detach().clone().requires_grad_(True)
    mask = gd > 0
    ref = (torch.abs(gd - d2) / torch.sqrt(unc + 1e-10))[mask].sum() + 0.5 * torch.abs(gc - c2)[mask].sum()
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
    assert float((depth.grad - d2.grad).abs().max()) <= 1e-9 * float(d2.grad.abs().max())
    assert torch.equal(color.grad, c2.grad)


@pytest.mark.parametrize("cdtype", [torch.float32, torch.float64])
def test_get_sample_uv_fused_gather_is_bit_exact(cdtype):
    """common.get_sample_uv on GPU tensors (one enslam_gather_pixels launch behind the torch.randint draw) returns exactly what
    the reference's formulation returns -- meshgrid of the window, select_uv (common.py:125-141) -- for the same generator
    state: pixel coordinates, depth and colour samples bit for bit, float32 and float64 colours, an off-origin window."""
    import evennicer_slam_amd.common as C
    H, W, n = 60, 84, 500
    H0, H1, W0, W1 = 7, 55, 5, 80
    g = torch.Generator().manual_seed(3)
    depth = (torch.rand(H, W, generator=g) * 4).cuda()
    color = torch.rand(H, W, 3, generator=g).to(cdtype).cuda()
    torch.manual_seed(1234)
    i, j, d, c = C.get_sample_uv(H0, H1, W0, W1, n, depth, color, device='cuda:0')
    torch.manual_seed(1234)
    dd, cc = depth[H0:H1, W0:W1], color[H0:H1, W0:W1]
    ii, jj = torch.meshgrid(torch.linspace(W0, W1 - 1, W1 - W0).cuda(), torch.linspace(H0, H1 - 1, H1 - H0).cuda(), indexing='ij')
    ii, jj = ii.t(), jj.t()
    ri, rj, rd, rc = C.select_uv(ii, jj, n, dd, cc, device='cuda:0')
    assert i.dtype == ri.dtype and c.dtype == rc.dtype == cdtype and d.dtype == rd.dtype
    assert torch.equal(i, ri) and torch.equal(j, rj) and torch.equal(d, rd) and torch.equal(c, rc)


def test_tracking_loop_twin_against_the_oracle(monkeypatch):
    """the harness's tracking loop of one frame -- `tracking.iters` camera iterations of
    `TrackerIteration.optimize_cam_in_batch` (pose -> rays -> HIP render -> uncertainty-weighted loss -> backward -> Adam) with
    the least-loss candidate kept (slam.SLAM.track, Tracker.py:321-330) -- against an oracle-driven twin on the CPU
    (oracle/tracker_oracle.camera_iteration + torch.optim.Adam) ON THE SAME PIXEL DRAWS (both sides pop the draws from one
    pre-generated list: the CPU and HIP generators produce different streams).  Camera tensor after every iteration and the
    candidate finally kept agree to 1e-4."""
    import types
    import evennicer_slam_amd as E
    from evennicer_slam_amd.mapper import FusedAdam
    from oracle import tracker_oracle as TO
    from tests.hip_util import DEV, cfg_like, tiny_on_gpu
    from tests.util import tiny_scene
    g = load("tiny_tracker_iter")
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    H, W, fx, fy, cx, cy = [float(x) for x in g['cam']]
    H, W = int(H), int(W)
    He, We = int(g['edge'][0]), int(g['edge'][1])
    w_color, K, n, lr = float(g['w_color_loss']), 6, 48, 2e-3
    gen = torch.Generator().manual_seed(11)
    draws = [torch.randint((H - 2 * He) * (W - 2 * We), (n,), generator=gen) for _ in range(K)]
    ct0 = torch.from_numpy(g['camera_tensor']).float()
    gd_img, gc_img = torch.from_numpy(g['gt_depth']), torch.from_numpy(g['gt_color'])

    # ---- oracle twin (CPU)
    params, ogrids, obound, _ = tiny_scene()
    oct = ct0.clone().requires_grad_(True)
    oopt = torch.optim.Adam([oct], lr=lr)
    o_traj, o_losses = [], []
    for k in range(K):
        oopt.zero_grad()
        loss, _d, _v, _c, _inside = TO.camera_iteration(params, ogrids, obound, oct, gd_img, gc_img, (H, W, fx, fy, cx, cy), (He, We), n,
                                                        w_color, idx=draws[k])
        o_losses.append(float(loss.item()))
        o_traj.append(oct.detach().clone())              # the candidate this loss belongs to (the pose BEFORE the step)
        loss.backward()
        oopt.step()

    # ---- the harness's loop on the HIP path, fed the same draws
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        cfg = cfg_like()
        cfg['tracking'] = {'device': DEV, 'w_color_loss': w_color, 'ignore_edge_W': We, 'ignore_edge_H': He, 'handle_dynamic': False,
                           'use_color_in_tracking': True, 'lr': lr, 'pixels': n, 'iters': K}
        cfg['event'] = {'activate_events': False, 'blur': False, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                        'balancer': 0.025}
        slam = types.SimpleNamespace(nice=True, bound=bound, renderer=renderer, event_net=None, low_gpu_mem=False, H=H, W=W, fx=fx,
                                     fy=fy, cx=cx, cy=cy)
        trk = E.tracker.TrackerIteration(cfg, None, slam)
        trk.c, trk.decoders = grids, model
        queue = list(draws)
        real_randint = torch.randint

        def fake_randint(high, size, *a, device=None, **kw):
            if tuple(size) == (n,) and queue:
                return queue.pop(0).to(device if device is not None else 'cpu')
            return real_randint(high, size, *a, device=device, **kw)

        monkeypatch.setattr(torch, 'randint', fake_randint)
        ct = ct0.clone().to(DEV).requires_grad_(True)
        opt = FusedAdam([ct], lr=lr)
        gd_dev, gc_dev = gd_img.to(DEV), gc_img.to(DEV)
        best, best_loss, h_losses = None, None, []
        for k in range(K):
            before = ct.detach().clone()
            out = trk.optimize_cam_in_batch(ct, None, gc_dev, gd_dev, None, None, n, opt, 1, k, None, rgbd=True, event=False)
            h_losses.append(out[0])
            assert float((before.cpu() - o_traj[k]).abs().max()) <= 1e-4, k       # same pose entering iteration k
            if best_loss is None or out[0] < best_loss:
                best_loss, best = out[0], before
        assert not queue                                                        # every draw was consumed by the HIP side
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
    for a, b in zip(h_losses, o_losses):
        assert abs(a - b) <= 1e-4 * max(abs(b), 1.0)
    k_best = int(np.argmin(o_losses))
    assert float((best.cpu() - o_traj[k_best]).abs().max()) <= 1e-4
    assert float((ct.detach().cpu() - oct.detach()).abs().max()) <= 1e-4        # after the last Adam step
    assert float((ct0 - oct.detach()).abs().max()) > 1e-3                       # (the pose really moved)


def test_tracker_rays_entry_matches_the_torch_route():
    """enslam_tracker_rays (pixels, rays, in-bound mask, batch maxima in one launch) against get_sample_uv + rays_from_camera_tensor
    + the torch statements of Tracker.py:164-170 on the same indices."""
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd.tracker import _TrackerRays
    from tests.hip_util import DEV
    torch.manual_seed(3)
    H, W, He, We, n = 120, 160, 10, 12, 777
    fx, fy, cx, cy = 150.0, 152.0, 79.5, 59.5
    depth = (torch.rand(H, W, device=DEV) * 6.0)
    depth[::7, ::5] = 0.0
    bound = torch.tensor([[-2.0, 2.5], [-1.5, 2.0], [-1.0, 1.8]], dtype=torch.float64)
    ct = torch.tensor([0.9, 0.1, -0.2, 0.05, 0.3, -0.2, 0.1], device=DEV, requires_grad=True)
    ww = W - 2 * We
    idx = torch.randint((H - 2 * He) * ww, (n,), device=DEV)
    for dtype in (torch.float32, torch.float64):
        color = torch.rand(H, W, 3, device=DEV, dtype=dtype)
        ro, rd, gd, gc, inside, dmax = _TrackerRays.apply(ct, idx, He, We, ww, depth, color, fx, fy, cx, cy, EF.bound6(bound), True)
        i, j, d2, c2 = EF.gather_pixels(idx, He, We, ww, depth, color)
        ro2, rd2 = E.tracker.rays_from_camera_tensor(ct, i, j, fx, fy, cx, cy)
        assert torch.equal(ro, ro2) and torch.equal(rd, rd2) and torch.equal(gd, d2) and torch.equal(gc, c2.float())
        with torch.no_grad():
            t = (bound.to(DEV).unsqueeze(0) - ro2.detach().unsqueeze(-1)) / rd2.detach().unsqueeze(-1)
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            ins2 = t >= d2
            m = torch.where(ins2, d2, d2.new_zeros(())).max().reshape(1)
        assert torch.equal(inside.bool(), ins2) and 0 < int(ins2.sum()) < n
        assert torch.equal(dmax, torch.cat([m, m * 1.2]))
        g1 = torch.autograd.grad((rd * torch.arange(3, device=DEV)).sum() + ro.sum(), ct)[0]
        g2 = torch.autograd.grad((rd2 * torch.arange(3, device=DEV)).sum() + ro2.sum(), ct)[0]
        assert torch.equal(g1, g2)
    # without the prefilter: no mask, maxima over every ray
    ro, rd, gd, gc, inside, dmax = _TrackerRays.apply(ct, idx, He, We, ww, depth, color, fx, fy, cx, cy, EF.bound6(bound), False)
    assert inside is None and float(dmax[0]) == float(gd.max())


@pytest.mark.parametrize("n_rep", [1, 3, 13, 20])          # 64, 192, 832 rays (median by counting, 16 / 4 / 1 lanes per entry) / 1280 rays (sorting network)
@pytest.mark.parametrize("dynamic,use_color,masked", [(True, True, True), (True, True, False), (False, True, True), (True, False, False)])
def test_fused_tracker_loss_matches_the_torch_statements(n_rep, dynamic, use_color, masked):
    """Renderer.render_batch_ray_tracker_loss against render_batch_ray + Tracker.py:176-195 written in torch ops (median mask over
    the kept rays included): loss, outputs and the gradient to the rays."""
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    torch.manual_seed(5)
    ro0, rd0 = rays['rays_o'].repeat(n_rep, 1), rays['rays_d'].repeat(n_rep, 1)
    gd, gc = rays['gt_depth'].repeat(n_rep).clone(), rays['gt_color'].repeat(n_rep, 1)
    n = ro0.shape[0]
    rd0 = rd0 + 0.01 * torch.randn_like(rd0) * (torch.arange(n, device=DEV) >= 64)[:, None]     # (the copies look elsewhere)
    gd[torch.arange(n, device=DEV) % 11 == 3] = 0.0
    inside = (torch.arange(n, device=DEV) % 5 != 1) if masked else None
    w = 0.5
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        ro, rd = ro0.clone().requires_grad_(True), rd0.clone().requires_grad_(True)
        loss, depth, unc, color = renderer.render_batch_ray_tracker_loss(grids, model, rd, ro, DEV, 'color', gd, gc, w, inside=inside,
                                                                         handle_dynamic=dynamic, use_color=use_color)
        g_ro, g_rd = torch.autograd.grad(loss * 1.5, [ro, rd])
        ro2, rd2 = ro0.clone().requires_grad_(True), rd0.clone().requires_grad_(True)
        d2, u2, c2 = renderer.render_batch_ray(grids, model, rd2, ro2, DEV, 'color', gt_depth=gd)
        u2 = u2.detach()
        tmp = torch.abs(gd - d2) / torch.sqrt(u2 + 1e-10)
        keep = torch.ones(n, dtype=torch.bool, device=DEV) if inside is None else inside.clone()
        if dynamic:
            med = tmp.detach()[keep].median()
            keep = keep & (tmp.detach() < 10 * med)
        mask = keep & (gd > 0)
        ref = tmp[mask].sum()
        if use_color:
            ref = ref + w * torch.abs(gc - c2)[mask].sum()
        r_ro, r_rd = torch.autograd.grad(ref * 1.5, [ro2, rd2])
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
    assert torch.equal(depth, d2.detach()) and torch.equal(unc, u2) and torch.equal(color, c2.detach())
    assert 0 < int(mask.sum()) < n
    assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())          # (the colour term is a float32 sum in the torch statements)
    for a, b in ((g_ro, r_ro), (g_rd, r_rd)):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())           # (float atomics into the ray gradients)


def test_fused_adam_counts_its_own_step_for_a_camera_tensor():
    """one-workgroup FusedAdam jobs (a 7-number camera tensor): the launch increments the step count itself; same updates as
    torch.optim.Adam over five steps"""
    from evennicer_slam_amd.mapper import FusedAdam
    from tests.hip_util import DEV
    torch.manual_seed(2)
    a = torch.randn(7, device=DEV).requires_grad_(True)
    b = a.detach().clone().requires_grad_(True)
    oa, ob = FusedAdam([a], lr=2e-3), torch.optim.Adam([b], lr=2e-3)
    for k in range(5):
        g = torch.randn(7, device=DEV)
        a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
        assert int(oa.step_t.item()) == k + 1
        assert float((a - b).abs().max()) <= 1e-6


def test_tracker_loss_entry_refuses_batches_beyond_its_limit():
    import evennicer_slam_amd._lib as L
    from tests.hip_util import tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    assert L.lib().enslam_tracker_tail_max_rays() == 4096
    assert not renderer.tracker_loss_ok(5000, rays['gt_depth']) and renderer.tracker_loss_ok(4096, rays['gt_depth'])
    with pytest.raises(ValueError):
        renderer.render_batch_ray_tracker_loss(grids, model, rays['rays_d'].repeat(80, 1), rays['rays_o'].repeat(80, 1), 'cuda:0', 'color',
                                               rays['gt_depth'].repeat(80), rays['gt_color'].repeat(80, 1))


@pytest.mark.parametrize("dynamic", [True, False])
def test_fused_and_launch_per_step_routes_of_the_rgbd_term_agree(monkeypatch, dynamic):
    """TrackerIteration._rgbd_loss through the fused launches (enslam_tracker_rays + enslam_render_tracker_loss_fwd) and through
    the launch-per-step route (ENSLAM_TRACKER_FUSED=0: gather, pose, prefilter and median as torch ops, losses.tracker_loss) on
    the same pixel draw: same loss, same gradient to the camera tensor."""
    import types
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, cfg_like, tiny_on_gpu
    g = load("tiny_tracker_iter")
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    H, W, fx, fy, cx, cy = [float(x) for x in g['cam']]
    H, W = int(H), int(W)
    He, We = int(g['edge'][0]), int(g['edge'][1])
    n = 96
    idx = torch.randint((H - 2 * He) * (W - 2 * We), (n,), generator=torch.Generator().manual_seed(4)).to(DEV)
    monkeypatch.setattr(torch, 'randint', lambda *a, **k: idx)
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        cfg = cfg_like()
        cfg['tracking'] = {'device': DEV, 'w_color_loss': 0.5, 'ignore_edge_W': We, 'ignore_edge_H': He, 'handle_dynamic': dynamic,
                           'use_color_in_tracking': True, 'lr': 1e-3, 'pixels': n, 'iters': 4}
        cfg['event'] = {'activate_events': False, 'blur': False, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                        'balancer': 0.025}
        slam = types.SimpleNamespace(nice=True, bound=bound, renderer=renderer, event_net=None, low_gpu_mem=False, H=H, W=W, fx=fx,
                                     fy=fy, cx=cx, cy=cy)
        trk = E.tracker.TrackerIteration(cfg, None, slam)
        trk.c, trk.decoders = grids, model
        gd_img, gc_img = torch.from_numpy(g['gt_depth']).to(DEV), torch.from_numpy(g['gt_color']).to(DEV)
        out = []
        for fused in (True, False):
            monkeypatch.setattr(E.tracker, 'FUSED_ITERATION', fused)
            ct = torch.from_numpy(g['camera_tensor']).to(DEV).requires_grad_(True)
            for static in ((True, False) if not fused else (False,)):
                ct.grad = None
                loss = trk._rgbd_loss(ct, gc_img, gd_img, n, static)
                loss.backward()
                out.append((loss.item(), ct.grad.clone()))
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
    (l0, g0) = out[0]
    assert l0 > 0 and float(g0.abs().max()) > 0
    for l1, g1 in out[1:]:
        assert abs(l0 - l1) <= 1e-6 * abs(l1)
        assert float((g0 - g1).abs().max()) <= 1e-4 * float(g1.abs().max())
