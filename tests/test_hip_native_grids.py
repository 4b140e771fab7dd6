"""Feature grids given as channels_last_3d tensors (same shape / values / indexing as the reference's [1,32,D,H,W] grids, storage
= the kernels' own [V][32]): read in place, gradients returned in the same memory format without a transposed copy, sampler and
prepare roles in one launch.  Everything here compares against the contiguous-grid route, which the other files pin to the
reference fixtures (tests/test_hip_scenes.py also runs both layouts against them)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEYS = ('grid_middle', 'grid_fine', 'grid_color')


def _run(renderer, model, grids, rays, stage, layout, fused=False, need_rays=True, w=None):
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, as_layout
    for p in model.parameters():
        p.grad = None
    cg = {k: as_layout(v, layout).requires_grad_(True) for k, v in grids.items()}
    ro = rays['rays_o'].clone().requires_grad_(need_rays)
    rd = rays['rays_d'].clone().requires_grad_(need_rays)
    if fused:
        loss, depth, var, color = renderer.render_batch_ray_rgbd_loss(cg, model, rd, ro, DEV, stage, rays['gt_depth'], rays['gt_color'], 0.2)
    else:
        depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, DEV, stage, gt_depth=rays['gt_depth'])
        loss = E.losses.rgbd_loss(depth, color if stage == 'color' else None, rays['gt_depth'], rays['gt_color'], 0.2)
    (loss * (w if w is not None else 1.0)).backward()
    return cg, ro, rd, (depth.detach(), var.detach(), color.detach()), loss.detach()


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("stage", ['middle', 'fine', 'color'])
def test_values_and_gradients_equal_the_contiguous_route(stage, fused):
    from tests.hip_util import tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    a = _run(renderer, model, grids, rays, stage, 'contiguous', fused)
    pa = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    b = _run(renderer, model, grids, rays, stage, 'channels_last_3d', fused)
    pb = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    for x, y in zip(a[3], b[3]):
        assert torch.equal(x, y)                                       # the same gathers, the same arithmetic
    assert abs(float(a[4]) - float(b[4])) <= 1e-12 * abs(float(a[4]))            # (fused loss: float64 atomics, order varies)
    used = {'middle': KEYS[:1], 'fine': KEYS[:2], 'color': KEYS}[stage]
    for k in KEYS:
        ga, gb = a[0][k].grad, b[0][k].grad
        if k not in used:
            assert ga is None and gb is None
            continue
        assert gb.shape == ga.shape and gb.is_contiguous(memory_format=torch.channels_last_3d)
        assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max()), k      # (float-atomic ordering)
        assert bool(((ga != 0) == (gb != 0)).all()) or float((ga - gb).abs().max()) <= 1e-6 * float(ga.abs().max())
    for x, y in ((a[1].grad, b[1].grad), (a[2].grad, b[2].grad)):
        assert float((x - y).abs().max()) <= 1e-5 * float(x.abs().max())
    assert set(pa) == set(pb) and len(pa) > 0
    for n in pa:
        assert float((pa[n] - pb[n]).abs().max()) <= 1e-5 * max(float(pa[n].abs().max()), 1e-30), n


def test_mixed_layouts_in_one_call():
    """one grid channels_last_3d, the others contiguous: the converting route for those, in place for this one"""
    from tests.hip_util import DEV, as_layout, tiny_on_gpu
    import evennicer_slam_amd as E
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    a = _run(renderer, model, grids, rays, 'color', 'contiguous')
    cg = {k: as_layout(v, 'channels_last_3d' if k == 'grid_fine' else 'contiguous').requires_grad_(True) for k, v in grids.items()}
    ro, rd = rays['rays_o'].clone().requires_grad_(True), rays['rays_d'].clone().requires_grad_(True)
    depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, DEV, 'color', gt_depth=rays['gt_depth'])
    E.losses.rgbd_loss(depth, color, rays['gt_depth'], rays['gt_color'], 0.2).backward()
    assert torch.equal(depth.detach(), a[3][0]) and torch.equal(color.detach(), a[3][2])
    for k in KEYS:
        assert float((cg[k].grad - a[0][k].grad).abs().max()) <= 1e-5 * float(a[0][k].grad.abs().max()), k
    assert cg['grid_fine'].grad.is_contiguous(memory_format=torch.channels_last_3d) and cg['grid_color'].grad.is_contiguous()


def test_static_native_grids_and_forward_only_consumers():
    """grids without gradient (tracker, render_img, eval_points) in channels_last_3d: no cached copy is made, the values are
    the contiguous route's"""
    import evennicer_slam_amd.functional as EF
    from tests.hip_util import DEV, as_layout, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    nat = {k: as_layout(v, 'channels_last_3d') for k, v in grids.items()}
    n0 = len(EF._grid_cache.items)
    with torch.no_grad():
        d0, v0, c0 = renderer.render_batch_ray(grids, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
        n1 = len(EF._grid_cache.items)
        d1, v1, c1 = renderer.render_batch_ray(nat, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
        assert len(EF._grid_cache.items) == n1 and n1 > n0
        pts = (torch.rand(300, 3, device=DEV, dtype=torch.float64) - 0.5) * 2
        r0 = renderer.eval_points(pts, model, grids, 'color', DEV)
        r1 = renderer.eval_points(pts, model, nat, 'color', DEV)
    assert torch.equal(d0, d1) and torch.equal(v0, v1) and torch.equal(c0, c1) and torch.equal(r0, r1)
    # pose gradients through a fixed native map (the tracker's case): rays need grad, grids do not
    ro, rd = rays['rays_o'].clone().requires_grad_(True), rays['rays_d'].clone().requires_grad_(True)
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        out = []
        for c in (grids, nat):
            ro.grad = rd.grad = None
            d, v, col = renderer.render_batch_ray(c, model, rd, ro, DEV, 'color', gt_depth=rays['gt_depth'])
            (d.sum() + col.sum().double()).backward()
            out.append((ro.grad.clone(), rd.grad.clone()))
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
    for x, y in zip(out[0], out[1]):
        assert float((x - y).abs().max()) <= 1e-5 * float(x.abs().max())


def test_repeated_backward_returns_independent_gradients():
    """retain_graph: the second backward must not clobber what the first returned (the returned tensors alias the call's flat
    accumulator: a repeated backward adds into a fresh one)"""
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, as_layout, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    for layout in ('contiguous', 'channels_last_3d'):
        cg = {k: as_layout(v, layout).requires_grad_(True) for k, v in grids.items()}
        ro, rd = rays['rays_o'].clone().requires_grad_(True), rays['rays_d'].clone().requires_grad_(True)
        depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, DEV, 'color', gt_depth=rays['gt_depth'])
        loss = E.losses.rgbd_loss(depth, color, rays['gt_depth'], rays['gt_color'], 0.2)
        leaves = [cg['grid_fine'], cg['grid_color'], ro, rd]
        g1 = torch.autograd.grad(loss, leaves, retain_graph=True)
        keep = [t.clone() for t in g1]
        g2 = torch.autograd.grad(loss * 3.0, leaves)
        torch.cuda.synchronize()
        for a, k, b in zip(g1, keep, g2):
            assert torch.equal(a, k), layout                                            # the first result is untouched
            assert float((b - 3.0 * k).abs().max()) <= 2e-5 * float(k.abs().max()) * 3, layout


def test_adam_on_native_grids_matches_contiguous():
    """torch.optim.Adam over channels_last_3d leaves + gradients (what an unchanged Mapper does with such grids): the same
    parameter values after three steps as with contiguous grids"""
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, as_layout, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        res = {}
        for layout in ('contiguous', 'channels_last_3d'):
            cg = {k: as_layout(v, layout).requires_grad_(True) for k, v in grids.items()}
            opt = torch.optim.Adam([cg[k] for k in KEYS], lr=0.01)
            for _ in range(3):
                opt.zero_grad()
                depth, var, color = renderer.render_batch_ray(cg, model, rays['rays_d'], rays['rays_o'], DEV, 'color',
                                                              gt_depth=rays['gt_depth'])
                E.losses.rgbd_loss(depth, color, rays['gt_depth'], rays['gt_color'], 0.2).backward()
                opt.step()
            res[layout] = {k: cg[k].detach().contiguous() for k in KEYS}
            assert all(cg[k].is_contiguous(memory_format=torch.channels_last_3d) == (layout == 'channels_last_3d') for k in KEYS)
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
    for k in KEYS:
        assert float((res['contiguous'][k] - res['channels_last_3d'][k]).abs().max()) <= 2e-4 * 0.01 * 3 + 1e-6, k


def test_sample_prepare_entry_equals_the_two_launches():
    """C ABI: enslam_sample_prepare == enslam_sample_rays_g (z_vals bit-equal, same block marks) + enslam_step_prepare's
    clearing role (flat range and flagged blocks cleared)."""
    import evennicer_slam_amd._lib as L
    import evennicer_slam_amd.functional as EF
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    lib = L.lib()
    ro, rd, gd = rays['rays_o'].contiguous(), rays['rays_d'].contiguous(), rays['gt_depth'].contiguous()
    N, n_lin, n_surf = ro.shape[0], 32, 16
    t_lin, t_surf = renderer._t_vals(DEV, n_lin, n_surf)
    msc = L.Scene()
    msc.bound, msc.coarse_bound = EF.bound6(bound), EF.bound6(bound * 2)
    res = []
    for merged in (False, True):
        z = torch.empty((N, n_lin + n_surf), dtype=torch.float64, device=DEV)
        scratch = torch.empty(2, dtype=torch.float32, device=DEV)
        fl, fptr = {}, (ctypes.c_void_p * 4)()
        for k, key in ((1, 'grid_middle'), (2, 'grid_fine'), (3, 'grid_color')):
            D, H, W = grids[key].shape[2:]
            msc.grids[k].D, msc.grids[k].H, msc.grids[k].W = D, H, W
            fl[k] = torch.zeros((D * H * W + 63) // 64, dtype=torch.uint8, device=DEV)
            fptr[k] = fl[k].data_ptr()
        V = grids['grid_fine'].shape[2] * grids['grid_fine'].shape[3] * grids['grid_fine'].shape[4]
        acc = torch.ones(V * 32, device=DEV)
        need = torch.zeros((V + 63) // 64, dtype=torch.uint8, device=DEV)
        need[::3] = 1
        flat = torch.ones(5000, device=DEV)
        zd, zv, zn = (ctypes.c_void_p * 1)(acc.data_ptr()), (ctypes.c_int64 * 1)(V), (ctypes.c_void_p * 1)(need.data_ptr())
        if merged:
            L.check(lib.enslam_sample_prepare(N, n_lin, n_surf, EF._ptr(ro), EF._ptr(rd), EF._ptr(gd), msc.bound, EF._ptr(t_lin),
                                              EF._ptr(t_surf), 0, None, EF._ptr(scratch), 0, EF._ptr(z), 3, ctypes.byref(msc), fptr, 64,
                                              None, 0, None, None, None, 1, zd, zv, zn, EF._ptr(flat), 4999, None), "sample_prepare")
        else:
            L.check(lib.enslam_sample_rays_g(N, n_lin, n_surf, EF._ptr(ro), EF._ptr(rd), EF._ptr(gd), msc.bound, EF._ptr(t_lin),
                                             EF._ptr(t_surf), 0, None, EF._ptr(scratch), 0, EF._ptr(z), 3, ctypes.byref(msc), fptr, 64,
                                             None, None), "sample_rays_g")
            L.check(lib.enslam_step_prepare(0, None, None, None, 0, None, None, None, None, None, 1, zd, zv, zn, EF._ptr(flat), 4999,
                                            None), "step_prepare")
        torch.cuda.synchronize()
        res.append((z, fl, acc, flat))
    (z0, f0, a0, x0), (z1, f1, a1, x1) = res
    assert torch.equal(z0, z1) and np.array_equal(z1.cpu().numpy(), s['z_vals'] if 'z_vals' in s else z0.cpu().numpy())
    for k in (1, 2, 3):
        assert torch.equal(f0[k], f1[k]) and int(f1[k].sum()) > 0
    assert torch.equal(a0, a1) and torch.equal(x0, x1)
    assert float(x1[:4999].abs().sum()) == 0.0 and float(x1[4999]) == 1.0
    blocks = a1[:(V // 64) * 64 * 32].view(-1, 64 * 32)
    nb = blocks.shape[0]
    assert bool((blocks[::3] == 0).all()) and bool((blocks[1::3] == 1).all()) and nb > 3
    # n_rays = 0: the prepare roles alone
    flat.fill_(1.0)
    L.check(lib.enslam_sample_prepare(0, n_lin, n_surf, None, None, None, msc.bound, None, None, 0, None, None, 0, None, 3, None, None, 64,
                                      None, 0, None, None, None, 0, None, None, None, EF._ptr(flat), 100, None), "sample_prepare")
    torch.cuda.synchronize()
    assert float(flat[:100].abs().sum()) == 0.0 and float(flat[100]) == 1.0


def test_reference_mapper_formulation_on_native_grids():
    """The reference's optimize_map statements around this renderer (Mapper.py:343-361, 448-458, 573-602: compact leaves
    `val[mask]`, `val[mask] = val_grad` re-materialisation every iteration, torch.optim.Adam, write-back) with the shared grids in
    channels_last_3d: the same optimised voxels as with contiguous grids (the grid handed to the renderer is then a NON-leaf
    channels_last_3d tensor whose gradient flows on through IndexPutBackward)."""
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, as_layout, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        res = {}
        for layout in ('contiguous', 'channels_last_3d'):
            c = {k: as_layout(v, layout) for k, v in grids.items()}
            gen = torch.Generator().manual_seed(9)
            mask5, leaf = {}, {}
            for k in KEYS:
                m = (torch.rand(tuple(c[k].shape[2:]), generator=gen) < 0.6).to(DEV)
                mask5[k] = m[None, None].repeat(1, 32, 1, 1, 1)
                leaf[k] = c[k][mask5[k]].clone().requires_grad_(True)
            opt = torch.optim.Adam([leaf[k] for k in KEYS], lr=0.01)
            for _ in range(2):
                for k in KEYS:
                    val = c[k]
                    val[mask5[k]] = leaf[k]
                    c[k] = val
                opt.zero_grad()
                depth, var, color = renderer.render_batch_ray(c, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
                E.losses.rgbd_loss(depth, color, rays['gt_depth'], rays['gt_color'], 0.2).backward()
                opt.step()
                for k in KEYS:
                    val = c[k].detach()
                    val[mask5[k]] = leaf[k].clone().detach()
                    c[k] = val
            assert all(c[k].is_contiguous(memory_format=torch.channels_last_3d) == (layout == 'channels_last_3d') for k in KEYS)
            res[layout] = {k: c[k].contiguous().clone() for k in KEYS}
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
    for k in KEYS:
        a, b = res['contiguous'][k], res['channels_last_3d'][k]
        assert float((a - grids[k]).abs().max()) > 1e-3                      # (something was optimised)
        assert float((a - b).abs().max()) <= 1e-5 + 2e-4 * 0.01 * 2, k


def test_step_plans_serve_the_plain_calls_and_follow_changes(monkeypatch):
    """functional._PlanFn (one library call per direction from a cached enslam_step_plan) against _RenderFn (ENSLAM_STEP_PLANS=0) on
    the same call: identical outputs, gradients within float-atomic ordering; the cache follows what a loop changes -- gradient flags
    of the decoders (a tracker freezes them), a replaced grid tensor, a parameter's storage, the batch size, the layout."""
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from tests.hip_util import DEV, as_layout, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()

    def run(cg, n=64, params_grad=True):
        for p in model.parameters():
            p.grad = None
            p.requires_grad_(params_grad)
        ro, rd = rays['rays_o'][:n].clone().requires_grad_(True), rays['rays_d'][:n].clone().requires_grad_(True)
        for g in cg.values():
            g.grad = None
        d, v, c = renderer.render_batch_ray(cg, model, rd, ro, DEV, 'color', gt_depth=rays['gt_depth'][:n])
        (d.sum() + 0.3 * c.sum().double() + 0.1 * v.sum()).backward()
        return ([d.detach(), v.detach(), c.detach()],
                [ro.grad, rd.grad] + [cg[k].grad for k in KEYS] + [p.grad for p in model.color_decoder.parameters() if p.grad is not None])

    try:
        for layout in ('contiguous', 'channels_last_3d'):
            cg = {k: as_layout(v, layout).requires_grad_(True) for k, v in grids.items()}
            monkeypatch.setattr(EF, 'STEP_PLANS', False)
            o0, g0 = run(cg)
            monkeypatch.setattr(EF, 'STEP_PLANS', True)
            before = dict(EF.plan_stats)
            o1, g1 = run(cg)
            assert EF.plan_stats['built'] + EF.plan_stats['hits'] == before['built'] + before['hits'] + 1      # the plan route ran
            for a, b in zip(o0, o1):
                assert torch.equal(a, b)
            assert len(g0) == len(g1) > 20
            for a, b in zip(g0, g1):
                assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-30)
            if layout == 'channels_last_3d':
                assert all(cg[k].grad.is_contiguous(memory_format=torch.channels_last_3d) for k in KEYS)
            b0 = EF.plan_stats['built']
            run(cg)
            assert EF.plan_stats['built'] == b0                                  # same loop step again: found, not rebuilt
            o2, g2 = run(cg, params_grad=False)                                  # decoders frozen: another plan (light workspace)
            assert EF.plan_stats['built'] == b0 + 1 and torch.equal(o2[0], o1[0])
            assert float((g2[1] - g1[1]).abs().max()) <= 2e-5 * float(g1[1].abs().max())
            run(cg, n=48)                                                        # another batch size
            assert EF.plan_stats['built'] == b0 + 2
            cg2 = dict(cg)
            cg2['grid_fine'] = (cg['grid_fine'].detach() * 1.5).contiguous(memory_format=torch.channels_last_3d if layout != 'contiguous'
                                                                          else torch.contiguous_format).requires_grad_(True)
            o3, g3 = run(cg2)                                                    # a replaced grid tensor
            assert EF.plan_stats['built'] == b0 + 3 and not torch.equal(o3[0], o1[0])
            w = model.color_decoder.pts_linears[1].weight
            with torch.no_grad():
                w.data = (w.data * 0.5).clone()                                  # same parameter object, new storage
            o4, g4 = run(cg)
            assert EF.plan_stats['built'] == b0 + 4 and not torch.equal(o4[2], o1[2])
            monkeypatch.setattr(EF, 'STEP_PLANS', False)
            o5, g5 = run(cg)
            monkeypatch.setattr(EF, 'STEP_PLANS', True)
            assert torch.equal(o4[0], o5[0]) and torch.equal(o4[2], o5[2])
            with torch.no_grad():
                w.data = (w.data * 2.0).clone()
    finally:
        for p in model.parameters():
            p.requires_grad_(True)


@pytest.mark.parametrize("stage", ['middle', 'fine', 'color'])
@pytest.mark.parametrize("pattern", ['rays', 'grids', 'params', 'grids+params', 'all', 'fused-loss'])
def test_step_plan_route_equals_the_function_route(monkeypatch, stage, pattern):
    """every gradient pattern a caller can ask for (pose only: the tracker; grids only: mapper stages with fixed decoders; decoders
    only; everything), each stage, plain outputs or the fused mapper loss: the step-plan route against _RenderFn"""
    import evennicer_slam_amd.functional as EF
    from tests.hip_util import DEV, as_layout, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    want_rays = pattern in ('rays', 'all', 'fused-loss')
    want_grids = pattern in ('grids', 'grids+params', 'all', 'fused-loss')
    want_params = pattern in ('params', 'grids+params', 'all', 'fused-loss')

    def run():
        for p in model.parameters():
            p.grad = None
            p.requires_grad_(want_params)
        cg = {k: as_layout(v, 'channels_last_3d' if k != 'grid_color' else 'contiguous').requires_grad_(want_grids) for k, v in grids.items()}
        ro, rd = rays['rays_o'].clone().requires_grad_(want_rays), rays['rays_d'].clone().requires_grad_(want_rays)
        if pattern == 'fused-loss':
            loss, d, v, c = renderer.render_batch_ray_rgbd_loss(cg, model, rd, ro, DEV, stage, rays['gt_depth'], rays['gt_color'], 0.2)
        else:
            d, v, c = renderer.render_batch_ray(cg, model, rd, ro, DEV, stage, gt_depth=rays['gt_depth'])
            loss = d.sum() + 0.1 * v.sum() + (0.3 * c.sum().double() if stage == 'color' else 0.0)
        loss.backward()
        gs = [ro.grad, rd.grad] + [cg[k].grad for k in KEYS] + [p.grad for p in model.parameters()]
        return [d.detach(), v.detach(), c.detach(), loss.detach()], gs

    try:
        monkeypatch.setattr(EF, 'STEP_PLANS', False)
        o0, g0 = run()
        monkeypatch.setattr(EF, 'STEP_PLANS', True)
        n0 = EF.plan_stats['built'] + EF.plan_stats['hits']
        o1, g1 = run()
        assert EF.plan_stats['built'] + EF.plan_stats['hits'] == n0 + 1
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
    for a, b in zip(o0[:3], o1[:3]):
        assert torch.equal(a, b)
    assert abs(float(o0[3]) - float(o1[3])) <= 1e-12 * max(abs(float(o0[3])), 1.0)
    got = 0
    for a, b in zip(g0, g1):
        assert (a is None) == (b is None)
        if a is not None:
            got += 1
            assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-30)
    assert got > 0
