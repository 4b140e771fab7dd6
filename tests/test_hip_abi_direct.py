"""-m gpu: the stand-alone entry points of the C ABI that the Python product path no longer calls (it uses their merged
forms): enslam_render_fwd -> enslam_render_bwd (= composite_bwd + decoder_bwd + ray_grad_bwd), enslam_unpack_mlp_grads,
enslam_grid_from_voxel_major, enslam_mark_blocks -- driven by hand, checked against the autograd path."""
import ctypes

import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_workspace", [True, False])
def test_render_fwd_then_render_bwd_by_hand(with_workspace):
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from tests.hip_util import DEV, tiny_on_gpu
    L = E._lib
    lib = L.lib()
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    stage, kinds = 'color', (1, 2, 3)
    ro, rd, gd = rays['rays_o'].contiguous(), rays['rays_d'].contiguous(), rays['gt_depth'].contiguous()
    N, S = ro.shape[0], 48
    P = EF._ptr
    st = EF._stream()
    # reference: the product path
    cg = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    ro_a, rd_a = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
    for p in model.parameters():
        p.grad = None
    depth_a, var_a, color_a = renderer.render_batch_ray(cg, model, rd_a, ro_a, DEV, stage, gt_depth=gd)
    g = torch.Generator().manual_seed(7)
    gD = torch.randn(N, generator=g).double().to(DEV)
    gV = (0.1 * torch.randn(N, generator=g)).double().to(DEV)
    gC = torch.randn(N, 3, generator=g).to(DEV)
    ((depth_a * gD).sum() + (var_a * gV).sum() + (color_a * gC).sum().double()).backward()

    # by hand through the ABI
    z = EF.sample_rays(ro, rd, gd, bound, 32, 16)
    grids_vm = {k: EF._grid_cache.get(grids[L.GRID_NAMES[k]]) for k in kinds}
    dims = {k: tuple(grids[L.GRID_NAMES[k]].shape[2:]) for k in kinds}
    decs = {k: getattr(model, L.MLP_NAMES[k]) for k in kinds}
    packed = {k: EF.packed_decoder(decs[k], k) for k in kinds}
    sc = EF._scene_struct(stage, EF.bound6(bound), EF.bound6(bound * 2), grids_vm, dims, packed)
    depth = torch.empty(N, dtype=torch.float64, device=DEV)
    var = torch.empty(N, dtype=torch.float64, device=DEV)
    rgb = torch.empty((N, 3), dtype=torch.float32, device=DEV)
    raw = torch.empty((N * S, 4), dtype=torch.float32, device=DEV)
    act = torch.empty(lib.enslam_activation_floats(3, N, S, 0), dtype=torch.float32, device=DEV) if with_workspace else None
    L.check(lib.enslam_render_fwd(3, N, S, P(ro), P(rd), P(z), ctypes.byref(sc), P(depth), P(var), P(rgb), P(raw), P(act), 0, st),
            "enslam_render_fwd")
    assert torch.equal(depth, depth_a.detach()) and torch.equal(rgb, color_a.detach())
    gg = (L.Grid * 4)()
    gpk = (ctypes.c_void_p * 4)()
    acc_g, acc_p = {}, {}
    for k in kinds:
        V = dims[k][0] * dims[k][1] * dims[k][2]
        acc_g[k] = torch.zeros((V, 32), dtype=torch.float32, device=DEV)
        acc_p[k] = torch.zeros(lib.enslam_packed_grad_floats(k), dtype=torch.float32, device=DEV)
        gg[k].data, (gg[k].D, gg[k].H, gg[k].W) = acc_g[k].data_ptr(), dims[k]
        gpk[k] = acc_p[k].data_ptr()
    g_ro = torch.zeros((N, 3), dtype=torch.float32, device=DEV)
    g_rd = torch.zeros((N, 3), dtype=torch.float32, device=DEV)
    d_raw = torch.empty((N * S, 4), dtype=torch.float32, device=DEV)
    dgw = torch.empty(lib.enslam_grid_handoff_floats(3, N, S), dtype=torch.float32, device=DEV) if with_workspace else None
    L.check(lib.enslam_render_bwd(3, N, S, P(ro), P(rd), P(z), ctypes.byref(sc), P(raw), P(depth), P(gD), P(gV), P(gC), gg, gpk,
                                  P(g_ro), P(g_rd), P(d_raw), P(act), 0, P(dgw), st), "enslam_render_bwd")
    torch.cuda.synchronize()
    assert rel_err(g_ro.cpu().numpy(), ro_a.grad.cpu().numpy()) <= 1e-4
    assert rel_err(g_rd.cpu().numpy(), rd_a.grad.cpu().numpy()) <= 1e-4
    for k in kinds:
        name = L.GRID_NAMES[k]
        dense = torch.empty_like(grids[name])
        L.check(lib.enslam_grid_from_voxel_major(P(acc_g[k]), P(dense), acc_g[k].shape[0], st), "grid_from_voxel_major")
        assert rel_err(dense.cpu().numpy(), cg[name].grad.cpu().numpy()) <= 1e-4, name
        ps = EF.decoder_params(decs[k], k)
        outs = [torch.zeros_like(p) for p in ps]
        L.check(lib.enslam_unpack_mlp_grads(k, P(acc_p[k]), ctypes.byref(EF._fill_params_struct(k, outs)), st),
                "enslam_unpack_mlp_grads")
        torch.cuda.synchronize()
        for p, o in zip(ps, outs):
            assert rel_err(o.cpu().numpy(), p.grad.cpu().numpy()) <= 1e-4 or float(p.grad.abs().max()) == 0.0
    for p in model.parameters():
        p.grad = None


def test_mark_blocks_covers_the_samplers_marks():
    """enslam_mark_blocks (stand-alone) flags the same 64-voxel blocks as the marking inside the sampler."""
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd import parallel as PAR
    from tests.hip_util import DEV, tiny_on_gpu
    L = E._lib
    lib = L.lib()
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    ro, rd, gd = rays['rays_o'].contiguous(), rays['rays_d'].contiguous(), rays['gt_depth'].contiguous()
    z = EF.sample_rays(ro, rd, gd, bound, 32, 16)
    in_sampler = PAR.batch_block_flags(renderer, grids, model, ro, rd, gd, 'color')
    msc = L.Scene()
    msc.bound, msc.coarse_bound = EF.bound6(bound), EF.bound6(bound * 2)
    fptr = (ctypes.c_void_p * 4)()
    own = {}
    for k in (1, 2, 3):
        g = grids[L.GRID_NAMES[k]]
        msc.grids[k].D, msc.grids[k].H, msc.grids[k].W = (int(x) for x in g.shape[2:])
        own[k] = torch.zeros((g.shape[2] * g.shape[3] * g.shape[4] + 63) // 64, dtype=torch.uint8, device=DEV)
        fptr[k] = own[k].data_ptr()
    L.check(lib.enslam_mark_blocks(3, ro.shape[0], 48, EF._ptr(ro), EF._ptr(rd), EF._ptr(z), ctypes.byref(msc), fptr, EF._stream()),
            "enslam_mark_blocks")
    torch.cuda.synchronize()
    for k in (1, 2, 3):
        assert torch.equal(own[k], in_sampler[id(grids[L.GRID_NAMES[k]])]), L.GRID_NAMES[k]
        assert int(own[k].sum()) > 0


def test_step_finish_rays_prev_persistent_destination():
    """enslam_step_finish_rays_prev by hand: three calls into the SAME dense gradient with different block flags give, each time,
    exactly what enslam_step_finish writes into a fresh tensor; the flags move to `prev` (the live flags end all-zero); blocks
    untouched twice in a row are not written at all (a sentinel planted there survives)."""
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from tests.hip_util import DEV
    L = E._lib
    lib = L.lib()
    P, st = EF._ptr, EF._stream()
    D, H, W = 12, 20, 28                                       # 6720 voxels = 105 blocks of 64
    V = D * H * W
    nblk = (V + 63) // 64
    gen = torch.Generator().manual_seed(5)
    vm = torch.randn((V, 32), generator=gen).to(DEV)           # a voxel-major gradient accumulator (values everywhere)
    dense = torch.zeros((1, 32, D, H, W), dtype=torch.float32, device=DEV)
    prev = torch.zeros(nblk, dtype=torch.uint8, device=DEV)
    arr = lambda *ptrs: (ctypes.c_void_p * len(ptrs))(*ptrs)
    nvox = (ctypes.c_int64 * 1)(V)
    masks = [torch.rand(nblk, generator=gen) < 0.3 for _ in range(3)]
    masks[2][:8] = False
    masks[1][:8] = False                                       # blocks 0..7: untouched in calls 2 and 3
    for it, m in enumerate(masks):
        flags = m.to(torch.uint8).to(DEV)
        want = torch.empty_like(dense)
        L.check(lib.enslam_step_finish(1, arr(vm.data_ptr()), arr(want.data_ptr()), nvox, arr(flags.data_ptr()), 0, None, None, None, st),
                "enslam_step_finish")
        if it == 2:
            dense.view(32, V)[:, :8 * 64] = 123.0              # untouched now and in the previous call: must not be written
        live = flags.clone()
        L.check(lib.enslam_step_finish_rays_prev(1, arr(vm.data_ptr()), arr(dense.data_ptr()), nvox, arr(live.data_ptr()),
                                                 arr(prev.data_ptr()), 0, None, None, None, 3, 0, 48, None, None, None, None, None,
                                                 None, None, None, None, st), "enslam_step_finish_rays_prev")
        torch.cuda.synchronize()
        assert torch.equal(prev, flags) and int(live.sum()) == 0
        got = dense.clone()
        if it == 2:
            assert bool((got.view(32, V)[:, :8 * 64] == 123.0).all())
            got.view(32, V)[:, :8 * 64] = 0.0
        assert torch.equal(got, want), it
    # prev without flags is refused
    assert lib.enslam_step_finish_rays_prev(1, arr(vm.data_ptr()), arr(dense.data_ptr()), nvox, None, arr(prev.data_ptr()), 0, None,
                                            None, None, 3, 0, 48, None, None, None, None, None, None, None, None, None, st) == -1
