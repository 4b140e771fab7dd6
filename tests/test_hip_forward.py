"""GPU parity, forward: HIP path (through the C ABI) vs the reference-generated goldens.

Bars (BASELINE.json north_star): sample distances / points / masks / voxel indices bit-exact;
rendered depth, uncertainty, colour within 1e-4 relative."""
import numpy as np
import pytest
import torch

from tests.util import GRID_KEYS, STAGES, load, rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-4     # north_star tolerance for rendered depth / colour


def _close(a, b, rtol=RTOL, atol_frac=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    atol = atol_frac * max(np.abs(b).max(), 1e-30)
    bad = np.abs(a - b) > rtol * np.abs(b) + atol
    return not bad.any(), float(np.abs(a - b).max()), float(np.abs(b).max())


@pytest.fixture(scope="module")
def tiny():
    from tests.hip_util import tiny_on_gpu
    return tiny_on_gpu()


def test_library_is_native_and_loaded():
    import evennicer_slam_amd as E
    lib = E._lib.lib()
    assert lib.enslam_arch() == b"gfx950"
    assert torch.cuda.is_available()


@pytest.mark.parametrize("stage", STAGES)
def test_sampling_points_mask_bit_exact(tiny, stage):
    import evennicer_slam_amd.functional as EF
    s, bound, model, grids, rays, renderer = tiny
    g = load("tiny_" + stage)
    gd = None if stage == 'coarse' else rays['gt_depth']
    z = EF.sample_rays(rays['rays_o'], rays['rays_d'], gd, bound, 32, 16)
    assert z.dtype == torch.float64 and tuple(z.shape) == g["z_vals"].shape
    assert np.array_equal(z.cpu().numpy(), g["z_vals"])                        # bit-exact float64
    pts, mask = EF.ray_points(rays['rays_o'], rays['rays_d'], z, bound)
    assert np.array_equal(pts.cpu().numpy(), g["pts"])                         # bit-exact float64
    assert np.array_equal(mask.cpu().numpy(), g["mask"])


@pytest.mark.parametrize("stage", STAGES)
def test_voxel_indices_bit_exact(tiny, stage):
    import evennicer_slam_amd.functional as EF
    s, bound, model, grids, rays, renderer = tiny
    g = load("tiny_" + stage)
    pts = torch.from_numpy(g["pts"]).cuda()
    used = {'coarse': ['grid_coarse'], 'middle': ['grid_middle'], 'fine': ['grid_fine', 'grid_middle'],
            'color': ['grid_fine', 'grid_middle', 'grid_color']}[stage]
    for key in used:
        b = bound * 2 if key == 'grid_coarse' else bound
        ix, iy, iz, fx, fy, fz = [t.cpu().numpy() for t in EF.voxel_index(pts, b, tuple(grids[key].shape[2:]))]
        for name, got in (("ix", ix), ("iy", iy), ("iz", iz), ("fx", fx), ("fy", fy), ("fz", fz)):
            assert np.array_equal(got, g[f"vox_{key}_{name}"]), (key, name)


def test_grid_layout_round_trip():
    import ctypes
    import evennicer_slam_amd as E
    lib = E._lib.lib()
    for V in (1, 63, 64, 65, 7 * 8 * 11):
        src = torch.randn(32, V, device='cuda')
        vm = torch.empty(V, 32, device='cuda')
        back = torch.empty_like(src)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.enslam_grid_to_voxel_major(src.data_ptr(), vm.data_ptr(), V, st) == 0
        assert lib.enslam_grid_from_voxel_major(vm.data_ptr(), back.data_ptr(), V, st) == 0
        assert torch.equal(vm, src.t().contiguous())
        assert torch.equal(back, src)


@pytest.mark.parametrize("stage", STAGES)
def test_render_forward_matches_golden(tiny, stage):
    s, bound, model, grids, rays, renderer = tiny
    g = load("tiny_" + stage)
    with torch.no_grad():
        depth, var, color = renderer.render_batch_ray(grids, model, rays['rays_d'], rays['rays_o'], 'cuda:0', stage,
                                                      gt_depth=rays['gt_depth'])
    assert depth.dtype == torch.float64 and var.dtype == torch.float64 and color.dtype == torch.float32
    for name, got, ref in (("depth", depth, g["depth"]), ("var", var, g["var"]), ("color", color, g["color"])):
        ok, err, mag = _close(got.cpu().numpy(), ref)
        assert ok, f"{stage} {name}: max abs err {err:.3e} vs magnitude {mag:.3e}"


def test_eval_points_matches_golden(tiny):
    s, bound, model, grids, rays, renderer = tiny
    g = load("tiny_eval_points")
    p = torch.from_numpy(g["p"]).cuda()
    for stage in STAGES:
        with torch.no_grad():
            raw = renderer.eval_points(p, model, grids, stage, 'cuda:0')
        assert tuple(raw.shape) == (200, 4)
        ok, err, mag = _close(raw.cpu().numpy(), g["raw_" + stage])
        assert ok, f"{stage}: {err:.3e} / {mag:.3e}"
        assert float(raw[0, 3]) == 100.0 and float(raw[1, 3]) == 100.0
        # NICE.forward: same values without the bound mask
        with torch.no_grad():
            raw2 = model(p[None], c_grid=grids, stage=stage)
        inside = (raw[:, 3] != 100.0)
        assert torch.equal(raw2[inside], raw[inside])
        assert float(raw2[0, 3]) != 100.0


def test_room0_coarse200_config0():
    """BASELINE configs[0] shape on the GPU path: room0 coarse grid, 200 rays x 32 samples."""
    from tests.hip_util import model_from_state, renderer_for
    g = load("room0_coarse200")
    bound = torch.from_numpy(g["bound"].copy())
    model = model_from_state(g, bound)
    grids = {"grid_coarse": torch.from_numpy(g["grid_coarse"].copy()).cuda()}
    renderer = renderer_for(bound, cam=(680, 1200, 600.0, 600.0, 599.5, 339.5))
    ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
    with torch.no_grad():
        depth, var, color = renderer.render_batch_ray(grids, model, rd, ro, 'cuda:0', 'coarse', gt_depth=None)
    ok, err, mag = _close(depth.cpu().numpy(), g["depth"])
    assert ok, (err, mag)
    ok, err, mag = _close(var.cpu().numpy(), g["var"])
    assert ok, (err, mag)
    assert float(color.abs().max()) == 0.0


def test_fourier_embedding_large_arguments():
    """The embedding's own sin (forward: float32 Cody-Waite reduction + minimax polynomials) and cos (backward: Cody-Waite
    reduction by 2 pi + the hardware's v_cos_f32), csrc/common.hpp, against float64 libm on the argument range of p @ B
    (B ~ 25 * randn, |p| up to ~10: a few thousand), at quadrant boundaries and on huge / tiny / special arguments.  Bars: sin
    2e-7 absolute (measured 9e-8: its values decide ReLU masks, it stays as close to the reference's torch.sin as float32
    allows), cos 1e-6 (measured 3.5e-7; it only scales d_arg).  The outputs' 1e-4 budget is spent on the float32 rounding of the
    argument itself, 1.2e-4 at |x| = 2000."""
    import evennicer_slam_amd.functional as EF
    g = torch.Generator().manual_seed(3)
    x = torch.cat([
        (torch.rand(200000, generator=g) - 0.5) * 8000.0,                           # |x| <= 4000
        torch.randn(50000, generator=g) * 25.0 * 3.0,                               # typical p @ B
        torch.arange(-2600, 2600, dtype=torch.float64).mul(np.pi / 4).float(),      # every multiple of pi/4 to |x| ~ 2000
        torch.arange(-2600, 2600, dtype=torch.float64).mul(np.pi / 4).float() * (1 + 6e-8),
        torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1e-8, 0.5, -0.5, 2000.0, -2000.0, 1999.9999, 4000.0, 3e4, -3e4]),
    ]).cuda()
    s, c = EF.fourier_sincos(x)
    xd = x.double().cpu()
    es = (s.double().cpu() - torch.sin(xd)).abs()
    ec = (c.double().cpu() - torch.cos(xd)).abs()
    assert float(es.max()) < 2e-7, float(es.max())
    assert float(ec.max()) < 1e-6, float(ec.max())
    assert float(s[x == 0].abs().max()) == 0.0 and float((c[x == 0] - 1).abs().max()) == 0.0
    # the pair is consistent: sin^2 + cos^2 = 1 to float32 rounding
    assert float((s.double() ** 2 + c.double() ** 2 - 1).abs().max()) < 2e-6
