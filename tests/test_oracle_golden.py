"""Pin the CPU oracle (oracle/render_oracle.py) to the reference-generated goldens.

Bars: sample distances, points, masks and voxel indices bit-exact; rendered
outputs <= 1e-6 relative; gradients <= 1e-5 relative to the tensor's max."""
import numpy as np
import pytest
import torch

from oracle import render_oracle as R
from tests.util import GRID_KEYS, STAGES, load, rel_err, tiny_scene


def test_scene_bounds_and_grid_shapes():
    b = load("bounds")
    for tag in ("room0", "office0", "recording4"):
        bound = R.scene_bound(b[tag + "_cfg_bound"], float(b[tag + "_scale"]), 0.32)
        assert np.array_equal(bound.numpy(), b[tag + "_bound"]), tag          # bit-exact float64
        shp = R.grid_shapes(bound, dict(coarse=2, middle=0.32, fine=0.16, color=0.16))
        assert [shp[k] for k in GRID_KEYS] == b[tag + "_shapes"].tolist()
    assert float(R.scene_bound(b["room0_cfg_bound"])[0, 1]) == 8.94000015258789


def test_get_samples_rays_bit_exact():
    r = load("rays")
    H, W, fx, fy, cx, cy = r["cam"]
    H, W = int(H), int(W)
    depth = torch.from_numpy(r["depth_img"])
    color = torch.from_numpy(r["color_img"])
    c2w = torch.from_numpy(r["c2w"])
    for tag in ("full", "edge"):
        H0, H1, W0, W1, n = [int(v) for v in r[tag + "_args"]]
        torch.manual_seed(int(r["seed"]))
        ro, rd, d, c = R.sample_pixels(H0, H1, W0, W1, n, c2w, depth, color, fx, fy, cx, cy)
        assert np.array_equal(ro.numpy(), r[tag + "_rays_o"])
        assert np.array_equal(rd.numpy(), r[tag + "_rays_d"])
        assert np.array_equal(d.numpy(), r[tag + "_depth"])
        assert np.array_equal(c.numpy(), r[tag + "_color"])
        # explicit indices reproduce the same rays (RNG-free entry)
        ro2, rd2, _, _ = R.sample_pixels(H0, H1, W0, W1, n, c2w, depth, color, fx, fy, cx, cy,
                                         idx=torch.from_numpy(r[tag + "_idx"]))
        assert np.array_equal(rd2.numpy(), r[tag + "_rays_d"])
    ro, rd = R.image_rays(H, W, int(H * 0.15), int(W * 0.15), c2w, fx, fy, cx, cy)
    assert np.array_equal(rd.numpy(), r["rescale_rays_d"])
    assert np.array_equal(ro.numpy(), r["rescale_rays_o"])
    ro, rd = R.image_rays(H, W, H, W, c2w, fx, fy, cx, cy)
    assert np.array_equal(rd.numpy()[::7, ::11], r["img_rays_d"])


def test_pose_gradient_through_ray_generation():
    r = load("rays")
    H, W, fx, fy, cx, cy = r["cam"]
    c2w = torch.from_numpy(r["c2w"]).requires_grad_(True)
    ro, rd, _, _ = R.sample_pixels(0, int(H), 0, int(W), 100, c2w, torch.from_numpy(r["depth_img"]),
                                   torch.from_numpy(r["color_img"]), fx, fy, cx, cy,
                                   idx=torch.from_numpy(r["full_idx"]))
    ((ro * torch.from_numpy(r["cot_o"])).sum() + (rd * torch.from_numpy(r["cot_d"])).sum()).backward()
    assert rel_err(c2w.grad.numpy(), r["g_c2w"]) < 1e-6


@pytest.mark.parametrize("stage", STAGES)
def test_tiny_forward_intermediates(stage):
    params, grids, bound, s = tiny_scene()
    g = load("tiny_" + stage)
    ro, rd, gd = [torch.from_numpy(s[k]) for k in ("rays_o", "rays_d", "gt_depth")]
    depth, var, rgb, aux = R.render_batch_ray(params, grids, rd, ro, stage, bound, gt_depth=gd, return_aux=True)
    assert np.array_equal(aux["z_vals"].numpy(), g["z_vals"])               # bit-exact float64
    assert np.array_equal(aux["pts"].numpy(), g["pts"])                     # bit-exact float64
    assert np.array_equal(R.inside_bound(aux["pts"], bound).numpy(), g["mask"])
    assert g["mask"].sum() < g["mask"].size                                 # fixture has out-of-bound samples
    assert rel_err(aux["raw"].reshape(-1, 4).numpy(), g["raw"]) < 1e-6
    assert rel_err(aux["weights"].numpy(), g["weights"]) < 1e-6
    assert rel_err(depth.numpy(), g["depth"]) < 1e-6
    assert rel_err(var.numpy(), g["var"]) < 1e-6
    assert rel_err(rgb.numpy(), g["color"]) < 1e-6
    assert depth.dtype == torch.float64 and var.dtype == torch.float64 and rgb.dtype == torch.float32


@pytest.mark.parametrize("stage", STAGES)
def test_tiny_voxel_indices_bit_exact(stage):
    params, grids, bound, s = tiny_scene()
    g = load("tiny_" + stage)
    pts = torch.from_numpy(g["pts"])
    for key in R.STAGE_GRIDS[stage]:
        b = bound * 2 if key == "grid_coarse" else bound
        (ix, iy, iz), (fx, fy, fz), _ = R.voxel_coords(pts, b, grids[key].shape[2:])
        for name, got in (("ix", ix), ("iy", iy), ("iz", iz)):
            assert np.array_equal(got.numpy().astype(np.int32), g[f"vox_{key}_{name}"])
        for name, got in (("fx", fx), ("fy", fy), ("fz", fz)):
            assert np.array_equal(got.numpy(), g[f"vox_{key}_{name}"])
        # explicit 8-corner form == F.grid_sample port
        a = R.trilinear_explicit(grids[key], pts, b)
        bb = R.trilinear(grids[key], pts, b)
        assert rel_err(a.numpy(), bb.numpy()) < 1e-6


def _backward(stage, cot=None, mapper=None):
    params, grids, bound, s = tiny_scene()
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    grids = {k: v.requires_grad_(True) for k, v in grids.items()}
    ro = torch.from_numpy(s["rays_o"]).requires_grad_(True)
    rd = torch.from_numpy(s["rays_d"]).requires_grad_(True)
    gd = torch.from_numpy(s["gt_depth"])
    depth, var, rgb = R.render_batch_ray(params, grids, rd, ro, stage, bound, gt_depth=gd)
    if mapper is None:
        loss = (depth * cot[0]).sum() + (var * cot[1]).sum() + (rgb * cot[2]).sum()
    else:
        loss = R.mapper_loss(depth, rgb, gd, torch.from_numpy(s["gt_color"]), stage)
    loss.backward()
    return params, grids, ro, rd, loss


def _check_grads(g, params, grids, ro, rd, tol=1e-5):
    n = 0
    assert rel_err(ro.grad.numpy(), g["g_rays_o"]) < tol
    assert rel_err(rd.grad.numpy(), g["g_rays_d"]) < tol
    for k in GRID_KEYS:
        if "g_" + k in g:
            assert grids[k].grad is not None, k
            assert rel_err(grids[k].grad.numpy(), g["g_" + k]) < tol, k
            n += 1
        else:
            assert grids[k].grad is None or float(grids[k].grad.abs().max()) == 0.0, k
    for k, v in params.items():
        if "gp_" + k in g:
            assert v.grad is not None, k
            assert rel_err(v.grad.numpy(), g["gp_" + k]) < tol, k
            n += 1
        else:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    return n


@pytest.mark.parametrize("stage", STAGES)
def test_tiny_backward_random_cotangents(stage):
    g = load("tiny_" + stage)
    cot = [torch.from_numpy(g[k]) for k in ("cot_depth", "cot_var", "cot_color")]
    params, grids, ro, rd, loss = _backward(stage, cot=cot)
    assert abs(loss.item() - float(g["loss"])) <= 1e-6 * max(1.0, abs(float(g["loss"])))
    assert _check_grads(g, params, grids, ro, rd) >= 2


def test_tiny_backward_mapper_loss():
    g = load("tiny_color_mapperloss")
    params, grids, ro, rd, loss = _backward("color", mapper=True)
    assert abs(loss.item() - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    _check_grads(g, params, grids, ro, rd)


def test_eval_points_masking():
    params, grids, bound, s = tiny_scene()
    g = load("tiny_eval_points")
    p = torch.from_numpy(g["p"])
    for stage in STAGES:
        raw = R.eval_points(params, grids, p, stage, bound)
        assert rel_err(raw.numpy(), g["raw_" + stage]) < 1e-6
        assert float(raw[0, 3]) == 100.0 and float(raw[1, 3]) == 100.0      # on the faces: strict compare


def test_room0_coarse200_config0():
    """BASELINE configs[0]: room0 coarse grid only, 200 rays x 32 samples, CPU."""
    g = load("room0_coarse200")
    params = {k[3:]: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in g.items() if k.startswith("sd_")}
    grids = {"grid_coarse": torch.from_numpy(g["grid_coarse"].copy()).requires_grad_(True)}
    bound = torch.from_numpy(g["bound"])
    ro = torch.from_numpy(g["rays_o"]).requires_grad_(True)
    rd = torch.from_numpy(g["rays_d"]).requires_grad_(True)
    depth, var, rgb, aux = R.render_batch_ray(params, grids, rd, ro, "coarse", bound, gt_depth=None, return_aux=True)
    assert aux["z_vals"].shape == (200, 32)
    assert np.array_equal(aux["z_vals"].numpy(), g["z_vals"])
    assert rel_err(depth.detach().numpy(), g["depth"]) < 1e-6
    assert rel_err(var.detach().numpy(), g["var"]) < 1e-6
    loss = R.mapper_loss(depth, rgb, torch.from_numpy(g["gt_depth"]), torch.from_numpy(g["gt_color"]), "coarse")
    assert abs(loss.item() - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    loss.backward()
    assert rel_err(grids["grid_coarse"].grad.numpy(), g["g_grid_coarse"]) < 1e-5
    assert rel_err(rd.grad.numpy(), g["g_rays_d"]) < 1e-5
    for k, v in params.items():
        if "gp_" + k in g:
            assert rel_err(v.grad.numpy(), g["gp_" + k]) < 1e-5, k


def test_oracle_hierarchical_sampling_matches_reference():
    """N_importance = 8 (Renderer.py:182-197 + sample_pdf, common.py:19-63): the oracle's second pass against the
    reference fixture -- merged sample distances, outputs and gradients."""
    import torch
    from oracle import render_oracle as R
    from tests.util import tiny_scene
    g = load("tiny_importance")
    for stage in ('color', 'middle', 'coarse'):
        params, grids, bound, s = tiny_scene()
        params = {k: v.requires_grad_(True) for k, v in params.items()}
        grids = {k: v.requires_grad_(True) for k, v in grids.items()}
        ro = torch.from_numpy(s['rays_o']).requires_grad_(True)
        rd = torch.from_numpy(s['rays_d']).requires_grad_(True)
        gd, gc = torch.from_numpy(s['gt_depth']), torch.from_numpy(s['gt_color'])
        depth, var, rgb, aux = R.render_batch_ray(params, grids, rd, ro, stage, bound, gt_depth=None if stage == 'coarse' else gd,
                                                  return_aux=True, n_importance=int(g['N_importance']))
        assert aux['z_vals'].shape == g[f'{stage}_z_vals'].shape
        assert np.abs(aux['z_vals'].detach().numpy() - g[f'{stage}_z_vals']).max() <= 1e-9
        for name, got in (('depth', depth), ('var', var), ('color', rgb)):
            assert rel_err(got.detach().numpy(), g[f'{stage}_{name}']) <= 1e-6, (stage, name)
        if stage == 'coarse':
            loss = (depth * torch.from_numpy(g['cot_depth'])).sum() + (var * torch.from_numpy(g['cot_var'])).sum() + \
                   (rgb * torch.from_numpy(g['cot_color'])).sum()
        else:
            loss = R.mapper_loss(depth, rgb, gd, gc, stage)
        loss.backward()
        assert abs(loss.item() - float(g[f'{stage}_loss'])) <= 1e-6 * max(1.0, abs(float(g[f'{stage}_loss'])))
        assert rel_err(rd.grad.numpy(), g[f'{stage}_g_rays_d']) <= 1e-5
        key = {'color': 'grid_color', 'middle': 'grid_middle', 'coarse': 'grid_coarse'}[stage]
        assert rel_err(grids[key].grad.numpy(), g[f'{stage}_g_{key}']) <= 1e-5
