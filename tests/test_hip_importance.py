"""Hierarchical sampling on the HIP path: Renderer.render_batch_ray with rendering.N_importance = 8 against
the reference fixture (tests/golden/tiny_importance.npz): merged sample distances, outputs, and every gradient of the
second pass (56 = 32 + 16 + 8 samples per ray, padded to 64 inside; 40 -> 48 for the coarse stage)."""
import numpy as np
import pytest
import torch

from tests.util import GRID_KEYS, load, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup():
    from tests.hip_util import cfg_like, model_from_state, renderer_for
    s = load('tiny_scene')
    bound = torch.from_numpy(s['bound'].copy())
    model = model_from_state(s, bound)
    cfg = cfg_like()
    cfg['rendering']['N_importance'] = 8
    renderer = renderer_for(bound, cfg=cfg)
    grids = {k: torch.from_numpy(s[k].copy()).to(DEV) for k in GRID_KEYS}
    rays = {k: torch.from_numpy(s[k].copy()).to(DEV) for k in ('rays_o', 'rays_d', 'gt_depth', 'gt_color')}
    return s, bound, model, grids, rays, renderer


@pytest.mark.parametrize("stage", ['color', 'middle', 'coarse'])
def test_hierarchical_render_matches_reference(stage):
    import evennicer_slam_amd as E
    s, bound, model, grids, rays, renderer = _setup()
    g = load('tiny_importance')
    assert renderer.N_importance == int(g['N_importance'])
    for p in model.parameters():
        p.grad = None
    cg = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    ro = rays['rays_o'].clone().requires_grad_(True)
    rd = rays['rays_d'].clone().requires_grad_(True)
    gd = None if stage == 'coarse' else rays['gt_depth']
    depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, DEV, stage, gt_depth=gd)
    assert depth.dtype == torch.float64 and tuple(color.shape) == (ro.shape[0], 3)
    for name, got in (('depth', depth), ('var', var), ('color', color)):
        a, b = got.detach().cpu().numpy().astype(np.float64), g[f'{stage}_{name}'].astype(np.float64)
        assert np.all(np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-5 * max(np.abs(b).max(), 1e-30)), (stage, name, np.abs(a - b).max())
    if stage == 'coarse':
        cot = [torch.from_numpy(g[k]).to(DEV) for k in ('cot_depth', 'cot_var', 'cot_color')]
        loss = (depth * cot[0]).sum() + (var * cot[1]).sum() + (color * cot[2]).sum()
    else:
        loss = E.losses.rgbd_loss(depth, color if stage == 'color' else None, rays['gt_depth'], rays['gt_color'], 0.2)
    loss.backward()
    assert abs(loss.item() - float(g[f'{stage}_loss'])) <= 1e-4 * max(1.0, abs(float(g[f'{stage}_loss'])))
    assert rel_err(ro.grad.cpu().numpy(), g[f'{stage}_g_rays_o']) < 1e-3
    assert rel_err(rd.grad.cpu().numpy(), g[f'{stage}_g_rays_d']) < 1e-3
    used = {'color': ('grid_middle', 'grid_fine', 'grid_color'), 'middle': ('grid_middle',), 'coarse': ('grid_coarse',)}[stage]
    for key in used:
        ref = g.get(f'{stage}_g_{key}')
        if ref is not None and np.abs(ref).max() > 0:
            assert rel_err(cg[key].grad.cpu().numpy(), ref) < 1e-3, key
    n = 0
    for name, p in model.named_parameters():
        ref = g.get(f'{stage}_gp_{name}')
        if ref is not None and np.abs(ref).max() > 0:
            assert rel_err(p.grad.cpu().numpy(), ref) < 1e-3, name
            n += 1
    assert n >= (10 if stage == 'coarse' else 20)


def test_second_pass_distances_and_no_grad_call():
    """The merged distances the second pass renders (functional level) and a forward-only call."""
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd.common import sample_pdf
    s, bound, model, grids, rays, renderer = _setup()
    g = load('tiny_importance')
    with torch.no_grad():
        z1 = EF.sample_rays(rays['rays_o'], rays['rays_d'], rays['gt_depth'], bound, 32, 16)
        pts, _ = EF.ray_points(rays['rays_o'], rays['rays_d'], z1, bound)
        raw1 = EF.eval_points(pts, model, grids, 'color', bound)
        _, _, _, w = EF.composite(raw1.view(-1, 48, 4), z1)
        zs = sample_pdf(.5 * (z1[..., 1:] + z1[..., :-1]), w[..., 1:-1], 8, det=True, device=DEV)
        z2 = torch.sort(torch.cat([z1, zs.double()], -1), -1)[0]
        # the inverse CDF divides by bin masses down to 1e-5 (common.py:54-56): float32 differences of the weights between
        # devices move a few of the 8 extra distances by up to a fraction of their bin; the first pass's 48 are exact
        diff = np.abs(z2.cpu().numpy() - g['color_z_vals'])
        assert np.mean(diff > 1e-5) < 0.01 and diff.max() < 0.02, (np.mean(diff > 1e-5), diff.max())
        d, u, c = renderer.render_batch_ray(grids, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
    assert not d.requires_grad
    assert rel_err(d.cpu().numpy(), g['color_depth']) < 1e-4


def test_hierarchical_render_is_chunked_with_the_whole_call_depth_maxima():
    """a depth-guided hierarchical call above the 64-sample ray limit is chunked internally; the sampler's
    batch maxima span the whole call, so every chunk gets them and the result equals the unchunked call."""
    s, bound, model, grids, rays, renderer = _setup()
    with torch.no_grad():
        want = renderer.render_batch_ray(grids, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
        renderer.HIERARCHICAL_MAX_RAYS = 24                     # 64 rays -> chunks of 24, 24, 16
        got = renderer.render_batch_ray(grids, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
    assert renderer.depth_max_override is None
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.allclose(a.double(), b.double(), rtol=1e-6, atol=1e-9), float((a.double() - b.double()).abs().max())
