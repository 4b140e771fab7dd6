"""GPU parity on the path's edge cases, HIP (through the C ABI) vs the CPU oracle on identical inputs:
perturbed / inverse-depth sampling, no surface samples, ragged and empty batches, sharded depth maximum,
grids at their borders, repeated calls with in-place updated grids."""
import numpy as np
import pytest
import torch

from oracle import render_oracle as R
from tests.util import STAGES, load, rel_err, tiny_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    from tests.hip_util import tiny_on_gpu
    return tiny_on_gpu()


@pytest.mark.parametrize("lindisp", [False, True])
@pytest.mark.parametrize("perturb", [False, True])
def test_sampling_variants_bit_exact(tiny, lindisp, perturb):
    import evennicer_slam_amd.functional as EF
    s, bound, model, grids, rays, renderer = tiny
    ro, rd, gd = [torch.from_numpy(s[k]) for k in ("rays_o", "rays_d", "gt_depth")]
    gd = gd.clamp(min=0.05) if lindisp else gd          # 1/near needs near > 0 (reference divides by it too)
    t_rand = torch.rand(ro.shape[0], 32, generator=torch.Generator().manual_seed(3)) if perturb else None
    for with_depth in (True, False):
        z_ref = R.sample_depths(ro, rd, gd if with_depth else None, bound, 32, 16, 'color', lindisp, t_rand)
        z = EF.sample_rays(ro.cuda(), rd.cuda(), gd.cuda() if with_depth else None, bound, 32, 16, lindisp,
                           t_rand.cuda() if perturb else None)
        assert tuple(z.shape) == tuple(z_ref.shape)
        assert np.array_equal(z.cpu().numpy(), z_ref.numpy()), (lindisp, perturb, with_depth)


def test_depth_max_override_matches_whole_batch(tiny):
    """A shard with the batch-global maximum samples exactly like the whole batch (parallel.ShardedRenderer)."""
    import evennicer_slam_amd.functional as EF
    s, bound, model, grids, rays, renderer = tiny
    ro, rd, gd = rays['rays_o'], rays['rays_d'], rays['gt_depth']
    z_all = EF.sample_rays(ro, rd, gd, bound, 32, 16)
    dm = EF.batch_depth_max(gd)
    z_a = EF.sample_rays(ro[:20], rd[:20], gd[:20], bound, 32, 16, depth_max=dm)
    z_b = EF.sample_rays(ro[20:], rd[20:], gd[20:], bound, 32, 16, depth_max=dm)
    assert torch.equal(torch.cat([z_a, z_b]), z_all)
    z_wrong = EF.sample_rays(ro[:20], rd[:20], gd[:20], bound, 32, 16)          # local maximum differs
    assert not torch.equal(z_wrong, z_all[:20])


@pytest.mark.parametrize("n", [1, 3, 17, 63])
def test_ragged_batches_match_oracle(tiny, n):
    s, bound, model, grids, rays, renderer = tiny
    params, ogrids, obound, _ = tiny_scene()
    idx = torch.arange(n) * 64 // max(n, 1) % 64
    ro, rd, gd = [torch.from_numpy(s[k])[idx] for k in ("rays_o", "rays_d", "gt_depth")]
    gro, grd = ro.cuda().requires_grad_(True), rd.cuda().requires_grad_(True)
    d, v, c = renderer.render_batch_ray(grids, model, grd, gro, 'cuda:0', 'color', gt_depth=gd.cuda())
    (d.sum() + c.sum()).backward()
    oro, ord_ = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
    d0, v0, c0 = R.render_batch_ray(params, ogrids, ord_, oro, 'color', obound, gt_depth=gd)
    (d0.sum() + c0.sum()).backward()
    assert rel_err(d.detach().cpu().numpy(), d0.detach().numpy()) < 1e-4
    assert rel_err(c.detach().cpu().numpy(), c0.detach().numpy()) < 1e-4
    assert rel_err(v.detach().cpu().numpy(), v0.detach().numpy()) < 1e-4
    assert rel_err(grd.grad.cpu().numpy(), ord_.grad.numpy()) < 1e-3
    for p_ in model.parameters():
        p_.grad = None


def test_empty_batch_behaviour(tiny):
    s, bound, model, grids, rays, renderer = tiny
    e3, e1 = torch.zeros(0, 3, device='cuda'), torch.zeros(0, device='cuda')
    with pytest.raises(RuntimeError):
        renderer.render_batch_ray(grids, model, e3, e3, 'cuda:0', 'color', gt_depth=e1)
    d, v, c = renderer.render_batch_ray(grids, model, e3, e3, 'cuda:0', 'coarse')
    assert d.shape == (0,) and c.shape == (0, 3) and d.dtype == torch.float64


def test_no_surface_samples_32(tiny):
    """N_surface = 0 with depth guidance: 32 unsorted-merge-free samples (Renderer.py:169 skips the sort)."""
    from tests.hip_util import cfg_like, renderer_for
    s, bound, model, grids, rays, renderer = tiny
    params, ogrids, obound, _ = tiny_scene()
    r32 = renderer_for(bound, cfg=cfg_like(32, 0))
    d, v, c = r32.render_batch_ray(grids, model, rays['rays_d'], rays['rays_o'], 'cuda:0', 'fine', gt_depth=rays['gt_depth'])
    ro, rd, gd = [torch.from_numpy(s[k]) for k in ("rays_o", "rays_d", "gt_depth")]
    d0, v0, c0 = R.render_batch_ray(params, ogrids, rd, ro, 'fine', obound, gt_depth=gd, n_samples=32, n_surface=0)
    assert rel_err(d.detach().cpu().numpy(), d0.detach().numpy()) < 1e-4
    assert rel_err(v.detach().cpu().numpy(), v0.detach().numpy()) < 1e-4
    for p_ in model.parameters():
        p_.grad = None


@pytest.mark.parametrize("n_samples,n_surface,stage", [(32, 0, "color"), (16, 0, "color"), (16, 16, "fine"), (16, 0, "middle")])
def test_other_sample_counts_forward_and_backward(tiny, n_samples, n_surface, stage):
    """S = 32 and S = 16 (2 and 1 tiles per ray): outputs and every gradient against the oracle."""
    from tests.hip_util import cfg_like, renderer_for
    s, bound, model, grids, rays, renderer = tiny
    params, ogrids, obound, _ = tiny_scene()
    r = renderer_for(bound, cfg=cfg_like(n_samples, n_surface))
    for p_ in model.parameters():
        p_.grad = None
    cg = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    gro, grd = rays['rays_o'].clone().requires_grad_(True), rays['rays_d'].clone().requires_grad_(True)
    d, v, c = r.render_batch_ray(cg, model, grd, gro, 'cuda:0', stage, gt_depth=rays['gt_depth'])
    (d.sum() + 0.5 * v.sum() + c.sum()).backward()
    op = {k: t.clone().requires_grad_(True) for k, t in params.items()}
    og = {k: t.clone().requires_grad_(True) for k, t in ogrids.items()}
    ro, rd, gd = [torch.from_numpy(s[k]) for k in ("rays_o", "rays_d", "gt_depth")]
    oro, ord_ = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
    d0, v0, c0 = R.render_batch_ray(op, og, ord_, oro, stage, obound, gt_depth=gd, n_samples=n_samples, n_surface=n_surface)
    (d0.sum() + 0.5 * v0.sum() + c0.sum()).backward()
    assert rel_err(d.detach().cpu().numpy(), d0.detach().numpy()) < 1e-4
    assert rel_err(v.detach().cpu().numpy(), v0.detach().numpy()) < 1e-4
    assert rel_err(c.detach().cpu().numpy(), c0.detach().numpy()) < 1e-4
    assert rel_err(grd.grad.cpu().numpy(), ord_.grad.numpy()) < 1e-3
    assert rel_err(gro.grad.cpu().numpy(), oro.grad.numpy()) < 1e-3
    checked = 0
    for k in og:
        if og[k].grad is not None and float(og[k].grad.abs().max()) > 0:
            assert rel_err(cg[k].grad.cpu().numpy(), og[k].grad.numpy()) < 1e-3, k
            checked += 1
    for name, p_ in model.named_parameters():
        ref = op[name].grad
        if ref is not None and float(ref.abs().max()) > 0:
            assert rel_err(p_.grad.cpu().numpy(), ref.numpy()) < 1e-3, name
            checked += 1
    assert checked >= 10
    for p_ in model.parameters():
        p_.grad = None


def test_points_on_grid_borders_and_outside(tiny):
    """eval_points on the bound's faces, corners and far outside: border clamp + occ = 100 mask, as the oracle."""
    s, bound, model, grids, rays, renderer = tiny
    params, ogrids, obound, _ = tiny_scene()
    lo, hi = obound[:, 0], obound[:, 1]
    pts = [lo, hi, (lo + hi) / 2, lo - 1.0, hi + 1.0, torch.stack([lo[0], hi[1], (lo[2] + hi[2]) / 2]),
           lo + 1e-9, hi - 1e-9]
    p = torch.stack(pts).double()
    for stage in STAGES:
        with torch.no_grad():
            raw = renderer.eval_points(p.cuda(), model, grids, stage, 'cuda:0')
        ref = R.eval_points(params, ogrids, p, stage, obound)
        assert rel_err(raw.cpu().numpy(), ref.detach().numpy()) < 1e-4, stage
        assert (raw[:2, 3] == 100).all() and (raw[3:5, 3] == 100).all() and raw[2, 3] != 100


def test_in_place_grid_update_is_seen(tiny):
    """The voxel-major cache is keyed on tensor identity + version: optimiser-style in-place updates and
    replaced tensors both take effect (Mapper.py:450-458,633-641; Tracker.py:257-259)."""
    s, bound, model, grids, rays, renderer = tiny
    g2 = {k: v.clone() for k, v in grids.items()}
    args = (model, rays['rays_d'], rays['rays_o'], 'cuda:0', 'color')
    with torch.no_grad():
        a = renderer.render_batch_ray(g2, *args, gt_depth=rays['gt_depth'])[2].clone()
        g2['grid_color'].mul_(1.5)                                  # in place: version bump
        b = renderer.render_batch_ray(g2, *args, gt_depth=rays['gt_depth'])[2].clone()
        g2['grid_color'] = (g2['grid_color'] / 1.5).clone()         # replaced tensor
        c = renderer.render_batch_ray(g2, *args, gt_depth=rays['gt_depth'])[2]
    assert not torch.allclose(a, b)
    assert torch.allclose(a, c, rtol=1e-5, atol=1e-6)


def test_sparse_layout_cache_fills_incrementally(tiny):
    """The voxel-major copy converts only blocks a batch touches; a second batch with other rays on the SAME grid
    version must convert its own blocks (valid bitmap), and eval_points (dense path) must still be right."""
    s, bound, model, grids, rays, renderer = tiny
    params, ogrids, obound, _ = tiny_scene()
    g2 = {k: v.clone() for k, v in grids.items()}
    ro, rd, gd = [torch.from_numpy(s[k]) for k in ("rays_o", "rays_d", "gt_depth")]
    with torch.no_grad():
        for sl in (slice(0, 8), slice(40, 64), slice(0, 64)):
            d, v, c = renderer.render_batch_ray(g2, model, rays['rays_d'][sl], rays['rays_o'][sl], 'cuda:0', 'color',
                                                gt_depth=rays['gt_depth'][sl])
            d0, v0, c0 = R.render_batch_ray(params, ogrids, rd[sl], ro[sl], 'color', obound, gt_depth=gd[sl])
            assert rel_err(d.cpu().numpy(), d0.numpy()) < 1e-4 and rel_err(c.cpu().numpy(), c0.numpy()) < 1e-4
        p = (torch.rand(300, 3, dtype=torch.float64) - 0.5) * 2.0
        raw = renderer.eval_points(p.cuda(), model, g2, 'color', 'cuda:0')
        ref = R.eval_points(params, ogrids, p, 'color', obound)
        assert rel_err(raw.cpu().numpy(), ref.numpy()) < 1e-4


def test_backward_twice_with_retain_graph(tiny):
    """The accumulators are cleared by the forward's prepare launch; a second backward over the same graph clears
    them again itself: identical gradients both times."""
    s, bound, model, grids, rays, renderer = tiny
    for p_ in model.parameters():
        p_.grad = None
    cg = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    ro, rd = rays['rays_o'].clone().requires_grad_(True), rays['rays_d'].clone().requires_grad_(True)
    d, v, c = renderer.render_batch_ray(cg, model, rd, ro, 'cuda:0', 'color', gt_depth=rays['gt_depth'])
    loss = d.sum() + c.sum()
    loss.backward(retain_graph=True)
    first = {k: t.grad.clone() for k, t in cg.items() if t.grad is not None}
    first_ro = ro.grad.clone()
    w0 = model.color_decoder.pts_linears[0].weight.grad.clone()
    for t in cg.values():
        t.grad = None
    ro.grad = None
    for p_ in model.parameters():
        p_.grad = None
    loss.backward()
    for k, g in first.items():
        assert float((cg[k].grad - g).abs().max()) <= 1e-5 * float(g.abs().max()), k
    assert float((ro.grad - first_ro).abs().max()) <= 1e-5 * float(first_ro.abs().max())
    assert float((model.color_decoder.pts_linears[0].weight.grad - w0).abs().max()) <= 1e-5 * float(w0.abs().max())
    for p_ in model.parameters():
        p_.grad = None


def test_sample_counts_that_are_not_whole_tiles():
    """N_samples + N_surface = 20 + 8 = 28 (padded to 32 inside) against the CPU oracle: outputs and gradients."""
    import numpy as np
    from oracle import render_oracle as R
    from tests.hip_util import DEV, cfg_like, model_from_state, renderer_for
    from tests.util import GRID_KEYS, load, rel_err, tiny_scene
    s = load('tiny_scene')
    bound = torch.from_numpy(s['bound'].copy())
    model = model_from_state(s, bound)
    renderer = renderer_for(bound, cfg=cfg_like(n_samples=20, n_surface=8))
    grids = {k: torch.from_numpy(s[k].copy()).to(DEV).requires_grad_(True) for k in GRID_KEYS}
    ro = torch.from_numpy(s['rays_o']).to(DEV).requires_grad_(True)
    rd = torch.from_numpy(s['rays_d']).to(DEV).requires_grad_(True)
    gd, gc = torch.from_numpy(s['gt_depth']).to(DEV), torch.from_numpy(s['gt_color']).to(DEV)
    depth, var, color = renderer.render_batch_ray(grids, model, rd, ro, DEV, 'color', gt_depth=gd)
    (torch.abs(gd - depth)[gd > 0].sum() + 0.2 * torch.abs(gc - color).sum()).backward()
    params, ogrids, obound, _ = tiny_scene()
    og = {k: v.requires_grad_(True) for k, v in ogrids.items()}
    oro = torch.from_numpy(s['rays_o']).requires_grad_(True)
    ord_ = torch.from_numpy(s['rays_d']).requires_grad_(True)
    d0, v0, c0 = R.render_batch_ray(params, og, ord_, oro, 'color', obound, gt_depth=torch.from_numpy(s['gt_depth']),
                                    n_samples=20, n_surface=8)
    R.mapper_loss(d0, c0, torch.from_numpy(s['gt_depth']), torch.from_numpy(s['gt_color']), 'color').backward()
    for got, ref in ((depth, d0), (var, v0), (color, c0)):
        assert rel_err(got.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    assert rel_err(rd.grad.cpu().numpy(), ord_.grad.numpy()) < 1e-3
    for k in ('grid_middle', 'grid_fine', 'grid_color'):
        assert rel_err(grids[k].grad.cpu().numpy(), og[k].grad.numpy()) < 1e-3, k
