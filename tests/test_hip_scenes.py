"""GPU parity at the sizes of BASELINE configs 2, 4 and 5 against reference-generated goldens:

  room0_color1000       Replica room0   (48 MB of grids),  1000 rays x 48  (tests/golden/make_golden.py)
  office0_color5000     Replica office0 (91 MB of grids),  5000 rays x 48  (tests/golden/make_golden_scenes.py)
  recording4_color1000  RPG recording4  (206 MB of grids), 1000 rays x 48, RPG camera

Grids are regenerated from the seed (checked against the fixtures' checksums).  Compared: sample distances and voxel
indices / fractions bit-exact; decoder outputs of the first 256 rays, depth / uncertainty / colour within 1e-4; ray,
decoder and sampled grid gradients within 1e-3 of each tensor's maximum -- for the plain autograd path AND for the
exact step bench.py times (render_batch_ray_rgbd_loss + work list under graph.GraphedStep), with the work list on and
off.  Plus size-independent properties (nnz, sum |g|, linearity of the backward)."""
import gc

import numpy as np
import pytest
import torch

from tests.util import within_rel, load, rel_err

pytestmark = pytest.mark.gpu

FIXTURES = {'room0': 'room0_color1000', 'office0': 'office0_color5000', 'recording4': 'recording4_color1000'}
GRID_KEYS = ('grid_middle', 'grid_fine', 'grid_color')


@pytest.fixture(scope="module", params=['room0', 'office0', 'recording4'])
def scene(request):
    import types
    import bench
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    tag = request.param
    sc = bench.build_scene_cpu(tag, seed=0)
    g = load(FIXTURES[tag])
    assert np.allclose([float(sc['grids'][k].double().sum()) for k in sc['grids']], g["grid_checksum"], rtol=0, atol=1e-8)
    assert np.array_equal(sc['bound'].numpy(), g['bound'])
    if 'cam' in g:
        assert [sc['cam'][k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy')] == list(g['cam'])
    model = sc['model'].cuda()
    bench.attach_bounds(model, sc['bound'])
    grids = {k: v.cuda() for k, v in sc['grids'].items()}
    renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **sc['cam']))
    rays = {k: torch.from_numpy(g[k]).cuda() for k in ('rays_o', 'rays_d', 'gt_depth', 'gt_color')}
    yield tag, sc, g, model, grids, renderer, rays
    del model, grids, renderer, rays
    EF.clear_caches()
    gc.collect()
    torch.cuda.empty_cache()


def _check_outputs(g, depth, var, color, loss):
    for name, got in (("depth", depth), ("var", var), ("color", color)):
        # 1e-4 relative (north_star) with an explicit near-zero floor: depth and its variance are sums of positive terms (floor
        # 0.1 % of the largest value); the colour of random-init decoders is a SIGNED float32 sum over 48 samples that cancels to
        # near zero in places, where both implementations carry ~5e-6 of the largest value as rounding error (floor 10 %)
        ok, worst = within_rel(got.detach().cpu().numpy(), g[name], rel=1e-4, floor={"color": 0.1, "depth": 1e-3}.get(name, 1e-2))      # (variance: a second moment about the rendered depth, floor 1 %)
        assert ok, (name, worst)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))


def _check_grads(g, model, cg, ro, rd):
    assert rel_err(ro.grad.cpu().numpy(), g["g_rays_o"]) < 1e-3
    assert rel_err(rd.grad.cpu().numpy(), g["g_rays_d"]) < 1e-3
    n = 0
    for name, p in model.named_parameters():
        if "gp_" + name in g and np.abs(g["gp_" + name]).max() > 0:
            assert rel_err(p.grad.cpu().numpy(), g["gp_" + name]) < 1e-3, name
            n += 1
    assert n >= 60
    for key in GRID_KEYS:
        gg = cg[key].grad.reshape(-1)
        ref_sum, ref_abs, ref_nnz, ref_size = g[f"gstat_{key}"]
        assert gg.numel() == int(ref_size)
        assert abs(float(gg.double().abs().sum()) - ref_abs) < 1e-3 * ref_abs
        nnz = int((gg != 0).sum())
        assert abs(nnz - ref_nnz) <= 0.001 * ref_nnz + 8, (key, nnz, ref_nnz)
        idx = torch.from_numpy(g[f"gidx_{key}"]).cuda()
        assert rel_err(gg[idx].cpu().numpy(), g[f"gval_{key}"]) < 1e-3, key


def _leaves(model, grids, rays, layout='contiguous'):
    from tests.hip_util import as_layout
    for p in model.parameters():
        p.grad = None
    cg = {k: as_layout(v, layout).requires_grad_(True) for k, v in grids.items()}
    ro = rays['rays_o'].clone().requires_grad_(True)
    rd = rays['rays_d'].clone().requires_grad_(True)
    return cg, ro, rd


def test_sampling_and_voxel_indices_bit_exact(scene):
    """z_vals of the whole batch and the voxel index / fraction arithmetic of the first 256 rays' samples in the three
    grids (the linear index and the cell records are size-dependent: office0 and recording4 have 2-4x room0's voxels)."""
    import evennicer_slam_amd.functional as EF
    tag, sc, g, model, grids, renderer, rays = scene
    z = EF.sample_rays(rays['rays_o'], rays['rays_d'], rays['gt_depth'], sc['bound'], 32, 16)
    assert np.array_equal(z.cpu().numpy(), g["z_vals"])
    assert 'vox_grid_fine_ix' in g and 'mask_256' in g, f"{tag}: the fixture holds no voxel indices (regenerate it: tests/golden/make_golden*.py)"
    pts, mask = EF.ray_points(rays['rays_o'][:256], rays['rays_d'][:256], z[:256], sc['bound'])
    assert np.array_equal(mask.cpu().numpy().reshape(256, -1), g['mask_256'])
    for key in GRID_KEYS:
        ix, iy, iz, fx, fy, fz = EF.voxel_index(pts, sc['bound'], tuple(grids[key].shape[2:]))
        for name, got in (('ix', ix), ('iy', iy), ('iz', iz), ('fx', fx), ('fy', fy), ('fz', fz)):
            assert np.array_equal(got.cpu().numpy(), g[f'vox_{key}_{name}']), (key, name)


def test_decoder_outputs_of_256_rays(scene):
    import evennicer_slam_amd.functional as EF
    tag, sc, g, model, grids, renderer, rays = scene
    assert 'raw_256' in g, f"{tag}: the fixture holds no decoder outputs (regenerate it: tests/golden/make_golden*.py)"
    z = EF.sample_rays(rays['rays_o'], rays['rays_d'], rays['gt_depth'], sc['bound'], 32, 16)
    pts, _ = EF.ray_points(rays['rays_o'][:256], rays['rays_d'][:256], z[:256], sc['bound'])
    with torch.no_grad():
        raw = renderer.eval_points(pts, model, grids, 'color', 'cuda:0')
    ref = g['raw_256'].reshape(-1, 4)
    assert rel_err(raw.cpu().numpy()[:, :3], ref[:, :3]) < 1e-4
    assert np.all(np.abs(raw.cpu().numpy()[:, 3] - ref[:, 3]) <= 1e-4 * np.abs(ref[:, 3]) + 1e-5 * np.abs(ref[:, 3]).max())


@pytest.mark.parametrize("layout", ['contiguous', 'channels_last_3d'])
def test_autograd_path_against_reference(scene, layout):
    """render_batch_ray + the mapper loss in torch ops + backward (what a caller of the reference API runs); with the feature
    grids in the reference's contiguous layout and as channels_last_3d tensors (read in place, gradients in the same format)."""
    import bench
    tag, sc, g, model, grids, renderer, rays = scene
    cg, ro, rd = _leaves(model, grids, rays, layout)
    depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, 'cuda:0', 'color', gt_depth=rays['gt_depth'])
    loss = bench.mapper_loss(depth, color, rays['gt_depth'], rays['gt_color'], 'color')
    loss.backward()
    _check_outputs(g, depth, var, color, loss.item())
    _check_grads(g, model, cg, ro, rd)
    if layout == 'channels_last_3d':
        for k in ('grid_middle', 'grid_fine', 'grid_color'):
            assert cg[k].grad.is_contiguous(memory_format=torch.channels_last_3d) and cg[k].grad.shape == cg[k].shape, k


@pytest.mark.parametrize("work_list,layout", [(True, 'contiguous'), (False, 'contiguous'), (True, 'channels_last_3d')])
def test_bench_step_against_reference(scene, work_list, layout):
    """The exact step bench.py times: caches cleared, render_batch_ray_rgbd_loss (loss folded into the compositing
    launches, backward starting at the decoders from unit gradients), work list of non-zero tiles, captured in ONE
    hipGraph and replayed -- loss and every gradient against the reference fixture, with the work list on and off."""
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd.graph import GraphedStep
    tag, sc, g, model, grids, renderer, rays = scene
    cg, ro, rd = _leaves(model, grids, rays, layout)
    leaves = list(cg.values()) + [ro, rd] + list(model.parameters())
    out = {}
    one = {}
    prev = renderer.state.use_work_list
    renderer.state.use_work_list = work_list
    try:
        def step():
            EF.clear_caches()
            for t in leaves:
                t.grad = None
            loss, depth, var, color = renderer.render_batch_ray_rgbd_loss(cg, model, rd, ro, 'cuda:0', 'color', rays['gt_depth'],
                                                                          rays['gt_color'], 0.2)
            if 'g' not in one:
                one['g'] = torch.ones_like(loss)
            loss.backward(gradient=one['g'])
            out['o'] = (depth, var, color)
            return loss

        gc.collect()
        gs = GraphedStep(step)
        for _ in range(2):
            loss = gs.replay()
        torch.cuda.synchronize()
        frac = renderer.state.last_active_tile_fraction()
        assert (frac is not None and 0.0 < frac <= 1.0) if work_list else frac is None
        depth, var, color = out['o']
        _check_outputs(g, depth, var, color, loss.item())
        _check_grads(g, model, cg, ro, rd)
    finally:
        renderer.state.use_work_list = prev
        del gs
        gc.collect()


def test_backward_is_linear_in_the_loss(scene):
    import bench
    tag, sc, g, model, grids, renderer, rays = scene
    res = []
    for scale in (1.0, 2.0):
        cg, ro, rd = _leaves(model, grids, rays)
        depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, 'cuda:0', 'color', gt_depth=rays['gt_depth'])
        (bench.mapper_loss(depth, color, rays['gt_depth'], rays['gt_color'], 'color') * scale).backward()
        res.append((ro.grad.clone(), cg['grid_fine'].grad.clone(), model.color_decoder.pts_linears[3].weight.grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.allclose(b, 2 * a, rtol=1e-3, atol=1e-5 * float(a.abs().max()))
