"""GPU parity, backward: hand-derived HIP backward (through the C ABI + torch.autograd.Function) vs the
gradients the reference's autograd produced (tests/golden), for all four stages.

Bar: gradients within 1e-3 of the tensor's max magnitude (float atomics reorder sums),
in practice ~1e-5."""
import numpy as np
import pytest
import torch

from tests.util import GRID_KEYS, STAGES, load, rel_err

pytestmark = pytest.mark.gpu
GTOL = 1e-3


@pytest.fixture(scope="module")
def tiny():
    from tests.hip_util import tiny_on_gpu
    return tiny_on_gpu()


def _run(tiny, stage, cot=None, mapper=False):
    s, bound, model, grids, rays, renderer = tiny
    for p in model.parameters():
        p.grad = None
    cg = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    ro = rays['rays_o'].clone().requires_grad_(True)
    rd = rays['rays_d'].clone().requires_grad_(True)
    depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, 'cuda:0', stage, gt_depth=rays['gt_depth'])
    if mapper:
        m = rays['gt_depth'] > 0
        loss = torch.abs(rays['gt_depth'][m] - depth[m]).sum()
        if stage == 'color':
            loss = loss + 0.2 * torch.abs(rays['gt_color'] - color).sum()
    else:
        gd, gv, gc = cot
        loss = (depth * gd).sum() + (var * gv).sum() + (color * gc).sum()
    loss.backward()
    return cg, ro, rd, model, loss


def _check(g, cg, ro, rd, model, tol=GTOL):
    worst = {}
    worst['rays_o'] = rel_err(ro.grad.cpu().numpy(), g["g_rays_o"])
    worst['rays_d'] = rel_err(rd.grad.cpu().numpy(), g["g_rays_d"])
    for k in GRID_KEYS:
        if "g_" + k in g:
            assert cg[k].grad is not None, k
            assert cg[k].grad.shape == cg[k].shape
            worst[k] = rel_err(cg[k].grad.cpu().numpy(), g["g_" + k])
        else:
            assert cg[k].grad is None or float(cg[k].grad.abs().max()) == 0.0, k
    for name, p in model.named_parameters():
        if "gp_" + name in g:
            assert p.grad is not None, name
            ref = g["gp_" + name]
            if np.abs(ref).max() == 0:
                assert float(p.grad.abs().max()) == 0.0, name
            else:
                worst[name] = rel_err(p.grad.cpu().numpy(), ref)
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, f"gradient mismatch: {bad}"
    return worst


@pytest.mark.parametrize("stage", STAGES)
def test_backward_random_cotangents(tiny, stage):
    g = load("tiny_" + stage)
    cot = [torch.from_numpy(g[k]).cuda() for k in ("cot_depth", "cot_var", "cot_color")]
    cg, ro, rd, model, loss = _run(tiny, stage, cot=cot)
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"])))
    worst = _check(g, cg, ro, rd, model)
    assert len(worst) >= 4


def test_backward_mapper_loss_color(tiny):
    g = load("tiny_color_mapperloss")
    cg, ro, rd, model, loss = _run(tiny, "color", mapper=True)
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    _check(g, cg, ro, rd, model)


def test_backward_only_rays_need_grad(tiny):
    """Tracker case (Tracker.py:248-260): grids carry no grad; gradients flow to the rays only."""
    s, bound, model, grids, rays, renderer = tiny
    g = load("tiny_color")
    cot = [torch.from_numpy(g[k]).cuda() for k in ("cot_depth", "cot_var", "cot_color")]
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        ro = rays['rays_o'].clone().requires_grad_(True)
        rd = rays['rays_d'].clone().requires_grad_(True)
        depth, var, color = renderer.render_batch_ray(grids, model, rd, ro, 'cuda:0', 'color', gt_depth=rays['gt_depth'])
        ((depth * cot[0]).sum() + (var * cot[1]).sum() + (color * cot[2]).sum()).backward()
        assert rel_err(ro.grad.cpu().numpy(), g["g_rays_o"]) < GTOL
        assert rel_err(rd.grad.cpu().numpy(), g["g_rays_d"]) < GTOL
    finally:
        for p in model.parameters():
            p.requires_grad_(True)


@pytest.mark.parametrize("stage,train", [("middle", ()), ("fine", ()), ("color", ()), ("color", ("color_decoder",)),
                                         ("fine", ("fine_decoder",))])
def test_backward_with_fixed_decoders(tiny, stage, train):
    """Mapper cases: decoders that are not optimised carry no gradient (fix_fine / stages with decoders_lr 0): those
    roles run in the light kernel, the optimised one (if any) in the persistent kernel -- same grid, ray and decoder
    gradients as the all-parameters golden."""
    s, bound, model, grids, rays, renderer = tiny
    g = load("tiny_" + stage)
    cot = [torch.from_numpy(g[k]).cuda() for k in ("cot_depth", "cot_var", "cot_color")]
    for name, p in model.named_parameters():
        p.requires_grad_(name.split('.')[0] in train)
        p.grad = None
    try:
        cg = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
        ro = rays['rays_o'].clone().requires_grad_(True)
        rd = rays['rays_d'].clone().requires_grad_(True)
        depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, 'cuda:0', stage, gt_depth=rays['gt_depth'])
        ((depth * cot[0]).sum() + (var * cot[1]).sum() + (color * cot[2]).sum()).backward()
        assert rel_err(ro.grad.cpu().numpy(), g["g_rays_o"]) < GTOL
        assert rel_err(rd.grad.cpu().numpy(), g["g_rays_d"]) < GTOL
        for k in GRID_KEYS:
            if "g_" + k in g:
                assert rel_err(cg[k].grad.cpu().numpy(), g["g_" + k]) < GTOL, k
        n_checked = 0
        for name, p in model.named_parameters():
            if name.split('.')[0] in train and "gp_" + name in g and np.abs(g["gp_" + name]).max() > 0:
                assert rel_err(p.grad.cpu().numpy(), g["gp_" + name]) < GTOL, name
                n_checked += 1
            elif name.split('.')[0] not in train:
                assert p.grad is None
        assert n_checked > 0 or not train
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
            p.grad = None


def test_room0_coarse200_backward():
    from tests.hip_util import model_from_state, renderer_for
    g = load("room0_coarse200")
    bound = torch.from_numpy(g["bound"].copy())
    model = model_from_state(g, bound)
    grid = torch.from_numpy(g["grid_coarse"].copy()).cuda().requires_grad_(True)
    renderer = renderer_for(bound, cam=(680, 1200, 600.0, 600.0, 599.5, 339.5))
    ro = torch.from_numpy(g["rays_o"]).cuda().requires_grad_(True)
    rd = torch.from_numpy(g["rays_d"]).cuda().requires_grad_(True)
    gd = torch.from_numpy(g["gt_depth"]).cuda()
    depth, var, color = renderer.render_batch_ray({"grid_coarse": grid}, model, rd, ro, 'cuda:0', 'coarse')
    m = gd > 0
    loss = torch.abs(gd[m] - depth[m]).sum()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert rel_err(grid.grad.cpu().numpy(), g["g_grid_coarse"]) < GTOL
    assert rel_err(rd.grad.cpu().numpy(), g["g_rays_d"]) < GTOL
    assert rel_err(ro.grad.cpu().numpy(), g["g_rays_o"]) < GTOL
    for name, p in model.named_parameters():
        if "gp_" + name in g and np.abs(g["gp_" + name]).max() > 0:
            assert rel_err(p.grad.cpu().numpy(), g["gp_" + name]) < GTOL, name


@pytest.mark.parametrize("stage", ["middle", "fine", "color"])
def test_backward_recompute_fallback_matches(tiny, stage, monkeypatch):
    """Without an activation workspace (limit 0) the backward recomputes the decoder forward: same gradients."""
    import evennicer_slam_amd.functional as EF
    g = load("tiny_" + stage)
    cot = [torch.from_numpy(g[k]).cuda() for k in ("cot_depth", "cot_var", "cot_color")]
    monkeypatch.setattr(EF, "ACT_WORKSPACE_LIMIT_BYTES", 0)
    cg, ro, rd, model, loss = _run(tiny, stage, cot=cot)
    worst = _check(g, cg, ro, rd, model)
    assert len(worst) >= 4


def test_large_batch_forward_kernel_feeds_the_same_backward():
    """The one-wave-per-ray forward (ray counts above ENSLAM_TILE_MODE_MAX_RAYS: full-image renders) writes the same
    activation workspace: run the colour-stage golden check in a child process with the limit lowered to 16 rays."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from tests.util import load, rel_err, GRID_KEYS\n"
        "from tests.hip_util import tiny_on_gpu\n"
        "s, bound, model, grids, rays, renderer = tiny_on_gpu()\n"
        "g = load('tiny_color_mapperloss')\n"
        "cg = {k: v.clone().requires_grad_(True) for k, v in grids.items()}\n"
        "ro = rays['rays_o'].clone().requires_grad_(True); rd = rays['rays_d'].clone().requires_grad_(True)\n"
        "d, v, c = renderer.render_batch_ray(cg, model, rd, ro, 'cuda:0', 'color', gt_depth=rays['gt_depth'])\n"
        "m = rays['gt_depth'] > 0\n"
        "loss = torch.abs(rays['gt_depth'][m] - d[m]).sum() + 0.2 * torch.abs(rays['gt_color'] - c).sum()\n"
        "loss.backward()\n"
        "assert rel_err(d.detach().cpu().numpy(), g['depth']) <= 1e-4\n"
        "w = max(rel_err(cg[k].grad.cpu().numpy(), g['g_' + k]) for k in GRID_KEYS if 'g_' + k in g)\n"
        "w = max(w, rel_err(rd.grad.cpu().numpy(), g['g_rays_d']))\n"
        "w = max([w] + [rel_err(p.grad.cpu().numpy(), g['gp_' + n]) for n, p in model.named_parameters()\n"
        "               if 'gp_' + n in g and np.abs(g['gp_' + n]).max() > 0])\n"
        "print('WORST', w)\n"
        "assert w <= 1e-3\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ENSLAM_TILE_MODE_MAX_RAYS='16')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'WORST' in r.stdout


def _fixture_child(env_extra, scenes=(('room0', 'room0_color1000'),)):
    """A child process (switches that are read once per process) that checks the tiny colour scene (all gradients) and room0 at
    1000 x 48 in both grid layouts (ray gradients, every decoder parameter, sampled grid-gradient entries, gradient sums) against
    the reference-generated fixtures."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, os, types, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import bench, evennicer_slam_amd as E\n"
        "from tests.util import load, rel_err, GRID_KEYS\n"
        "from tests.hip_util import tiny_on_gpu\n"
        "s, bound, model, grids, rays, renderer = tiny_on_gpu()\n"
        "g = load('tiny_color_mapperloss')\n"
        "cg = {k: v.clone().requires_grad_(True) for k, v in grids.items()}\n"
        "ro = rays['rays_o'].clone().requires_grad_(True); rd = rays['rays_d'].clone().requires_grad_(True)\n"
        "d, v, c = renderer.render_batch_ray(cg, model, rd, ro, 'cuda:0', 'color', gt_depth=rays['gt_depth'])\n"
        "m = rays['gt_depth'] > 0\n"
        "(torch.abs(rays['gt_depth'][m] - d[m]).sum() + 0.2 * torch.abs(rays['gt_color'] - c).sum()).backward()\n"
        "w = max(rel_err(cg[k].grad.cpu().numpy(), g['g_' + k]) for k in GRID_KEYS if 'g_' + k in g)\n"
        "w = max(w, rel_err(rd.grad.cpu().numpy(), g['g_rays_d']), rel_err(ro.grad.cpu().numpy(), g['g_rays_o']))\n"
        "w = max([w] + [rel_err(p.grad.cpu().numpy(), g['gp_' + n]) for n, p in model.named_parameters()\n"
        "               if 'gp_' + n in g and np.abs(g['gp_' + n]).max() > 0])\n"
        "print('WORST tiny', w)\n"
        "assert w <= 1e-3\n"
        "dev = torch.device('cuda', 0)\n"
        "del model, grids, renderer, cg\n"
        "for tag, fixture in %r:\n"
        "  sc = bench.build_scene_cpu(tag, 0)\n"
        "  g = load(fixture)\n"
        "  model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])\n"
        "  renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **sc['cam']))\n"
        "  t = lambda k: torch.from_numpy(g[k]).to(dev)\n"
        "  for layout in ('contiguous', 'channels_last_3d'):\n"
        "    cg = {k: (v.to(dev).contiguous(memory_format=torch.channels_last_3d) if layout != 'contiguous' else v.to(dev)).requires_grad_(True)\n"
        "          for k, v in sc['grids'].items()}\n"
        "    for p in model.parameters(): p.grad = None\n"
        "    ro = t('rays_o').requires_grad_(True); rd = t('rays_d').requires_grad_(True)\n"
        "    gd, gc = t('gt_depth'), t('gt_color')\n"
        "    d, v, c = renderer.render_batch_ray(cg, model, rd, ro, dev, 'color', gt_depth=gd)\n"
        "    bench.mapper_loss(d, c, gd, gc, 'color').backward()\n"
        "    w = max(rel_err(rd.grad.cpu().numpy(), g['g_rays_d']), rel_err(ro.grad.cpu().numpy(), g['g_rays_o']))\n"
        "    w = max([w] + [rel_err(p.grad.cpu().numpy(), g['gp_' + n]) for n, p in model.named_parameters()\n"
        "                   if 'gp_' + n in g and np.abs(g['gp_' + n]).max() > 0])\n"
        "    for key in ('grid_middle', 'grid_fine', 'grid_color'):\n"
        "        gg = cg[key].grad.contiguous().reshape(-1).cpu().numpy()\n"
        "        ref = g['gval_' + key]\n"
        "        w = max(w, float(np.abs(gg[g['gidx_' + key]] - ref).max() / np.abs(ref).max()))\n"
        "        st = g['gstat_' + key]\n"
        "        assert abs(gg.astype(np.float64).sum() - st[0]) <= 1e-3 * st[1] and int(np.count_nonzero(gg)) <= st[2] * 1.001 + 8\n"
        "    print('WORST', tag, layout, w)\n"
        "    assert w <= 1e-3\n"
        "    del cg\n"
        "  del model, renderer, sc\n"
        "  torch.cuda.empty_cache()\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), tuple(scenes))
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert all(f'WORST {tag} channels_last_3d' in r.stdout for tag, _f in scenes)


def test_two_kernel_backward_matches_the_reference_fixtures():
    """ENS_BWD2=1: the two-kernel form of the saved-activation backward (csrc/render_bwd2.hip: dX-chain kernel with dedicated
    scatter waves + split-K weight-gradient kernel) against the reference-generated fixtures."""
    _fixture_child({'ENS_BWD2': '1'})


def test_deferred_scatter_matches_the_reference_fixtures():
    """ENSLAM_DEFER_SCATTER=1: the feature-gradient scatter as a launch of its own (csrc/grid_scatter.hip: spatially ordered ray
    groups, LDS table of 64-bit fixed-point sums) against the same fixtures.  Its one documented difference from float32
    accumulation -- elements below 2^-40 of a ray group's largest feature gradient come out as exact zeros -- is why the child
    bounds the non-zero count from above only."""
    _fixture_child({'ENSLAM_DEFER_SCATTER': '1'}, scenes=(('room0', 'room0_color1000'), ('office0', 'office0_color5000')))      # (5000 rays: five chunks of ray groups)


def test_weight_gradients_as_partial_images_match_the_atomic_path():
    """enslam_decoder_bwd_partials + enslam_step_finish_partials (per-workgroup partial images summed by the finish launch)
    against the default float-atomic accumulation, every decoder parameter of the colour stage."""
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    out = {}
    for mode in (False, True):
        old = EF.USE_DW_PARTIALS
        EF.USE_DW_PARTIALS = mode
        try:
            for p in model.parameters():
                p.grad = None
            g = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
            d, v, c = renderer.render_batch_ray(g, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
            E.losses.rgbd_loss(d, c, rays['gt_depth'], rays['gt_color'], 0.2).backward()
            torch.cuda.synchronize()
            out[mode] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        finally:
            EF.USE_DW_PARTIALS = old
    assert out[False].keys() == out[True].keys() and len(out[True]) >= 69
    for n, a in out[False].items():
        b = out[True][n]
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-30), n


def test_deferred_scatter_passes_the_edge_case_suites():
    """ENSLAM_DEFER_SCATTER=1 (read once per process: a child pytest) under the suites that vary the SHAPES the scatter launch
    sees -- ragged and empty batches, 32 / 48 / 64 samples per ray, N_surface = 0, importance sampling, rays leaving the bound,
    both grid layouts, the device-layout mapper step -- all against the oracle / the reference fixtures with their own bars."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    suites = ['tests/test_hip_edge_cases.py', 'tests/test_hip_importance.py', 'tests/test_hip_native_grids.py', 'tests/test_hip_mapper.py']
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-p', 'no:cacheprovider'] + suites, cwd=root,
                       env=dict(os.environ, ENSLAM_DEFER_SCATTER='1'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout
