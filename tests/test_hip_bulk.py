"""Forward-only bulk consumers: `Renderer.render_img` and Mesher-shaped `eval_points` lattices.

* render_img: value parity of the whole image API against the reference's own render_img on the tiny camera
  (tests/golden/tiny_render_img.npz, chunked by ray_batch_size = 1000: 3072 rays -> 4 chunks with per-chunk depth
  maxima), through both forward kernels (tile-per-wave ring kernel and one-wave-per-ray kernel);
* eval_points: a 128^3 lattice over (and beyond) the room0 bound against the CPU oracle, chunk-invariance of the
  points_batch_size loop, and a 256^3 Mesher-sized lattice (Mesher.py:281-319: 16.7 M points) for shape / finiteness /
  the out-of-bound convention."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.util import load, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _img_check(g, stage, d, u, c):
    assert d.dtype == torch.float64 and u.dtype == torch.float64 and c.dtype == torch.float32
    assert tuple(d.shape) == (48, 64) and tuple(c.shape) == (48, 64, 3)
    for name, got in (("depth", d), ("unc", u), ("color", c)):
        a, b = got.cpu().numpy().astype(np.float64), g[f"{stage}_{name}"].astype(np.float64)
        assert np.all(np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-5 * np.abs(b).max()), (stage, name, np.abs(a - b).max())


def test_render_img_matches_reference_image():
    from tests.hip_util import tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    g = load("tiny_render_img")
    renderer.ray_batch_size = int(g["ray_batch_size"])
    c2w = torch.from_numpy(g["c2w"]).cuda()
    gt = torch.from_numpy(g["depth_img"]).cuda()
    for stage in ("color", "middle"):
        d, u, c = renderer.render_img(grids, model, c2w, 'cuda:0', stage, gt_depth=gt)
        _img_check(g, stage, d, u, c)
    assert not d.requires_grad and not c.requires_grad


def test_render_img_one_wave_per_ray_kernel_matches_reference_image():
    """The kernel full-resolution images use (more rays than ENSLAM_TILE_MODE_MAX_RAYS): limit lowered in a child
    process, because the library reads the variable once."""
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.hip_util import tiny_on_gpu
from tests.util import load
from tests.test_hip_bulk import _img_check
s, bound, model, grids, rays, renderer = tiny_on_gpu()
g = load("tiny_render_img")
renderer.ray_batch_size = int(g["ray_batch_size"])
d, u, c = renderer.render_img(grids, model, torch.from_numpy(g["c2w"]).cuda(), 'cuda:0', 'color', gt_depth=torch.from_numpy(g["depth_img"]).cuda())
_img_check(g, 'color', d, u, c)
print("OK")
''' % ROOT
    env = dict(os.environ, ENSLAM_TILE_MODE_MAX_RAYS="16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.fixture(scope="module")
def room0():
    import types
    import bench
    import evennicer_slam_amd as E
    sc = bench.build_scene_cpu('room0', seed=0)
    model = sc['model'].cuda()
    bench.attach_bounds(model, sc['bound'])
    grids = {k: v.cuda() for k, v in sc['grids'].items()}
    renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
    return sc, model, grids, renderer


def _lattice(bound, n, pad=0.05):
    """n^3 points (float64 [n^3,3]) over the bound enlarged by `pad` of its extent per side (Mesher-style get_grid_uniform)."""
    axes = []
    for a in range(3):
        lo, hi = float(bound[a, 0]), float(bound[a, 1])
        e = (hi - lo) * pad
        axes.append(torch.linspace(lo - e, hi + e, n, dtype=torch.float64))
    gx, gy, gz = torch.meshgrid(*axes, indexing='ij')
    return torch.stack([gx, gy, gz], -1).reshape(-1, 3)


def test_eval_points_lattice_128_against_oracle(room0):
    from oracle import render_oracle as R
    sc, model, grids, renderer = room0
    p = _lattice(sc['bound'], 128)
    with torch.no_grad():
        raw = renderer.eval_points(p.cuda(), model, grids, 'color', 'cuda:0').cpu()
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cg = {k: v.cpu() for k, v in grids.items()}
    ref = []
    with torch.no_grad():
        for pi in torch.split(p, 1 << 18):
            ref.append(R.eval_points(params, cg, pi, 'color', sc['bound']))
    ref = torch.cat(ref)
    out_ref, out_got = ref[:, 3] == 100.0, raw[:, 3] == 100.0
    assert torch.equal(out_ref, out_got)                                # the strict bound mask, point by point
    assert 0.1 < float(out_ref.float().mean()) < 0.5
    inside = ~out_ref
    assert rel_err(raw[inside].numpy(), ref[inside].numpy()) < 1e-4
    a, b = raw[inside].numpy().astype(np.float64), ref[inside].numpy().astype(np.float64)
    assert np.all(np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-5 * np.abs(b).max())


def test_eval_points_chunking_is_invisible(room0):
    sc, model, grids, renderer = room0
    p = _lattice(sc['bound'], 96).cuda()
    with torch.no_grad():
        whole = renderer.eval_points(p, model, grids, 'color', 'cuda:0')
        old = renderer.points_batch_size
        renderer.points_batch_size = 100003                             # ragged chunks, last one short
        try:
            parts = renderer.eval_points(p, model, grids, 'color', 'cuda:0')
        finally:
            renderer.points_batch_size = old
    assert torch.equal(whole, parts)


def test_eval_points_mesher_sized_lattice(room0):
    """256^3 = 16.7 M points in 500 000-point chunks, as Mesher.get_mesh evaluates them."""
    sc, model, grids, renderer = room0
    p = _lattice(sc['bound'], 256).cuda()
    with torch.no_grad():
        raw = renderer.eval_points(p, model, grids, 'color', 'cuda:0')
    assert tuple(raw.shape) == (256 ** 3, 4) and bool(torch.isfinite(raw).all())
    lo, hi = sc['bound'][:, 0].cuda(), sc['bound'][:, 1].cuda()
    outside = ~((p > lo) & (p < hi)).all(dim=1)
    assert bool((raw[outside, 3] == 100.0).all()) and bool((raw[~outside, 3] != 100.0).all())
