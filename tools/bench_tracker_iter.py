"""Tracker camera iteration (RGB-D part; Tracker.py:141-197 + the optimiser step) on room0:
camera tensor -> pose -> 200 rays of random pixels -> render (colour stage, map and decoders fixed) ->
uncertainty-weighted loss -> backward to the 7 pose parameters -> Adam.  Whole iteration in one hipGraph (the
in-bound prefilter and the boolean masks are applied as multiplicative masks so that nothing syncs)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import evennicer_slam_amd as E
from evennicer_slam_amd.mapper import FusedAdam
from evennicer_slam_amd.graph import GraphedStep

dev = torch.device('cuda', 0)
N = int(os.environ.get('RAYS', 200))
sc = bench.build_scene_cpu('room0', 0)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
for p in model.parameters(): p.requires_grad_(False)
grids = {k: v.to(dev) for k, v in sc['grids'].items()}
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
H, W, fx, fy, cx, cy = (bench.CAM[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
g = torch.Generator().manual_seed(1)
depth_img = (torch.rand(H, W, generator=g) * 3.0 + 0.5).to(dev)
color_img = torch.rand(H, W, 3, generator=g).to(dev)
ct = torch.tensor([1.0, 0.0, 0.0, 0.0, 3.0, 1.0, 0.0], device=dev, requires_grad=True)
opt = FusedAdam([ct], lr=1e-3)
edge = 100
FUSED = os.environ.get('FUSED', '1') == '1'
one = {}

def it():
    opt.zero_grad()
    if FUSED:
        ro, rd, gd, gc = E.tracker.get_samples_from_camera_tensor(edge, H - edge, edge, W - edge, N, H, W, fx, fy, cx, cy, ct, depth_img, color_img, dev)
        depth, unc, color = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
        loss = E.losses.tracker_loss(depth, unc, color, gd, gc, 0.5)
    else:
        c2w = E.common.get_camera_from_tensor(ct)
        ro, rd, gd, gc = E.common.get_samples(edge, H - edge, edge, W - edge, N, H, W, fx, fy, cx, cy, c2w, depth_img, color_img, dev)
        depth, unc, color = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
        m = (gd > 0).to(depth.dtype)
        loss = (torch.abs(gd - depth) / torch.sqrt(unc.detach() + 1e-10) * m).sum() + 0.5 * (torch.abs(gc - color) * m[:, None].float()).sum()
    if 'one' not in one: one['one'] = torch.ones_like(loss)
    loss.backward(gradient=one['one'])
    opt.step()
    return loss

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = None
    for _i in range(n):
        out = None
        out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, out

for _i in range(5): it()
te, last = timed(it, 30)
del last
opt.zero_grad()
import gc as _gc; _gc.collect()
gs = GraphedStep(it)
tg, loss = timed(gs.replay, int(os.environ.get('STEPS', 200)))
print(f"tracker iteration ({'fused pose/loss glue' if FUSED else 'torch pose/loss glue'}), room0, {N} rays x 48, colour stage: eager {te * 1e6:.1f} us, hipGraph {tg * 1e6:.1f} us/iter ({N / tg / 1e6:.2f} M rays/s), loss {float(loss.item()):.3f}")
