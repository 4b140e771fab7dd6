"""Forward-only bulk consumers: Renderer.render_img (816 000 rays x 48) and eval_points over a dense
lattice (Mesher: 256^3 = 16.7 M points in 500 000-point chunks).  room0, colour stage."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import evennicer_slam_amd as E
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
grids = {k: v.to(dev) for k, v in sc['grids'].items()}
H, W = bench.CAM['H'], bench.CAM['W']
c2w = torch.eye(4, device=dev)[:3].clone(); c2w[:, 3] = torch.tensor([3.0, 1.0, 0.0], device=dev)
gdepth = (torch.rand(H, W, device=dev) * 3.0 + 0.5)

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)

t = timed(lambda: renderer.render_img(grids, model, c2w, dev, 'color', gt_depth=gdepth))
print(f"render_img {H}x{W} = {H * W} rays x 48: {t * 1e3:.1f} ms  ({H * W / t / 1e6:.2f} M rays/s, {H * W * 48 * 103306 / t / 1e12:.1f} TFLOP/s)")
P = int(os.environ.get('POINTS', 4_000_000))
b = sc['bound'].to(dev)
pts = (torch.rand(P, 3, device=dev, dtype=torch.float64) * (b[:, 1] - b[:, 0]) + b[:, 0])
def ev():
    out = []
    for i in range(0, P, 500000):            # Mesher.py:296-319 chunking (points_batch_size)
        out.append(renderer.eval_points(pts[i:i + 500000], model, grids, 'color', dev))
    return out
t = timed(ev)
print(f"eval_points {P} points (500k chunks), colour stage: {t * 1e3:.1f} ms  ({P / t / 1e6:.1f} M points/s, {P * 103306 / t / 1e12:.1f} TFLOP/s)")
