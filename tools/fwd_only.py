"""Forward-only timing of render_batch_ray (no activation workspace): RAYS env, stage color."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import evennicer_slam_amd as E
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
n = int(os.environ.get('RAYS', 1000))
rays = bench.make_rays(sc, n, 1000)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
ro, rd, gd, gc = [t.to(dev) for t in rays]
grids = {k: v.to(dev) for k, v in sc['grids'].items()}
with torch.no_grad():
    for it in range(3):
        ts = []
        for i in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            d, v, c = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
            b.record(); ts.append((a, b))
        torch.cuda.synchronize()
    t = np.array([x.elapsed_time(y) for x, y in ts]) * 1e3
print(f"rays {n}: render_batch_ray no_grad median {np.median(t):.1f} us  ({n / np.median(t):.2f} M rays/s)")
