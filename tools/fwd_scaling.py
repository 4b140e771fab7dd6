import sys, os, types
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
import evennicer_slam_amd as E
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
grids = {k: v.to(dev) for k, v in sc['grids'].items()}
for p in model.parameters(): p.requires_grad_(False)
for n in (1000, 2000, 5000, 18360):
    rays = bench.make_rays(sc, n, 1000)
    ro, rd, gd, gc = [t.to(dev) for t in rays]
    for mode in ("nograd", "raygrad"):
        ts = []
        for i in range(12):
            r0 = ro.clone().requires_grad_(mode == "raygrad"); r1 = rd.clone().requires_grad_(mode == "raygrad")
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.set_grad_enabled(mode == "raygrad"):
                a.record()
                d, v, c = renderer.render_batch_ray(grids, model, r1, r0, dev, 'color', gt_depth=gd)
                b.record()
            ts.append((a, b))
        torch.cuda.synchronize()
        t = np.median([x.elapsed_time(y) for x, y in ts[3:]]) * 1e3
        print(f"rays {n:6d} {mode:8s} fwd(all launches) {t:8.1f} us  = {t / n * 1000:6.1f} us per 1000 rays")
