"""Time decoder_bwd (HIP events) with subsets of the gradients requested: prices dW, scatter and ray-grad parts."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF

dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
rays = bench.make_rays(sc, int(os.environ.get('RAYS', 1000)), 1000)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
ro0, rd0, gd, gc = [t.to(dev) for t in rays]

def run(grid_grad, par_grad, ray_grad, stage='color', steps=30):
    grids = {k: v.to(dev).requires_grad_(bool(grid_grad)) for k, v in sc['grids'].items()}
    for p in model.parameters(): p.requires_grad_(bool(par_grad))
    ro = ro0.clone().requires_grad_(bool(ray_grad)); rd = rd0.clone().requires_grad_(bool(ray_grad))
    renderer.state.profile['decoder_bwd'] = []
    fwd = []
    for i in range(steps + 5):
        for p in model.parameters(): p.grad = None
        for g in grids.values(): g.grad = None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        d, v, c = renderer.render_batch_ray(grids, model, rd, ro, dev, stage, gt_depth=gd)
        b.record()
        fwd.append((a, b))
        bench.mapper_loss(d, c, gd, gc, stage).backward()
    torch.cuda.synchronize()
    ev = renderer.state.profile.pop('decoder_bwd')
    t = np.array([x.elapsed_time(y) for x, y in ev[5:]]) * 1e3
    tf = np.array([x.elapsed_time(y) for x, y in fwd[5:]]) * 1e3
    print(f"stage {stage:6s} grid_grad={int(grid_grad)} par_grad={int(par_grad)} ray_grad={int(ray_grad)}: decoder_bwd median {np.median(t):8.1f} us  min {t.min():8.1f}   fwd(all launches) median {np.median(tf):7.1f} us", flush=True)

combos = [tuple(int(x) for x in os.environ["COMBO"].split(","))] if os.environ.get("COMBO") else [(1,1,1)] if os.environ.get("QUICK") else [(0,0,1),(1,1,1)] if os.environ.get("TRACK") else [(1,1,1),(0,1,1),(1,0,1),(1,1,0),(0,0,1),(0,1,0),(1,0,0)]
for combo in combos:
    run(*combo)
for st in (() if (os.environ.get("QUICK") or os.environ.get("TRACK") or os.environ.get("COMBO")) else ("middle","fine","coarse")):
    run(1,1,1,stage=st)
