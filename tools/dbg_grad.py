import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
rays = bench.make_rays(sc, 64, 1000)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
ro, rd, gd, gc = [t.to(dev) for t in rays]
def run(limit, par):
    EF.ACT_WORKSPACE_LIMIT_BYTES = limit
    grids = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc['grids'].items()}
    for p in model.parameters(): p.requires_grad_(par); p.grad = None
    ro_ = ro.clone().requires_grad_(RG); rd_ = rd.clone().requires_grad_(RG)
    d, v, c = renderer.render_batch_ray(grids, model, rd_, ro_, dev, STG, gt_depth=gd)
    (d.sum() + c.sum()).backward()
    return {k: g.grad.cpu().numpy() for k, g in grids.items() if g.grad is not None}
for par, RG, STG in ((True, True, 'middle'), (False, True, 'middle'), (True, True, 'color')):
    a = run(8 << 30, par); b = run(0, par)
    for k in a:
        x, y = a[k], b[k]
        print(par, RG, STG, k, "nan", np.isnan(x).sum(), "nz saved", (x != 0).sum(), "nz ref", (y != 0).sum(),
              "maxdiff", np.nanmax(np.abs(x - y)), "ref max", np.abs(y).max())
        if k == 'grid_middle':
            bad = np.argwhere(np.isnan(x) | (np.abs(x - y) > 1e-4 * np.abs(y).max()))
            print("  bad count", len(bad), "first", bad[:6].tolist())
