import sys, os, types, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, evennicer_slam_amd as E
from tests.util import load, rel_err
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
g = load('room0_color1000')
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
t = lambda k: torch.from_numpy(g[k]).to(dev)
cg = {k: v.to(dev).requires_grad_(True) for k, v in sc['grids'].items()}
ro = t('rays_o').requires_grad_(True); rd = t('rays_d').requires_grad_(True)
gd, gc = t('gt_depth'), t('gt_color')
d, v, c = renderer.render_batch_ray(cg, model, rd, ro, dev, 'color', gt_depth=gd)
bench.mapper_loss(d, c, gd, gc, 'color').backward()
for key in ('grid_middle', 'grid_fine', 'grid_color'):
    gg = cg[key].grad.contiguous().reshape(-1).cpu().numpy()
    ref = g['gval_' + key]; st = g['gstat_' + key]
    w = float(np.abs(gg[g['gidx_' + key]] - ref).max() / np.abs(ref).max())
    print(key, 'worst sampled', w, 'sum', gg.astype(np.float64).sum(), 'ref', st[0], 'abs', np.abs(gg.astype(np.float64)).sum(), 'ref', st[1], 'nnz', np.count_nonzero(gg), st[2])
