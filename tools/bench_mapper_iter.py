"""Mapper inner iteration on room0, colour stage, 1000 rays x 48 samples: render + RGB-D loss + backward
+ Adam on the frustum-masked grid voxels and the colour decoder.

  glue   : this repo's device-layout path (mapper.MaskedGridOptimizer: grids stay voxel-major, one fused Adam
           launch), whole iteration captured in a hipGraph
  torch  : the reference's own formulation (Mapper.py:448-458, 573-602: val[mask] = val_grad, torch.optim.Adam on the
           compact leaves, write-back) around the same HIP renderer, Python-driven (boolean-mask indexing syncs)

Secondary measurement for DESIGN.md; the headline metric stays bench.py."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF
from evennicer_slam_amd.mapper import MaskedGridOptimizer, FusedAdam
from evennicer_slam_amd.graph import GraphedStep

dev = torch.device('cuda', 0)
KEYS = ('grid_middle', 'grid_fine', 'grid_color')
LR = {'grid_middle': 0.005, 'grid_fine': 0.005, 'grid_color': 0.005}
sc = bench.build_scene_cpu('room0', 0)
rays = bench.make_rays(sc, int(os.environ.get('RAYS', 1000)), 1000)
ro, rd, gd, gc = [t.to(dev) for t in rays]
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
gen = torch.Generator().manual_seed(3)
masks = {}
for k in KEYS:
    D, H, W = sc['grids'][k].shape[2:]
    m = torch.zeros(D, H, W, dtype=torch.bool)
    m[:, :, W // 4: 3 * W // 4] = True                      # half of the volume, frustum-like slab
    masks[k] = m.to(dev)
STEPS = int(os.environ.get('STEPS', 100))


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = None
    for _i in range(n):
        out = None
        out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, out


def run_glue(FREEZE):
    model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
    import copy; model = copy.deepcopy(model)
    grids = {k: v.to(dev).clone() for k, v in sc['grids'].items()}
    if FREEZE:      # fix_fine: True / middle and coarse decoders are never optimised (nice_slam.yaml:51-52, Mapper.py:363-369)
        for name in ('coarse_decoder', 'middle_decoder', 'fine_decoder'):
            for p in getattr(model, name).parameters():
                p.requires_grad_(False)
    opt = MaskedGridOptimizer(grids, masks, keys=KEYS)
    dec_opt = FusedAdam(list(model.color_decoder.parameters()), lr=0.005)
    opt.set_lr(LR)
    c = opt.render_grids()
    one = {}

    def it():
        dec_opt.zero_grad()
        loss, depth, var, color = renderer.render_batch_ray_rgbd_loss(c, model, rd, ro, dev, 'color', gd, gc, 0.2)
        if 'one' not in one: one['one'] = torch.ones_like(loss)
        loss.backward(gradient=one['one'])
        dec_opt.step()
        opt.step()
        return loss

    for _i in range(5): it()
    t_eager, last = timed(it, 30)
    del last                                     # no autograd graph of an eager step may be alive at capture time
    dec_opt.zero_grad()
    import gc as _gc; _gc.collect()
    g = GraphedStep(it)
    t_graph, loss = timed(g.replay, STEPS)
    return t_eager, t_graph, float(loss.item())


def run_torch():
    import copy
    model = copy.deepcopy(sc['model']).to(dev); bench.attach_bounds(model, sc['bound'])
    c = {k: v.to(dev).clone() for k, v in sc['grids'].items()}
    masked, mask5 = {}, {}
    for k in KEYS:
        mask5[k] = masks[k][None, None].repeat(1, 32, 1, 1, 1)
        masked[k] = c[k][mask5[k]].clone().requires_grad_(True)
    optim = torch.optim.Adam([{'params': list(model.color_decoder.parameters()), 'lr': 0.005},
                              {'params': [masked['grid_middle']], 'lr': 0.005},
                              {'params': [masked['grid_fine']], 'lr': 0.005},
                              {'params': [masked['grid_color']], 'lr': 0.005}])

    def it():
        for k in KEYS:
            val = c[k]; val[mask5[k]] = masked[k]; c[k] = val
        optim.zero_grad()
        depth, var, color = renderer.render_batch_ray(c, model, rd, ro, dev, 'color', gt_depth=gd)
        loss = bench.mapper_loss(depth, color, gd, gc, 'color')
        loss.backward()
        optim.step()
        optim.zero_grad()
        for k in KEYS:
            val = c[k].detach(); val[mask5[k]] = masked[k].clone().detach(); c[k] = val
        return loss

    for _ in range(3): it()
    t, loss = timed(it, 20)
    return t, float(loss.item())


te, tg, lg = run_glue(True)
te2, tg2, lg2 = run_glue(False)
tt, lt = run_torch()
n = ro.shape[0]
print(f"mapper iteration, room0 colour stage, {n} rays x 48, masks = 50% of the voxels")
print(f"  device-layout glue, fixed decoders frozen : eager {te * 1e6:8.1f} us/iter, hipGraph {tg * 1e6:8.1f} us/iter  ({n / tg / 1e6:.2f} M rays/s)  loss {lg:.3f}")
print(f"  same, un-optimised decoders left requires_grad=True like the reference: hipGraph {tg2 * 1e6:8.1f} us/iter")
print(f"  torch glue (ref.)  : eager {tt * 1e6:8.1f} us/iter                         ({n / tt / 1e6:.3f} M rays/s)  loss {lt:.3f}")
