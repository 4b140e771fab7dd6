"""Host + device cost of the gradient-bucket glue of the ray-sharded step (parallel._allreduce_hip) WITHOUT the
collectives (dist.all_reduce replaced by a no-op): flags -> prefix sum -> host read -> pack -> unpack, on the gradients
and block flags of one real bench step (room0, colour stage, 1000 rays)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF
from evennicer_slam_amd import parallel as PAR

dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
grids = {k: v.to(dev).requires_grad_(True) for k, v in sc['grids'].items()}
ro, rd, gd, gc = [t.to(dev) for t in bench.make_rays(sc, 1000, seed=1000)]
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
kinds = EF.stage_kinds('color')
leaves = [grids[E._lib.GRID_NAMES[k]] for k in kinds]
for k in kinds: leaves += list(getattr(model, E._lib.MLP_NAMES[k]).parameters())
depth, var, color = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
E.losses.rgbd_loss(depth, color, gd, gc, 0.2).backward()
flags = EF.last_block_flags()
dist.all_reduce = lambda *a, **k: None
nbytes = PAR._allreduce_hip(leaves, None, True, flags)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): PAR._allreduce_hip(leaves, None, True, flags)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / n
print(f"bucket glue without collectives: {t * 1e6:.0f} us per step, bucket {nbytes / 1e6:.2f} MB, {len(leaves)} leaves")
