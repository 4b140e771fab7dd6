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

# ---- what the bucket would weigh at finer block granularities: count the touched g-voxel blocks of the
#      actual gradients.  Per-link ring time for n_gpus = 8 over xGMI (153 GB/s per link): 2 * (n-1)/n * bytes / BW.
print("bucket size by block granularity (grid part; +0.24 MB of decoder gradients):")
for gsz in (8, 16, 32, 64):
    total = 0
    for k in kinds:
        g = grids[E._lib.GRID_NAMES[k]].grad
        C, V = g.shape[1], g.shape[2] * g.shape[3] * g.shape[4]
        g2 = g.reshape(C, V)
        nfull = V // gsz
        touched = (g2[:, :nfull * gsz].reshape(C, nfull, gsz) != 0).any(dim=2).any(dim=0)
        total += int(touched.sum()) * gsz * C * 4 + (V - nfull * gsz) * C * 4
    nnz = sum(int((grids[E._lib.GRID_NAMES[k]].grad != 0).sum()) * 4 for k in kinds)
    wire = 2 * 7 / 8 * total / 153e9
    print(f"  {gsz:3d}-voxel blocks: {total / 1e6:6.2f} MB  (non-zero gradient entries: {nnz / 1e6:.2f} MB)  "
          f"8-GPU ring all-reduce over one xGMI link ~ {wire * 1e6:5.0f} us; flag bytes {sum((grids[E._lib.GRID_NAMES[k]].numel() // 32 + gsz - 1) // gsz for k in kinds) / 1e3:.1f} KB")
