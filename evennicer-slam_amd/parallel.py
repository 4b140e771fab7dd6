"""Ray-sharded data parallelism for one optimisation step (new functionality -- the reference
has no multi-GPU path).

Rays are independent units.  One process per GPU; every rank holds replicas of grids and decoders, renders a
contiguous block of the batch, and the leaf gradients are summed with ONE bucketed all-reduce
(torch.distributed: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).  The only cross-ray
coupling in the path, the two batch-global maxima of gt_depth in the sampler (Renderer.py:110,145), is
resolved before sharding so every shard samples exactly what the unsharded call would."""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_depth_max(gt_depth_local, group=None):
    """float32 [2] {max, fl32(max*1.2)} of gt_depth over all ranks' shards (one tiny MAX all-reduce)."""
    m = gt_depth_local.detach().float().max().reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return torch.cat([m, m * 1.2]).contiguous()


def _touched_ranges(grads, group):
    """For feature-grid gradients [1,C,D,H,W] (nonzero only where this step's rays passed): the union over ranks
    of the touched range [lo, hi) of the flattened voxel index (z-major).  One small MAX all-reduce + one host
    read; lets the bucket carry only that slab of every channel row instead of the whole grid."""
    if not grads:
        return []
    dev = grads[0].device
    stats = []
    for g in grads:
        V = g.shape[2] * g.shape[3] * g.shape[4]
        prof = g.reshape(g.shape[1], V).abs().amax(dim=0) > 0                 # [V] touched by this rank
        idx = torch.arange(V, device=dev)
        lo = torch.where(prof, idx, torch.full_like(idx, V)).min()
        hi = torch.where(prof, idx + 1, torch.zeros_like(idx)).max()
        stats += [-lo, hi]                                                      # MAX-reduce both
    st = torch.stack(stats).to(torch.int64)
    dist.all_reduce(st, op=dist.ReduceOp.MAX, group=group)
    vals = st.tolist()
    return [(max(0, -vals[2 * i]), vals[2 * i + 1]) for i in range(len(grads))]


def allreduce_gradients(tensors, group=None, compact_grids=True):
    """Sum `.grad` of the given leaf tensors over ranks through one flat bucket (one collective per step).
    Leaves whose grad is None on this rank contribute zeros, so every rank issues the same collectives.
    compact_grids: 5-D feature-grid gradients contribute only the slab of voxels touched on any rank."""
    tensors = [t for t in tensors if t is not None and t.requires_grad]
    if not tensors or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    dev = tensors[0].device
    for t in tensors:
        if t.grad is None:
            t.grad = torch.zeros_like(t)
    grid_ids = [i for i, t in enumerate(tensors) if compact_grids and t.dim() == 5 and t.shape[0] == 1]
    ranges = dict(zip(grid_ids, _touched_ranges([tensors[i].grad for i in grid_ids], group)))
    views = []
    for i, t in enumerate(tensors):
        if i in ranges:
            lo, hi = ranges[i]
            V = t.shape[2] * t.shape[3] * t.shape[4]
            views.append(t.grad.reshape(t.shape[1], V)[:, lo:max(hi, lo)])       # strided slab of every channel row
        else:
            views.append(t.grad.reshape(-1))
    sizes = [v.numel() for v in views]
    bucket = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    o = 0
    for v, n in zip(views, sizes):
        bucket[o:o + n].view(v.shape).copy_(v)
        o += n
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for v, n in zip(views, sizes):
        v.copy_(bucket[o:o + n].view(v.shape))
        o += n
    return bucket.numel() * 4


class ShardedRenderer:
    """Wraps a Renderer: `render_batch_ray` takes the WHOLE batch (identical on every rank), renders this
    rank's block and returns the block's outputs plus the slice it covers.  The caller computes its loss on
    the block, calls backward, then `allreduce_gradients(leaves)`; identical optimiser steps keep the
    replicas in sync."""

    def __init__(self, renderer, rank=None, world=None, group=None):
        self.renderer = renderer
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(group) if inited else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if inited else 1)

    def render_batch_ray(self, c, decoders, rays_d, rays_o, device, stage, gt_depth=None):
        lo, hi = shard_range(rays_o.shape[0], self.rank, self.world)
        override = None
        if gt_depth is not None and stage != 'coarse':
            m = gt_depth.detach().float().max().reshape(1)            # whole batch is local: no collective
            override = torch.cat([m, m * 1.2]).contiguous()
        prev = getattr(self.renderer, 'depth_max_override', None)
        self.renderer.depth_max_override = override
        try:
            gd = gt_depth[lo:hi] if gt_depth is not None else None
            out = self.renderer.render_batch_ray(c, decoders, rays_d[lo:hi], rays_o[lo:hi], device, stage, gt_depth=gd)
        finally:
            self.renderer.depth_max_override = prev
        return out, slice(lo, hi)
