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


def allreduce_gradients(tensors, group=None):
    """Sum `.grad` of the given leaf tensors over ranks through one flat bucket (one collective per step).
    Leaves whose grad is None on this rank contribute zeros, so every rank issues the same collective."""
    tensors = [t for t in tensors if t is not None and t.requires_grad]
    if not tensors or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    dev = tensors[0].device
    sizes = [t.numel() for t in tensors]
    bucket = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    o = 0
    for t, n in zip(tensors, sizes):
        if t.grad is None:
            bucket[o:o + n].zero_()
        else:
            bucket[o:o + n].copy_(t.grad.reshape(-1))
        o += n
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for t, n in zip(tensors, sizes):
        g = bucket[o:o + n].view_as(t)
        if t.grad is None:
            t.grad = g.clone()
        else:
            t.grad.copy_(g)
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
