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


def global_depth_max(gt_depth_local, group=None, force=False, out=None):
    """float32 [2] {max, fl32(max*1.2)} of gt_depth over all ranks' shards (one tiny MAX all-reduce).
    force: issue the collective even in a 1-rank group (rehearsals).  out: a float32 [2] tensor to fill in place (the
    renderer's static `depth_max_override` of a captured step): three launches and the collective, no temporaries."""
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=gt_depth_local.device)
    g = gt_depth_local.detach()
    torch.amax(g if g.dtype == torch.float32 else g.float(), dim=0, keepdim=True, out=out[0:1])
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(out[0:1], op=dist.ReduceOp.MAX, group=group)
    torch.mul(out[0:1], 1.2, out=out[1:2])
    return out


def block_flags_of(grads):
    """uint8 flags of the 64-voxel blocks (flattened z-major voxel index) in which a feature-grid gradient
    [1,C,D,H,W] is nonzero -- the fallback when the renderer's own flags (functional.last_block_flags) are not at
    hand (e.g. CPU tests)."""
    out = []
    for g in grads:
        C, V = g.shape[1], g.shape[2] * g.shape[3] * g.shape[4]
        touched = (g.reshape(C, V) != 0).any(dim=0)
        pad = (-V) % 64
        if pad:
            touched = torch.cat([touched, touched.new_zeros(pad)])
        out.append(touched.view(-1, 64).any(dim=1).to(torch.uint8))
    return out


def _block_views(g):
    """g: gradient [1,C,D,H,W] -> (view of the whole 64-voxel blocks [C,nfull,64], view of the partial last block or
    None).  The partial block (voxel count not a multiple of 64) always travels: a few hundred floats, no
    data-dependent branch."""
    C, V = g.shape[1], g.shape[2] * g.shape[3] * g.shape[4]
    g2 = g.reshape(C, V)
    nfull = V // 64
    main = g2[:, :nfull * 64].view(C, nfull, 64)
    tail = g2[:, nfull * 64:] if V > nfull * 64 else None
    return main, tail


def allreduce_gradients(tensors, group=None, compact_grids=True, block_flags=None, force=False, prepared=None):
    """Sum `.grad` of the given leaf tensors over ranks through one flat bucket (one SUM collective per step).
    Leaves whose grad is None on this rank contribute zeros, so every rank issues the same collectives.

    Feature-grid gradients ([1,C,D,H,W]) are nonzero only in the 64-voxel blocks this step's rays touched.  With
    compact_grids the bucket carries, per grid, only the union over ranks of those blocks: one small MAX
    all-reduce of the block flags (block_flags: {id(tensor): uint8 flags}, normally functional.last_block_flags();
    derived from the gradients when absent), a gather of the flagged blocks, the SUM all-reduce, a scatter back.
    Returns the bucket size in bytes.  force: run the whole sequence even in a 1-rank group (rehearsals).
    prepared: a PreparedFlags made before the local step from flags that are already the union over ranks (HIP
    tensors): no flag collective and no host wait here."""
    from .functional import VoxelMajorGrid
    tensors = [t for t in tensors if t is not None and (isinstance(t, VoxelMajorGrid) or t.requires_grad)]
    if not tensors or not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return 0
    for t in tensors:
        if not isinstance(t, VoxelMajorGrid) and t.grad is None:
            t.grad = torch.zeros_like(t)
    first = tensors[0].grad_vm if isinstance(tensors[0], VoxelMajorGrid) else tensors[0]
    if first.is_cuda:
        return _allreduce_hip(tensors, group, compact_grids, block_flags, prepared)
    if any(isinstance(t, VoxelMajorGrid) for t in tensors):
        raise TypeError("VoxelMajorGrid gradients live on a HIP device")
    for t in tensors:                       # (CPU route: block views need the [C, V] order)
        if not t.grad.is_contiguous():
            t.grad = t.grad.contiguous()
    grid_ids = [i for i, t in enumerate(tensors) if compact_grids and t.dim() == 5 and t.shape[0] == 1]
    plans = {}
    if grid_ids:
        flags = []
        for i in grid_ids:
            f = block_flags.get(id(tensors[i])) if block_flags else None
            flags.append(f if f is not None else block_flags_of([tensors[i].grad])[0])
        sizes_f = [f.numel() for f in flags]
        allf = torch.cat([f.reshape(-1).to(torch.uint8) for f in flags])
        dist.all_reduce(allf, op=dist.ReduceOp.MAX, group=group)          # union of the touched blocks
        views = [_block_views(tensors[i].grad) for i in grid_ids]
        whole = [f[:mv[0].shape[1]] for f, mv in zip(allf.split(sizes_f), views)]       # flags of the whole blocks
        counts = torch.stack([f.sum(dtype=torch.int64) for f in whole]).tolist()         # the one host sync: bucket sizes
        for i, f, n, (main, tail) in zip(grid_ids, whole, counts, views):
            idx = torch.nonzero_static(f, size=int(n)).reshape(-1)                       # known size: no further sync
            plans[i] = (main, idx, tail)
    parts = []
    for i, t in enumerate(tensors):
        if i in plans:
            main, full, tail = plans[i]
            parts.append(main.index_select(1, full).reshape(-1))
            if tail is not None:
                parts.append(tail.reshape(-1))
        else:
            parts.append(t.grad.reshape(-1))
    bucket = torch.cat(parts) if len(parts) > 1 else parts[0].clone()
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for i, t in enumerate(tensors):
        if i in plans:
            main, full, tail = plans[i]
            n = main.shape[0] * full.numel() * 64
            main.index_copy_(1, full, bucket[o:o + n].view(main.shape[0], full.numel(), 64))
            o += n
            if tail is not None:
                tail.copy_(bucket[o:o + tail.numel()].view(tail.shape))
                o += tail.numel()
        else:
            n = t.grad.numel()
            t.grad.copy_(bucket[o:o + n].view(t.grad.shape))
            o += n
    return bucket.numel() * 4


def _allreduce_hip(tensors, group, compact_grids, block_flags, prepared=None):
    """Device tensors: the bucket is packed and unpacked by one HIP launch each (`enslam_bucket_pack/_unpack`) instead
    of ~100 index_select / copy launches.  Per step: flag MAX all-reduce, prefix sum, ONE host read (the bucket
    size, which the collective needs on the host), pack, SUM all-reduce, unpack."""
    import ctypes
    from . import _lib as L
    from .functional import VoxelMajorGrid, _ptr, _stream
    lib = L.lib()
    dev = (tensors[0].grad_vm if isinstance(tensors[0], VoxelMajorGrid) else tensors[0]).device
    grid_items, small = [], []              # (gradient, V, layout, flags) / dense gradients
    C = None
    bs = None                               # voxels per flagged block (set by the first grid's flags)
    for t in tensors:
        if isinstance(t, VoxelMajorGrid):
            g, V, c, layout = t.grad_vm, t.grad_vm.shape[0], t.grad_vm.shape[1], 1
        elif compact_grids and t.dim() == 5 and t.shape[0] == 1:
            if t.grad.is_contiguous(memory_format=torch.channels_last_3d) and not t.grad.is_contiguous():
                # gradient of a channels_last_3d grid: its storage is [V][C] (functional.is_native_grid), a block is one run
                g, V, c, layout = t.grad, t.shape[2] * t.shape[3] * t.shape[4], t.shape[1], 1
            else:
                if not t.grad.is_contiguous():
                    t.grad = t.grad.contiguous()
                g, V, c, layout = t.grad, t.shape[2] * t.shape[3] * t.shape[4], t.shape[1], 0
        else:
            if not t.grad.is_contiguous():
                t.grad = t.grad.contiguous()
            small.append(t.grad)
            continue
        if g.dtype != torch.float32 or (C is not None and c != C) or len(grid_items) == 4:
            small.append(g)                 # travels dense
            continue
        C = c
        f = block_flags.get(id(t)) if block_flags else None
        # flags come at 64 voxels per block (the renderer's own) or finer (batch_block_flags(..., block_voxels=16)); every grid
        # of one bucket must use the same granularity
        fbs = next((b for b in (64, 32, 16, 8) if f is not None and f.numel() == (V + b - 1) // b), None)
        if fbs is None or (bs is not None and fbs != bs):
            fbs = bs or 64
            g2 = g.reshape(c, V) if layout == 0 else (g.permute(0, 2, 3, 4, 1).reshape(V, c) if g.dim() == 5 else g)
            nfull = V // fbs
            if layout == 0:
                f = (g2[:, :nfull * fbs].reshape(c, nfull, fbs) != 0).any(dim=2).any(dim=0).to(torch.uint8)
            else:
                f = (g2[:nfull * fbs].reshape(nfull, fbs * c) != 0).any(dim=1).to(torch.uint8)
            if (V + fbs - 1) // fbs > nfull:                # the partial last block always travels
                f = torch.cat([f, f.new_ones(1)])
        bs = fbs
        grid_items.append((g, V, layout, f.reshape(-1)))
    for g in small:
        if g.dtype != torch.float32:
            raise TypeError(f"gradient bucket carries float32 tensors, got {g.dtype}")
    n_slots, allf, pos = 0, None, None
    if grid_items and prepared is not None and prepared.matches([it[3] for it in grid_items]):
        allf, pos, n_slots = prepared.allf, prepared.pos, prepared.n_slots()      # made while the local step ran
    elif grid_items:
        allf = torch.cat([it[3] for it in grid_items]) if len(grid_items) > 1 else grid_items[0][3].clone()
        dist.all_reduce(allf, op=dist.ReduceOp.MAX, group=group)          # union of the touched blocks
        pos = torch.cumsum(allf, 0, dtype=torch.int32)
        n_slots = int(pos[-1].item())                                     # the one host read: the bucket size
    bs = bs or 64
    slot = (C or 0) * bs
    n_small = sum(g.numel() for g in small)
    bucket = torch.empty(n_slots * slot + n_small, dtype=torch.float32, device=dev)
    ng = len(grid_items)
    gp = (ctypes.c_void_p * max(ng, 1))(*[it[0].data_ptr() for it in grid_items])
    nv = (ctypes.c_int64 * max(ng, 1))(*[it[1] for it in grid_items])
    lay = (ctypes.c_int32 * max(ng, 1))(*[it[2] for it in grid_items])

    def run(fn, name):
        base, first = n_slots * slot, True
        for lo in range(0, max(len(small), 1), L.MAX_SMALL_TENSORS):
            part = small[lo:lo + L.MAX_SMALL_TENSORS]
            sp = (ctypes.c_void_p * max(len(part), 1))(*[g.data_ptr() for g in part])
            sn = (ctypes.c_int64 * max(len(part), 1))(*[g.numel() for g in part])
            L.check(fn(ng if first else 0, gp, C or 1, nv, lay, _ptr(allf), _ptr(pos), len(part), sp, sn, base, _ptr(bucket),
                       bs, _stream()), name)
            base += sum(g.numel() for g in part)
            first = False

    run(lib.enslam_bucket_pack_g, "enslam_bucket_pack")
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    run(lib.enslam_bucket_unpack_g, "enslam_bucket_unpack")
    # The unpack has just written the UNION's blocks of every grid gradient.  Where that gradient is the persistent buffer of
    # a captured step (functional._persist_prev), the finish launch of the next replay only rewrites the blocks it knows of
    # -- this rank's, now and one replay earlier -- so the other ranks' blocks are added to its flags here; without this their
    # sums stay in the buffer and are sent (and grow) again every step.
    if grid_items:
        from .functional import note_foreign_blocks
        for gi, ((g, _V, layout, f), seg) in enumerate(zip(grid_items, allf.split([it[3].numel() for it in grid_items]))):
            if layout == 0 or g.dim() == 5:
                if bs != 64:                # the finish launch keeps its flags per 64 voxels: any finer block set -> its 64-block
                    pre_c = getattr(prepared, 'coarse', None) if (prepared is not None and allf is prepared.allf) else None
                    seg = pre_c[gi] if pre_c is not None else _coarsen(seg, 64 // bs)
                note_foreign_blocks(g, seg)
    return bucket.numel() * 4


def _coarsen(f, k):
    """uint8 flags per b voxels -> per k*b voxels (any fine block set -> its coarse block)."""
    pad = (-f.numel()) % k
    return (torch.cat([f, f.new_zeros(pad)]) if pad else f).view(-1, k).amax(dim=1)


_PINNED = {'buf': None, 'next': 0}


def _pinned_slot():
    """One int32 of page-locked host memory from a small ring allocated once: a fresh `torch.empty(..., pin_memory=True)` per
    step goes through hipHostMalloc / hipHostFree, which wait for the device -- the host then cannot run ahead of the GPU
    and a 0.5 ms step became 1.5 ms (1-rank RCCL rehearsal, round 3)."""
    if _PINNED['buf'] is None:
        _PINNED['buf'] = torch.empty(64, dtype=torch.int32, pin_memory=True)
    i = _PINNED['next']
    _PINNED['next'] = (i + 1) % 64
    return _PINNED['buf'][i:i + 1]


class PreparedFlags:
    """Bucket layout of a step whose touched-block flags are known BEFORE its local step runs and are already the union
    over ranks -- the case of ray sharding, where every rank holds the whole batch (ShardedRenderer) and can mark the
    blocks of all rays itself (`batch_block_flags`).  Construction enqueues the prefix sum and an asynchronous read of
    the bucket size; `allreduce_gradients(..., prepared=this)` after the backward then needs neither the flag
    collective nor a host wait (the read finished while the step was running).

    flag_list: the uint8 flag tensors of the grids, in the order the grids appear in the tensors later handed to
    allreduce_gradients (the same objects as in its block_flags dict).  coarse: their 64-voxel forms when the flags are finer
    (batch_block_flags leaves them under the keys ('c64', id(grid)); derived here when absent)."""

    def __init__(self, flag_list, block_voxels=64, coarse=None):
        self.ids = [(f.data_ptr(), f.numel()) for f in flag_list]
        self.allf = torch.cat([f.reshape(-1) for f in flag_list]) if len(flag_list) > 1 else flag_list[0].reshape(-1).clone()
        # flags finer than the finish launch's 64 voxels per block: their 64-voxel form (what allreduce_gradients reports to the
        # persistent gradient buffers of a captured step) is made here, before the local step, not behind the collective
        self.coarse = None
        if block_voxels != 64:
            self.coarse = list(coarse) if coarse is not None else [_coarsen(f.reshape(-1), 64 // block_voxels) for f in flag_list]
        self.pos = torch.cumsum(self.allf, 0, dtype=torch.int32)
        self.count = _pinned_slot()
        self.count.copy_(self.pos[-1:], non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def matches(self, flag_list):
        return [(f.data_ptr(), f.numel()) for f in flag_list] == self.ids

    def n_slots(self):
        self.event.synchronize()
        return int(self.count[0])


def batch_block_flags(renderer, c, decoders, rays_o, rays_d, gt_depth, stage, out=None, block_voxels=64):
    """{id(grid tensor): uint8 flags} of the 64-voxel blocks the samples of a ray batch touch in the dense grids of
    `stage` -- the sampler's block marking on its own (one launch), for batches that are not rendered here: the other
    ranks' blocks of a sharded batch.  block_voxels 32 / 16 / 8: flags per that many voxels (a second launch marks them from
    the sampled distances): the bucket of a step then carries 3.35 MB (16) instead of 6.52 MB (64) at room0 / 1000 rays.  The batch maxima of gt_depth are taken over the rays given, i.e. pass the whole
    batch.  Deterministic sampling only (perturb == 0).  out: flags dict of an earlier call to refill."""
    import ctypes
    from . import _lib as L
    from . import functional as EF
    if renderer.perturb > 0.:
        raise ValueError("batch_block_flags: perturbed sampling draws its own random numbers per call")
    lib = L.lib()
    dev = rays_o.device
    N = rays_o.shape[0]
    guided = gt_depth is not None and stage != 'coarse'
    n_lin, n_surf = renderer.N_samples, (renderer.N_surface if guided else 0)
    t_lin, t_surf = renderer._t_vals(dev, n_lin, renderer.N_surface)
    kinds = [k for k in EF.stage_kinds(stage) if not isinstance(c[L.GRID_NAMES[k]], EF.VoxelMajorGrid)]
    grids = {k: c[L.GRID_NAMES[k]] for k in kinds}
    bv = int(block_voxels)
    if out is None:                 # every flag array of the call (fine and, when finer than 64, the 64-voxel form) in ONE buffer:
        sizes, keys = [], []        # one clearing launch per call instead of one per array
        for g in grids.values():
            V = g.shape[2] * g.shape[3] * g.shape[4]
            sizes.append((V + bv - 1) // bv)
            keys.append(id(g))
            if bv != 64:
                sizes.append((V + 63) // 64)
                keys.append(('c64', id(g)))
        flat = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
        out = dict(zip(keys, flat.split(sizes)))
        out['_flat'] = flat
    if '_flat' in out:
        out['_flat'].zero_()
    else:
        for k_, f_ in out.items():
            f_.zero_()
    msc = L.Scene()
    msc.bound, msc.coarse_bound = EF.bound6(renderer.bound), EF.bound6(renderer._coarse_bound(decoders))
    fptr = (ctypes.c_void_p * 4)()
    for k, g in grids.items():
        msc.grids[k].D, msc.grids[k].H, msc.grids[k].W = (int(x) for x in g.shape[2:])
        fptr[k] = out[id(g)].data_ptr()
    ro, rd = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float()
    gd = gt_depth.detach().contiguous().float().reshape(-1) if guided else None
    z = torch.empty((N, n_lin + n_surf), dtype=torch.float64, device=dev)
    scratch = torch.empty(2, dtype=torch.float32, device=dev)
    fptr64 = None
    if bv != 64:                    # the 64-voxel form of the same marks (what the finish launch keeps), from the same launch
        fptr64 = (ctypes.c_void_p * 4)()
        for k, g in grids.items():
            key = ('c64', id(g))
            if key not in out:
                out[key] = torch.zeros((g.shape[2] * g.shape[3] * g.shape[4] + 63) // 64, dtype=torch.uint8, device=dev)
            fptr64[k] = out[key].data_ptr()
    L.check(lib.enslam_sample_rays_g(N, n_lin, n_surf, EF._ptr(ro), EF._ptr(rd), EF._ptr(gd), msc.bound, EF._ptr(t_lin),
                                     EF._ptr(t_surf), int(bool(renderer.lindisp)), None, EF._ptr(scratch), 0, EF._ptr(z), L.STAGE[stage],
                                     ctypes.byref(msc), fptr, bv, fptr64, EF._stream()), "enslam_sample_rays")
    return out


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

    def render_img_rescale(self, c, decoders, c2w, device, stage, gt_depth=None, scale_factor=0.1):
        """`Renderer.render_img_rescale` (Renderer.py:258-319) with every chunk's rays split over the ranks: each rank
        renders its block (with gradient), the blocks are all-gathered into the full images, identical on every
        rank, so a replicated consumer (the event U-Net and its loss, Tracker.py:150-232) runs unchanged.  In the
        backward pass a rank keeps the gradient rows of its own block; `allreduce_gradients([camera_tensor])` then
        sums the ranks' pose gradients."""
        from .common import get_rays_rescale
        from .event import resize_bilinear
        r = self.renderer
        H, W = r.H, r.W
        new_H, new_W = int(H * scale_factor), int(W * scale_factor)
        rays_o, rays_d = get_rays_rescale(H, W, new_H, new_W, r.fx, r.fy, r.cx, r.cy, c2w, device)
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        gd = resize_bilinear(gt_depth[None].float(), (new_H, new_W)).reshape(-1) if gt_depth is not None else None
        depths, uncs, colors = [], [], []
        step = r.ray_batch_size
        for i in range(0, rays_d.shape[0], step):                      # the reference's chunks (per-chunk depth maxima)
            g = gd[i:i + step] if gd is not None else None
            (d, u, col), sl = self.render_batch_ray(c, decoders, rays_d[i:i + step], rays_o[i:i + step], device, stage,
                                                    gt_depth=g)
            n = min(step, rays_d.shape[0] - i)
            packed = torch.cat([d.double()[:, None], u.double()[:, None], col.double()], dim=1)     # one collective per chunk
            full = gather_blocks(packed, n, self.rank, self.world, self.group)
            depths.append(full[:, 0])
            uncs.append(full[:, 1])
            colors.append(full[:, 2:5].to(col.dtype))
        depth, unc, color = torch.cat(depths), torch.cat(uncs), torch.cat(colors)
        return depth.reshape(new_H, new_W), unc.reshape(new_H, new_W), color.reshape(new_H, new_W, 3)


class _GatherBlocks(torch.autograd.Function):
    """All-gather of the ranks' row blocks (`shard_range` split of n rows) into the full [n, C] tensor.  Backward:
    every rank holds the same gradient of the replicated consumer, so it keeps the rows of its own block."""

    @staticmethod
    def forward(ctx, block, n, rank, world, group):
        sizes = [shard_range(n, k, world) for k in range(world)]
        width = max(hi - lo for lo, hi in sizes)
        pad = torch.zeros((width,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
        pad[:block.shape[0]] = block.detach()
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        ctx.own = sizes[rank]
        return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)

    @staticmethod
    def backward(ctx, g):
        lo, hi = ctx.own
        return g[lo:hi], None, None, None, None


def gather_blocks(block, n, rank, world, group=None):
    if world == 1:
        return block
    return _GatherBlocks.apply(block, n, rank, world, group)
