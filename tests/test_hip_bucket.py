"""-m gpu: the gradient bucket of the ray-sharded step (enslam_bucket_pack / _unpack through the C ABI, and
parallel.allreduce_gradients on HIP tensors over a 2-rank gloo group sharing the one GPU)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _arrs(items, ctype):
    return (ctype * max(len(items), 1))(*items)


def test_bucket_pack_unpack_roundtrip_and_layout():
    from evennicer_slam_amd import _lib as L
    from evennicer_slam_amd.functional import _ptr, _stream
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(0)
    C = 32
    shapes = [(5, 7, 11), (4, 8, 8), (3, 5, 13)]                    # 385 (partial last block), 256, 195 voxels
    Vs = [a * b * c for a, b, c in shapes]
    grads = [torch.randn(1, C, *shapes[0], generator=g).to(dev), torch.randn(Vs[1], C, generator=g).to(dev),
             torch.randn(1, C, *shapes[2], generator=g).to(dev)]
    layout = [0, 1, 0]
    nblk = [(v + 63) // 64 for v in Vs]
    flags = (torch.rand(sum(nblk), generator=g) < 0.5).to(torch.uint8)
    flags[nblk[0] - 1] = 1                                           # the partial block of grid 0 travels
    flags[sum(nblk) - 1] = 1                                         # ... and of grid 2
    flags = flags.to(dev)
    pos = torch.cumsum(flags, 0, dtype=torch.int32)
    n_slots = int(pos[-1])
    small = [torch.randn(n, generator=g).to(dev) for n in [1, 3, 1024, 1025, 4096 * 3 + 5] + [7] * 80]   # 85 tensors: two launches
    n_small = sum(t.numel() for t in small)
    bucket = torch.full((n_slots * C * 64 + n_small,), float('nan'), device=dev)
    lib = L.lib()

    def run(fn, grid_tensors, small_tensors, buf):
        base, first = n_slots * C * 64, True
        for lo in range(0, len(small_tensors), 72):
            part = small_tensors[lo:lo + 72]
            L.check(fn(3 if first else 0, _arrs([t.data_ptr() for t in grid_tensors], ctypes.c_void_p), C,
                       _arrs(Vs, ctypes.c_int64), _arrs(layout, ctypes.c_int32), _ptr(flags), _ptr(pos), len(part),
                       _arrs([t.data_ptr() for t in part], ctypes.c_void_p), _arrs([t.numel() for t in part], ctypes.c_int64),
                       base, _ptr(buf), _stream()), "bucket")
            base += sum(t.numel() for t in part)
            first = False

    run(lib.enslam_bucket_pack, grads, small, bucket)
    torch.cuda.synchronize()
    # expected bucket, block by block
    want = []
    fl = flags.cpu().numpy()
    b0 = 0
    for gi, (gr, V, lay) in enumerate(zip(grads, Vs, layout)):
        a = gr.cpu().numpy()
        a = a.reshape(C, V) if lay == 0 else a                       # [C,V] | [V,C]
        for b in range(nblk[gi]):
            if not fl[b0 + b]:
                continue
            lo, hi = b * 64, min(V, b * 64 + 64)
            slot = np.zeros((C, 64), np.float32) if lay == 0 else np.zeros((64, C), np.float32)
            if lay == 0:
                slot[:, :hi - lo] = a[:, lo:hi]
            else:
                slot[:hi - lo] = a[lo:hi]
            want.append(slot.reshape(-1))
        b0 += nblk[gi]
    want += [t.cpu().numpy() for t in small]
    assert np.array_equal(bucket.cpu().numpy(), np.concatenate(want))
    # unpack 2 x bucket into zeroed copies: flagged blocks and small tensors doubled, everything else untouched (zero)
    outs = [torch.zeros_like(t) for t in grads]
    souts = [torch.full_like(t, 5.0) for t in small]
    run(lib.enslam_bucket_unpack, outs, souts, bucket * 2)
    torch.cuda.synchronize()
    b0 = 0
    for gi, (gr, o, V, lay) in enumerate(zip(grads, outs, Vs, layout)):
        vmask = np.repeat(fl[b0:b0 + nblk[gi]], 64)[:V].astype(bool)
        a, r = gr.cpu().numpy(), o.cpu().numpy()
        if lay == 0:
            assert np.array_equal(r.reshape(C, V), a.reshape(C, V) * 2 * vmask[None, :])
        else:
            assert np.array_equal(r, a * 2 * vmask[:, None])
        b0 += nblk[gi]
    for t, o in zip(small, souts):
        assert torch.equal(o, t * 2)
    # argument errors come back as status codes
    assert lib.enslam_bucket_pack(5, None, C, None, None, None, None, 0, None, None, 0, _ptr(bucket), _stream()) != 0
    assert lib.enslam_bucket_pack(0, None, C, None, None, None, None, 0, None, None, 0, None, _stream()) != 0


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from evennicer_slam_amd.parallel import allreduce_gradients
        from evennicer_slam_amd.functional import VoxelMajorGrid
        dev = 'cuda:0'
        g = torch.Generator().manual_seed(5 + rank)
        grid = torch.zeros(1, 32, 5, 7, 11, device=dev).requires_grad_(True)       # 385 voxels
        V = 385
        grad = torch.zeros(32, V)
        for b in ([1, 4] if rank == 0 else [4, 6]):                                # rank 1 touches the partial block
            lo, hi = b * 64, min(V, b * 64 + 64)
            grad[:, lo:hi] = torch.randn(32, hi - lo, generator=g)
        grid.grad = grad.view(1, 32, 5, 7, 11).to(dev)
        vm_grad = torch.zeros(256, 32)
        vm_grad[64 * (rank + 1):64 * (rank + 2)] = torch.randn(64, 32, generator=g)
        vmg = VoxelMajorGrid((4, 8, 8), torch.zeros(256, 32, device=dev), vm_grad.to(dev))
        other = torch.zeros(3, 2, device=dev).requires_grad_(True)
        other.grad = torch.full((3, 2), float(rank + 1), device=dev)
        unused = torch.zeros(4, device=dev).requires_grad_(True)                   # grad None on both ranks
        nbytes = allreduce_gradients([grid, vmg, other, unused])
        torch.cuda.synchronize()
        q.put({'rank': rank, 'nbytes': nbytes, 'grid': grid.grad.cpu().numpy(), 'own': grad.numpy(),
               'vm': vmg.grad_vm.cpu().numpy(), 'own_vm': vm_grad.numpy(), 'other': other.grad.cpu().numpy(),
               'unused': unused.grad.cpu().numpy()})
    finally:
        dist.destroy_process_group()


def test_two_rank_allreduce_of_hip_gradients():
    import torch.multiprocessing as mp
    world, port = 2, 35500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = (outs[0]['own'] + outs[1]['own']).reshape(1, 32, 5, 7, 11)
    want_vm = outs[0]['own_vm'] + outs[1]['own_vm']
    for o in outs:
        assert np.array_equal(o['grid'], want)
        assert np.array_equal(o['vm'], want_vm)
        assert np.array_equal(o['other'], np.full((3, 2), 3.0, dtype=np.float32))
        assert np.array_equal(o['unused'], np.zeros(4, np.float32))
        # grid: blocks 1, 4, 6 (partial, always sent); voxel-major grid: blocks 1, 2; small: 6 + 4 floats
        assert o['nbytes'] == 4 * (32 * 64 * (3 + 2) + 10)


def _shard_worker(rank, world, port, q, block_voxels=64, layout='contiguous'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import evennicer_slam_amd as E
        from evennicer_slam_amd import parallel as PAR
        from tests.hip_util import DEV, as_layout, tiny_on_gpu
        from tests.util import GRID_KEYS
        s, bound, model, grids, rays, renderer = tiny_on_gpu()
        g = {k: as_layout(v, layout).requires_grad_(True) for k, v in grids.items()}
        ro, rd, gd, gc = rays['rays_o'], rays['rays_d'], rays['gt_depth'], rays['gt_color']
        leaves = [g[k] for k in ('grid_middle', 'grid_fine', 'grid_color')] + list(model.color_decoder.parameters())
        # every rank holds the whole batch: union of the touched blocks and the bucket layout before the local step
        flags = PAR.batch_block_flags(renderer, g, model, ro, rd, gd, 'color', block_voxels=block_voxels)
        prepared = PAR.PreparedFlags([flags[id(t)] for t in leaves if t.dim() == 5], block_voxels=block_voxels)
        sr = PAR.ShardedRenderer(renderer)
        (depth, var, color), sl = sr.render_batch_ray(g, model, rd, ro, DEV, 'color', gt_depth=gd)
        E.losses.rgbd_loss(depth, color, gd[sl], gc[sl], 0.2).backward()
        own_flags = E.functional.last_block_flags()
        def coarsen(f):                     # flags per block_voxels voxels -> per 64 voxels (the renderer's own granularity)
            k = 64 // block_voxels
            pad = (-f.numel()) % k
            return (torch.cat([f, f.new_zeros(pad)]) if pad else f).view(-1, k).amax(dim=1)
        covered = all(bool((coarsen(flags[id(t)]) >= own_flags[id(t)]).all()) for t in leaves if t.dim() == 5)
        nbytes = PAR.allreduce_gradients(leaves, block_flags=flags, prepared=prepared)
        torch.cuda.synchronize()
        q.put({'rank': rank, 'nbytes': nbytes, 'covered': covered, 'slice': (sl.start, sl.stop),
               'grads': [t.grad.cpu().numpy() for t in leaves]})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("block_voxels,layout", [(64, 'contiguous'), (16, 'contiguous'), (16, 'channels_last_3d')])
def test_two_rank_sharded_step_with_local_union_flags_matches_unsharded(block_voxels, layout):
    """Ray-sharded step on HIP tensors, 2 ranks: the union of touched blocks marked locally over the whole batch
    (batch_block_flags, at 64 or 16 voxels per block), bucket layout prepared before the step (PreparedFlags), ONE
    collective -- the summed gradients equal the unsharded step's."""
    import torch.multiprocessing as mp
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, tiny_on_gpu
    world, port = 2, 37500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port + block_voxels + (7 if layout != 'contiguous' else 0), q, block_voxels, layout))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    g = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
    leaves = [g[k] for k in ('grid_middle', 'grid_fine', 'grid_color')] + list(model.color_decoder.parameters())
    depth, var, color = renderer.render_batch_ray(g, model, rays['rays_d'], rays['rays_o'], DEV, 'color', gt_depth=rays['gt_depth'])
    E.losses.rgbd_loss(depth, color, rays['gt_depth'], rays['gt_color'], 0.2).backward()
    want = [t.grad.cpu().numpy() for t in leaves]
    assert outs[0]['slice'] == (0, 32) and outs[1]['slice'] == (32, 64)
    for o in outs:
        assert o['covered'] and o['nbytes'] > 0
        for a, b in zip(o['grads'], want):
            assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-30)
    for a, b in zip(outs[0]['grads'], outs[1]['grads']):
        assert np.array_equal(a, b)                         # replicas hold identical sums


def _replay_worker(rank, world, port, q, layout='contiguous'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import evennicer_slam_amd as E
        from evennicer_slam_amd import parallel as PAR
        from evennicer_slam_amd.graph import GraphedStep
        from tests.hip_util import DEV, as_layout, tiny_on_gpu
        s, bound, model, grids, rays, renderer = tiny_on_gpu()
        g = {k: as_layout(v, layout).requires_grad_(True) for k, v in grids.items()}
        ro, rd, gd, gc = [rays[k].clone() for k in ('rays_o', 'rays_d', 'gt_depth', 'gt_color')]
        base = [t.clone() for t in (ro, rd, gd, gc)]
        leaves = [g[k] for k in ('grid_middle', 'grid_fine', 'grid_color')] + list(model.color_decoder.parameters())
        sr = PAR.ShardedRenderer(renderer)

        def local_step():
            for t in leaves:
                t.grad = None
            (depth, var, color), sl = sr.render_batch_ray(g, model, rd, ro, DEV, 'color', gt_depth=gd)
            loss = E.losses.rgbd_loss(depth, color, gd[sl], gc[sl], 0.2)
            loss.backward()
            return loss

        gs = GraphedStep(local_step)
        out = []
        for it in range(4):                     # a different half of the rays per replay: the touched blocks move
            for t, b in zip((ro, rd, gd, gc), base):
                t.copy_(torch.roll(b, 16 * it, 0) if it < 3 else b)
            gs.replay()
            PAR.allreduce_gradients(leaves, block_flags=renderer.state.last_block_flags())
            torch.cuda.synchronize()
            out.append([t.grad.cpu().numpy().copy() for t in leaves])
        q.put({'rank': rank, 'grads': out})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", ['contiguous', 'channels_last_3d'])
def test_two_rank_graph_replay_with_allreduce_keeps_foreign_blocks_clean(layout):
    """under hipGraph replay the dense grid gradients are persistent buffers of which the finish launch
    rewrites only the blocks this rank touched now or one replay earlier; the all-reduce's unpack writes the union over ranks
    into the same buffers.  Four replays with changing rays + allreduce_gradients after each must equal the unsharded eager
    gradients of the same rays (stale sums of the other rank's blocks would be sent again and grow)."""
    import torch.multiprocessing as mp
    import evennicer_slam_amd as E
    from tests.hip_util import DEV, tiny_on_gpu
    world, port = 2, 39500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_replay_worker, args=(r, world, port + (11 if layout != 'contiguous' else 0), q, layout)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=900) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    base = [rays[k] for k in ('rays_o', 'rays_d', 'gt_depth', 'gt_color')]
    for it in range(4):
        ro, rd, gd, gc = [torch.roll(b, 16 * it, 0) if it < 3 else b for b in base]
        g = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
        leaves = [g[k] for k in ('grid_middle', 'grid_fine', 'grid_color')] + list(model.color_decoder.parameters())
        for t in model.color_decoder.parameters():
            t.grad = None
        depth, var, color = renderer.render_batch_ray(g, model, rd, ro, DEV, 'color', gt_depth=gd)
        E.losses.rgbd_loss(depth, color, gd, gc, 0.2).backward()
        want = [t.grad.cpu().numpy() for t in leaves]
        for o in outs:
            for a, b in zip(o['grads'][it], want):
                assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30), (it, o['rank'])
                assert np.array_equal(a == 0, b == 0) or np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30)


def _pose_hip_step(sharded):
    """A bundle-adjustment batch on the HIP path: rays of two frames from their camera tensors (tracker.get_samples_from_camera_tensor:
    the fused pose -> ray launch), colour-stage render, mapper loss, backward to grids, colour decoder and BOTH camera tensors."""
    import evennicer_slam_amd as E
    from evennicer_slam_amd import parallel as PAR
    from evennicer_slam_amd.tracker import get_samples_from_camera_tensor
    from tests.hip_util import DEV, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    g = {k: v.detach().clone().requires_grad_(True) for k, v in grids.items()}
    H, W, fx, fy, cx, cy = 48, 64, 50.0, 50.0, 31.5, 23.5
    gen = torch.Generator().manual_seed(4)
    depth_img = (torch.rand(H, W, generator=gen) * 0.8 + 0.4).to(DEV)
    color_img = torch.rand(H, W, 3, generator=gen).to(DEV)
    cams = [torch.tensor([1.0, 0.02, -0.03, 0.01, 0.1, -0.05, 0.2], device=DEV).requires_grad_(True),
            torch.tensor([0.98, -0.05, 0.04, 0.02, 0.12, -0.02, 0.18], device=DEV).requires_grad_(True)]
    torch.manual_seed(11)                                            # (every rank draws the same pixels: it holds the whole batch)
    parts = [get_samples_from_camera_tensor(0, H, 0, W, 32, H, W, fx, fy, cx, cy, ct, depth_img, color_img, DEV) for ct in cams]
    ro, rd, gd, gc = (torch.cat([p[i].float() for p in parts]) for i in range(4))
    leaves = [g[k] for k in ('grid_middle', 'grid_fine', 'grid_color')] + list(model.color_decoder.parameters()) + cams
    if sharded:
        (depth, var, color), sl = PAR.ShardedRenderer(renderer).render_batch_ray(g, model, rd, ro, DEV, 'color', gt_depth=gd)
        E.losses.rgbd_loss(depth, color, gd[sl], gc[sl], 0.2).backward()
        PAR.allreduce_gradients(leaves)
    else:
        depth, var, color = renderer.render_batch_ray(g, model, rd, ro, DEV, 'color', gt_depth=gd)
        E.losses.rgbd_loss(depth, color, gd, gc, 0.2).backward()
    torch.cuda.synchronize()
    return [t.grad.cpu().numpy() for t in leaves]


def _pose_hip_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put({'rank': rank, 'grads': _pose_hip_step(True)})
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_pose_gradients_on_hip_match_unsharded():
    """north_star: "RCCL all-reduce of grid / MLP / pose gradients" -- the camera tensors of a BA batch in the bucket's small-tensor
    section: rank 0 renders frame 0's rays, rank 1 frame 1's; after ONE bucketed all-reduce every rank holds the unsharded
    gradient of both camera tensors (and of the grids and the colour decoder)."""
    import torch.multiprocessing as mp
    world, port = 2, 41500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_pose_hip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _pose_hip_step(False)
    assert np.abs(want[-1]).max() > 0 and np.abs(want[-2]).max() > 0
    for o in outs:
        for a, b in zip(o['grads'], want):
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30)
    for a, b in zip(outs[0]['grads'], outs[1]['grads']):
        assert np.array_equal(a, b)                         # replicas hold identical sums
