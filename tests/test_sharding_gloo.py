"""world_size-2 gloo test (CPU) of the ray-sharded step: shard -> render -> loss -> backward -> one bucketed
all-reduce reproduces the unsharded outputs and gradients.  The render function is the CPU oracle here
(tests may use it); on the GPU box the same wrapper drives the HIP Renderer."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import GRID_KEYS, tiny_scene


class _OracleRenderer:
    """Renderer-shaped adapter around the oracle (supports depth_max_override like the HIP Renderer)."""

    def __init__(self, params, bound):
        self.params, self.bound, self.depth_max_override = params, bound, None

    def render_batch_ray(self, c, decoders, rays_d, rays_o, device, stage, gt_depth=None):
        from oracle import render_oracle as R
        if self.depth_max_override is None or gt_depth is None:
            return R.render_batch_ray(self.params, c, rays_d, rays_o, stage, self.bound, gt_depth=gt_depth)
        # emulate the batch-global maximum: append a zero-contribution ray carrying the max depth
        pad_o = torch.cat([rays_o, rays_o[:1].detach()])
        pad_d = torch.cat([rays_d, rays_d[:1].detach()])
        pad_g = torch.cat([gt_depth, self.depth_max_override[:1].to(gt_depth.dtype)])
        d, v, c_ = R.render_batch_ray(self.params, c, pad_d, pad_o, stage, self.bound, gt_depth=pad_g)
        return d[:-1], v[:-1], c_[:-1]


def _worker(rank, world, port, q, n_rays=64):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from evennicer_slam_amd.parallel import ShardedRenderer, allreduce_gradients
        from oracle import render_oracle as R
        params, grids, bound, s = tiny_scene()
        params = {k: v.requires_grad_(True) for k, v in params.items()}
        grids = {k: v.requires_grad_(True) for k, v in grids.items()}
        ro, rd = torch.from_numpy(s['rays_o'])[:n_rays], torch.from_numpy(s['rays_d'])[:n_rays]
        gd, gc = torch.from_numpy(s['gt_depth'])[:n_rays], torch.from_numpy(s['gt_color'])[:n_rays]
        sr = ShardedRenderer(_OracleRenderer(params, bound))
        (depth, var, color), sl = sr.render_batch_ray(grids, None, rd, ro, 'cpu', 'color', gt_depth=gd)
        R.mapper_loss(depth, color, gd[sl], gc[sl], 'color').backward()
        leaves = [grids[k] for k in GRID_KEYS] + list(params.values())
        nbytes = allreduce_gradients(leaves)
        out = {'rank': rank, 'slice': (sl.start, sl.stop), 'depth': depth.detach().numpy(), 'nbytes': nbytes,
               'g_fine': grids['grid_fine'].grad.numpy().copy(),
               'g_w': params['color_decoder.pts_linears.0.weight'].grad.numpy().copy(),
               'g_coarse_is_none_or_zero': grids['grid_coarse'].grad is None or float(grids['grid_coarse'].grad.abs().max()) == 0}
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_step_matches_unsharded():
    from oracle import render_oracle as R
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded reference step
    params, grids, bound, s = tiny_scene()
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    grids = {k: v.requires_grad_(True) for k, v in grids.items()}
    ro, rd = torch.from_numpy(s['rays_o']), torch.from_numpy(s['rays_d'])
    gd, gc = torch.from_numpy(s['gt_depth']), torch.from_numpy(s['gt_color'])
    depth, var, color = R.render_batch_ray(params, grids, rd, ro, 'color', bound, gt_depth=gd)
    R.mapper_loss(depth, color, gd, gc, 'color').backward()
    assert outs[0]['slice'] == (0, 32) and outs[1]['slice'] == (32, 64)
    got = np.concatenate([o['depth'] for o in outs])
    assert np.array_equal(got, depth.detach().numpy())          # shards sample exactly like the whole batch
    for o in outs:
        assert o['nbytes'] > 0 and o['g_coarse_is_none_or_zero']
        dense_bytes = 4 * (sum(grids[k].numel() for k in GRID_KEYS) + sum(v.numel() for v in params.values()))
        assert o['nbytes'] <= dense_bytes         # (tiny grids: every 64-voxel block is touched)
        ref = grids['grid_fine'].grad.numpy()
        assert np.abs(o['g_fine'] - ref).max() <= 1e-5 * np.abs(ref).max()
        ref = params['color_decoder.pts_linears.0.weight'].grad.numpy()
        assert np.abs(o['g_w'] - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.array_equal(outs[0]['g_fine'], outs[1]['g_fine'])  # replicas hold identical summed gradients


def _sparse_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from evennicer_slam_amd.parallel import allreduce_gradients
        g = torch.Generator().manual_seed(5)
        grid = torch.zeros(1, 4, 5, 7, 11).requires_grad_(True)        # 385 voxels: 6 whole blocks + a partial one
        other = torch.zeros(3, 2).requires_grad_(True)
        V = 385
        grad = torch.zeros(4, V)
        for b in ([1, 4] if rank == 0 else [4, 6]):                      # rank 1 touches the partial block (6)
            lo, hi = b * 64, min(V, b * 64 + 64)
            grad[:, lo:hi] = torch.randn(4, hi - lo, generator=g) + rank
        grid.grad = grad.view(1, 4, 5, 7, 11).clone()
        other.grad = torch.full((3, 2), float(rank + 1))
        nbytes = allreduce_gradients([grid, other])
        q.put({'rank': rank, 'nbytes': nbytes, 'grid': grid.grad.numpy().copy(), 'own': grad.numpy(), 'other': other.grad.numpy().copy()})
    finally:
        dist.destroy_process_group()


def test_block_sparse_bucket_sums_only_touched_blocks():
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (outs[0]['own'] + outs[1]['own']).reshape(1, 4, 5, 7, 11)
    for o in outs:
        assert np.array_equal(o['grid'], want)
        assert np.array_equal(o['other'], np.full((3, 2), 3.0, dtype=np.float32))
        # blocks 1 and 4 (whole) + the partial block of 1 voxel + the 6 floats of `other`
        assert o['nbytes'] == 4 * (4 * 2 * 64 + 4 * 1 + 6)


def _img_inputs():
    g = torch.Generator().manual_seed(11)
    H, W = 48, 64
    depth_img = torch.rand(H, W, generator=g) * 1.4 + 0.2
    depth_img[20:24, :] = 0.0
    th = 0.3
    c2w = torch.tensor([[np.cos(th), 0, np.sin(th), 0.1], [0, 1, 0, -0.05], [-np.sin(th), 0, np.cos(th), 0.2]],
                       dtype=torch.float32)
    wts = torch.rand(12, 16, 3, generator=g)
    return depth_img, c2w, wts


def _img_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from evennicer_slam_amd.parallel import ShardedRenderer, allreduce_gradients
        params, grids, bound, s = tiny_scene()
        depth_img, c2w, wts = _img_inputs()
        c2w.requires_grad_(True)
        inner = _OracleRenderer(params, bound)
        inner.H, inner.W, inner.fx, inner.fy, inner.cx, inner.cy, inner.ray_batch_size = 48, 64, 50.0, 50.0, 31.5, 23.5, 80
        sr = ShardedRenderer(inner)
        depth, unc, color = sr.render_img_rescale(grids, None, c2w, 'cpu', 'color', gt_depth=depth_img, scale_factor=0.25)
        loss = (color * wts).sum() + 0.1 * depth.sum()              # a replicated consumer of the full image
        loss.backward()
        own = c2w.grad.clone()
        allreduce_gradients([c2w])
        q.put({'rank': rank, 'color': color.detach().numpy(), 'depth': depth.detach().numpy(), 'own': own.numpy(),
               'g': c2w.grad.numpy().copy()})
    finally:
        dist.destroy_process_group()


def test_two_rank_row_sharded_image_render_matches_unsharded():
    """render_img_rescale sharded over 2 ranks (3 chunks of 80/80/32 rays, each split 40+40 / 16+16): gathered image
    and summed pose gradient equal the unsharded chunk loop (Renderer.py:296-318)."""
    from oracle import render_oracle as R
    from evennicer_slam_amd.common import get_rays_rescale
    from oracle.event_oracle import resize_bilinear
    world, port = 2, 33500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_img_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params, grids, bound, s = tiny_scene()
    depth_img, c2w, wts = _img_inputs()
    c2w.requires_grad_(True)
    ro, rd = get_rays_rescale(48, 64, 12, 16, 50.0, 50.0, 31.5, 23.5, c2w, 'cpu')
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    gd = torch.from_numpy(resize_bilinear(depth_img.numpy()[None], (12, 16)).reshape(-1))
    ds, cs = [], []
    for i in range(0, 192, 80):
        d, v, col = R.render_batch_ray(params, grids, rd[i:i + 80], ro[i:i + 80], 'color', bound, gt_depth=gd[i:i + 80])
        ds.append(d.double())
        cs.append(col)
    depth, color = torch.cat(ds).reshape(12, 16), torch.cat(cs).reshape(12, 16, 3)
    ((color * wts).sum() + 0.1 * depth.sum()).backward()
    want = c2w.grad.numpy()
    for o in outs:
        # (CPU GEMMs round differently for different batch sizes: tolerance instead of bit equality)
        assert np.abs(o['color'] - color.detach().numpy()).max() <= 1e-5
        assert np.abs(o['depth'] - depth.detach().numpy()).max() <= 1e-5
        assert np.abs(o['g'] - want).max() <= 1e-5 * np.abs(want).max()
    assert np.abs(outs[0]['own'] + outs[1]['own'] - want).max() <= 1e-5 * np.abs(want).max()
    assert np.array_equal(outs[0]['color'], outs[1]['color'])                   # replicas see the identical image
    assert np.abs(outs[0]['own'] - want).max() > 1e-3 * np.abs(want).max()      # each rank really holds only a part


@pytest.mark.parametrize("world,n_rays", [(3, 61), (4, 62)])
def test_sharded_step_with_ray_counts_not_divisible_by_the_world_size(world, n_rays):
    """world sizes 3 and 4, ray counts the world size does not divide (blocks of 21/20/20 and 16/16/15/15
    rays): the shards tile the batch, sample exactly like the whole batch (batch-global depth maxima) and the one bucketed
    all-reduce leaves every rank with the unsharded gradients."""
    from oracle import render_oracle as R
    from evennicer_slam_amd.parallel import shard_range
    port = 33500 + (os.getpid() % 2000) + world
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_rays)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params, grids, bound, s = tiny_scene()
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    grids = {k: v.requires_grad_(True) for k, v in grids.items()}
    ro, rd = torch.from_numpy(s['rays_o'])[:n_rays], torch.from_numpy(s['rays_d'])[:n_rays]
    gd, gc = torch.from_numpy(s['gt_depth'])[:n_rays], torch.from_numpy(s['gt_color'])[:n_rays]
    depth, var, color = R.render_batch_ray(params, grids, rd, ro, 'color', bound, gt_depth=gd)
    R.mapper_loss(depth, color, gd, gc, 'color').backward()
    assert [o['slice'] for o in outs] == [shard_range(n_rays, r, world) for r in range(world)]
    sizes = [b - a for a, b in (o['slice'] for o in outs)]
    assert sum(sizes) == n_rays and max(sizes) - min(sizes) == 1
    # (torch's CPU matmul blocks 21- and 61-row batches differently: equal to float32 rounding, not bit for bit)
    assert np.allclose(np.concatenate([o['depth'] for o in outs]), depth.detach().numpy(), rtol=1e-5, atol=1e-7)
    for o in outs:
        ref = grids['grid_fine'].grad.numpy()
        assert np.abs(o['g_fine'] - ref).max() <= 1e-5 * np.abs(ref).max()
        ref = params['color_decoder.pts_linears.0.weight'].grad.numpy()
        assert np.abs(o['g_w'] - ref).max() <= 1e-5 * np.abs(ref).max()
        assert np.array_equal(o['g_fine'], outs[0]['g_fine'])          # replicas hold identical sums


def _pose_worker(rank, world, port, q):
    """Rays of TWO frames from their camera tensors (a bundle-adjustment batch: Mapper.py:374-390, 502-535), every rank holding
    the whole batch, rendering its shard; the camera tensors travel in the bucket's small-tensor section."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from evennicer_slam_amd.parallel import ShardedRenderer, allreduce_gradients
        q.put(dict(rank=rank, **_pose_step(lambda r: ShardedRenderer(r), allreduce_gradients)))
    finally:
        dist.destroy_process_group()


def _pose_step(wrap, allreduce):
    from oracle import render_oracle as R
    from oracle import tracker_oracle as TO
    params, grids, bound, s = tiny_scene()
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    grids = {k: v.requires_grad_(True) for k, v in grids.items()}
    H, W, fx, fy, cx, cy = 48, 64, 50.0, 50.0, 31.5, 23.5
    g = torch.Generator().manual_seed(4)
    depth_img = torch.rand(H, W, generator=g) * 0.8 + 0.4
    color_img = torch.rand(H, W, 3, generator=g)
    cams = [torch.tensor([1.0, 0.02, -0.03, 0.01, 0.1, -0.05, 0.2]).requires_grad_(True),
            torch.tensor([0.98, -0.05, 0.04, 0.02, 0.12, -0.02, 0.18]).requires_grad_(True)]
    ro, rd, gd, gc = [], [], [], []
    for ct in cams:
        idx = torch.randint(H * W, (20,), generator=g)
        o, d, dep, col = R.sample_pixels(0, H, 0, W, 20, TO.camera_from_tensor(ct), depth_img, color_img, fx, fy, cx, cy, idx=idx)
        ro.append(o); rd.append(d); gd.append(dep); gc.append(col)
    ro, rd, gd, gc = torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)
    renderer = _OracleRenderer(params, bound)
    leaves = [grids[k] for k in GRID_KEYS] + list(params.values()) + cams
    if wrap is None:
        depth, var, color = renderer.render_batch_ray(grids, None, rd, ro, 'cpu', 'color', gt_depth=gd)
        R.mapper_loss(depth, color, gd, gc, 'color').backward()
    else:
        (depth, var, color), sl = wrap(renderer).render_batch_ray(grids, None, rd, ro, 'cpu', 'color', gt_depth=gd)
        R.mapper_loss(depth, color, gd[sl], gc[sl], 'color').backward()
        allreduce(leaves)
    return dict(g_cam=[c.grad.numpy().copy() for c in cams], g_fine=grids['grid_fine'].grad.numpy().copy())


def test_two_rank_sharded_pose_gradients_match_unsharded():
    """north_star: "an RCCL all-reduce of grid / MLP / POSE gradients".  The camera tensors of a BA batch are leaves like any
    other: each rank back-propagates its ray shard to them, the bucket sums them, every rank ends with the unsharded gradient."""
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_pose_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _pose_step(None, None)
    for o in outs:
        for a, b in zip(o['g_cam'], ref['g_cam']):
            assert np.abs(b).max() > 0
            assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
        assert np.abs(o['g_fine'] - ref['g_fine']).max() <= 1e-5 * np.abs(ref['g_fine']).max()
    # rank 0 renders frame 0's rays only, rank 1 frame 1's: before the all-reduce each rank held a gradient for ONE camera
    assert all(np.array_equal(x, y) for x, y in zip(outs[0]['g_cam'], outs[1]['g_cam']))
