#!/usr/bin/env python3
"""Benchmark of the volume-rendering hot path (BASELINE.json metric: rendered rays/sec, fwd+bwd).

One "step" = Renderer.render_batch_ray (sampling, gather, decoders, compositing) + the mapper's loss
(Mapper.py:553-562) + backward producing gradients for the three feature grids, EVERY decoder parameter
(the reference never freezes any, so its autograd computes them all) and the rays -- on synthetic data.

  python bench.py --gpus N --steps K --warmup W [--config 2|4|5]

  --config 2 (default)  BASELINE configs[1]: Replica room0, full 4-level grid, stage `color`, 1000 rays x 48 samples
                        PER GPU (weak scaling: every rank renders its own 1000-ray block of an N x 1000 batch)
  --config 4            BASELINE configs[3]: Replica office0 (91 MB of grids), 5000 rays per iteration SPLIT over the
                        N GPUs (strong scaling: 625 rays per GPU at N = 8), RCCL all-reduce of the leaf gradients
  --config 5            BASELINE configs[4] shapes: RPG recording4 (206 MB of grids, RPG camera), synthetic images,
                        1000 rays per GPU (the real sequence and the callers' pipeline are not in the image)

  N > 1: one rank per GPU under torch.distributed.run.  Started WITHOUT it (`python bench.py --gpus N`, no WORLD_SIZE in the
  environment) the script starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself as a child process
  -- before this process has touched the GPU -- and exits with the child's code; rank 0 of the child prints the JSON line.
  Leaf gradients are summed with one bucketed RCCL all-reduce (touched 64-voxel blocks only).  The default run (config 2) also measures config 4 on the same ranks
  and reports it inside the SAME JSON line under "also" (disable with --no-secondary), so that a 1/2/4/8-GPU sweep
  records both the weak (room0) and the strong (office0, 5000 rays) curve.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the `roofline` and `cpu_baseline` objects).  At N = 1 the default
run adds, under "also": config 4 on this GPU, config 3 (the tracker's camera iteration with the event term: path only, path +
U-Net, one hipGraph per iteration), the 200-ray tracker iteration, and the headline step on a map with real surfaces (an
analytic room fitted by the path's own mapper) next to the random-init scene; `api_rays_per_s` is the Python-driven step through
the reference API only (render_batch_ray + torch loss + backward), no graph capture and no fused-loss entry point.
"""
import argparse
import gc as _gcmod
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# yardstick of SURVEY.md 8(d): linear-layer FLOPs per sample point, 2*MAC
FLOP_FWD_PER_POINT = {'coarse': 12352, 'middle': 30958, 'fine': 72156, 'color': 103306}
PEAK_F32_MFMA = 157.3e12        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM = 8.0e12

SCENES = {      # mapping.bound of the reference configs (configs/Replica/room0.yaml:3, office0.yaml:3, rpg/recording4.yaml:4)
    'room0': [[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]],
    'office0': [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]],
    'recording4': [[-7.0, 9.4], [-6.5, 3.6], [-9.2, 9.5]],
}
CAM = dict(H=680, W=1200, fx=600.0, fy=600.0, cx=599.5, cy=339.5)       # configs/Replica/replica.yaml:37-43
CAM_RPG = dict(H=260, W=346, fx=196.71854278974607, fy=196.68898128242577, cx=172.5, cy=129.5)     # configs/rpg/rpg.yaml:62-68
BUCKET_BLOCK = int(os.environ.get('ENSLAM_BUCKET_BLOCK', '16'))      # voxels per block of the gradient bucket (N > 1)
GRID_LEN = {'coarse': 2, 'middle': 0.32, 'fine': 0.16, 'color': 0.16, 'bound_divisible': 0.32}

# measurement configurations of BASELINE.json / SURVEY.md 8(d): scene, rays, how the rays relate to the GPU count
CONFIGS = {
    2: dict(scene='room0', rays=1000, scaling='weak', name='config 2'),
    4: dict(scene='office0', rays=5000, scaling='strong', name='config 4'),
    5: dict(scene='recording4', rays=1000, scaling='weak', name='config 5 shapes'),
}


def cfg_dict():
    return {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 32, 'N_surface': 16, 'N_importance': 0},
            'scale': 1, 'occupancy': True, 'coarse': True, 'data': {'dim': 3},
            'model': {'c_dim': 32, 'coarse_bound_enlarge': 2, 'pos_embedding_method': 'fourier'},
            'grid_len': dict(GRID_LEN)}


def build_scene_cpu(scene='room0', seed=0):
    """Seeded synthetic scene on the CPU, in the RNG order of BASELINE.md section 3 (decoders, then grids
    coarse/middle/fine/color, then the depth and colour images)."""
    import evennicer_slam_amd as E
    torch.manual_seed(seed)
    cfg = cfg_dict()
    model = E.get_model(cfg)
    bound = E.scene.scene_bound(SCENES[scene], 1.0, GRID_LEN['bound_divisible'])
    grids = E.scene.grid_init(bound, GRID_LEN)
    cam = scene_cam(scene)
    depth_img = torch.rand(cam['H'], cam['W']) * 3.0 + 0.5
    z0 = int(cam['H'] * 300 / 680)
    depth_img[z0:z0 + max(1, int(cam['H'] * 40 / 680)), :] = 0.0   # 5.9 % pixels without depth
    color_img = torch.rand(cam['H'], cam['W'], 3)
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 3] = torch.tensor([3.0, 1.0, 0.0]) if scene == 'room0' else torch.tensor([0.0, 0.0, 0.0])
    return dict(cfg=cfg, model=model, bound=bound, grids=grids, depth_img=depth_img, color_img=color_img, c2w=c2w, cam=cam)


def scene_cam(scene):
    return CAM_RPG if scene == 'recording4' else CAM


def attach_bounds(model, bound):
    model.bound = bound
    for name in ('middle_decoder', 'fine_decoder', 'color_decoder'):
        getattr(model, name).bound = bound
    model.coarse_decoder.bound = bound * 2


def make_rays(sc, n_rays, seed):
    from evennicer_slam_amd.common import get_samples
    torch.manual_seed(seed)
    cam = sc.get('cam', CAM)
    ro, rd, gd, gc = get_samples(0, cam['H'], 0, cam['W'], n_rays, cam['H'], cam['W'], cam['fx'], cam['fy'],
                                 cam['cx'], cam['cy'], sc['c2w'], sc['depth_img'], sc['color_img'], 'cpu')
    return ro.float().contiguous(), rd.float().contiguous(), gd.float().contiguous(), gc.float().contiguous()


def fit_map(renderer, grids, model, sc, dev, iters=400, rays=1000, seed=7):
    """Second scene variant of the bench: fit the map to the analytic room the images now show, through the path itself
    (random pixels -> render_batch_ray_rgbd_loss -> backward -> Adam on the grids and the decoders, colour stage), so that the
    timed step afterwards runs on a map with real surfaces (occupancy rising at gt_depth) instead of random-init occupancy.
    Returns the mean |depth - gt_depth| of a fixed ray batch before and after."""
    from evennicer_slam_amd.common import get_samples
    cam = sc['cam']
    dimg, cimg, c2w = sc['depth_img'].to(dev), sc['color_img'].to(dev), sc['c2w'].to(dev)
    gen_state = torch.cuda.get_rng_state(dev)
    torch.manual_seed(seed)

    def draw(n):
        ro, rd, gd, gc = get_samples(0, cam['H'], 0, cam['W'], n, cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy'],
                                     c2w, dimg, cimg, dev)
        return ro.float(), rd.float(), gd.float(), gc.float()

    ev = draw(rays)

    def depth_l1():
        with torch.no_grad():
            d, _u, _c = renderer.render_batch_ray(grids, model, ev[1], ev[0], dev, 'color', gt_depth=ev[2])
            return float((d - ev[2].double()).abs().mean())

    before = depth_l1()
    gl = [grids[k] for k in ('grid_middle', 'grid_fine', 'grid_color')]
    opt = torch.optim.Adam([{'params': gl, 'lr': 0.02}, {'params': [q for q in model.parameters() if q.requires_grad], 'lr': 2e-3}])
    for _ in range(iters):
        ro, rd, gd, gc = draw(rays)
        opt.zero_grad(set_to_none=True)
        loss, _d, _v, _c = renderer.render_batch_ray_rgbd_loss(grids, model, rd, ro, dev, 'color', gd, gc, 0.2)
        loss.backward()
        opt.step()
    after = depth_l1()
    for t in gl + list(model.parameters()):
        t.grad = None
    torch.cuda.set_rng_state(gen_state, dev)
    return {"iters": iters, "depth_l1_before_m": before, "depth_l1_after_m": after}


def mapper_loss(depth, color, gt_depth, gt_color, stage, w_color=0.2):
    """Mapper.py:553-562: L1 depth over pixels with valid depth (+ w_color * L1 colour in the colour stage).
    Written with a multiplicative mask instead of boolean indexing: same sum, no device->host sync."""
    m = gt_depth > 0
    zero = torch.zeros((), dtype=depth.dtype, device=depth.device)
    # sum over valid pixels of |gt - d|  ==  L1(sum) between the masked tensors (fused abs-diff-sum kernels)
    loss = torch.nn.functional.l1_loss(torch.where(m, depth, zero), torch.where(m, gt_depth.to(depth.dtype), zero),
                                       reduction='sum')
    if stage == 'color':
        loss = loss + w_color * torch.nn.functional.l1_loss(color, gt_color, reduction='sum')
    return loss


def cpu_baseline(sc, rays, stage, budget_s=20.0):
    """The CPU oracle (a port of the reference's PyTorch-CPU op sequence, pinned to reference goldens) timed on
    this host on the same 1000-ray workload: fwd + loss + backward, bounded to ~budget_s of CPU work."""
    from oracle import render_oracle as R
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sc['model'].state_dict().items()}
    grids = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sc['grids'].items()}
    ro, rd, gd, gc = [t.clone() for t in rays]
    ro.requires_grad_(True)
    rd.requires_grad_(True)

    def step():
        for t in list(params.values()) + list(grids.values()) + [ro, rd]:
            t.grad = None
        d, v, c = R.render_batch_ray(params, grids, rd, ro, stage, sc['bound'], gt_depth=gd)
        mapper_loss(d, c, gd, gc, stage).backward()

    # The op mix is many small tensors: more threads is not faster (128 threads ran 2.6x SLOWER than one on the
    # 64-core EPYC of the GPU box), so 1 thread, 8, 32 and ALL PHYSICAL CORES are timed; `value` is the
    # best of them, `cores` the physical core count of the host, `threads_best` the thread count `value` was measured at.
    n = ro.shape[0]
    all_threads = torch.get_num_threads()
    phys, model_name = physical_cores()
    results = {}
    try:
        for nt in sorted({1, 8, 32, phys}):
            if nt > max(all_threads, phys):
                continue
            torch.set_num_threads(nt)
            step()                                          # two warm-ups at this thread count
            step()
            times = []
            t_budget = time.perf_counter()
            while len(times) < 5 and (len(times) < 2 or time.perf_counter() - t_budget < budget_s / 4):
                t0 = time.perf_counter()
                step()
                times.append(time.perf_counter() - t0)
            results[nt] = (float(np.median(times)), len(times))
    finally:
        torch.set_num_threads(all_threads)
    best = min(results, key=lambda k: results[k][0])
    med, iters = results[best]
    others = ", ".join(f"{k} thr {n / v[0]:.0f}" for k, v in sorted(results.items()))
    return {"value": n / med, "unit": "rays/s", "cores": phys, "threads_best": best, "kind": "port", "cpu": model_name,
            "value_1_thread": n / results[1][0] if 1 in results else None,
            "value_all_cores": n / results[phys][0] if phys in results else None,
            "sample": f"{n} rays x 48 samples, stage {stage}, fwd+loss+bwd, torch CPU oracle; median of {iters} iterations after "
                      f"2 warm-ups per thread count (rays/s: {others}); value = the best thread count"}


def physical_cores():
    """(physical core count, CPU model name) of this host from /proc/cpuinfo (fallback: os.cpu_count())."""
    model_name, cores = "", set()
    try:
        phys_id = core_id = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and not model_name:
                    model_name = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    phys_id = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    core_id = ln.split(":", 1)[1].strip()
                elif not ln.strip():
                    if phys_id is not None and core_id is not None:
                        cores.add((phys_id, core_id))
                    phys_id = core_id = None
    except OSError:
        pass
    n = len(cores) if cores else (os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))           # a container's CPU share
    except (AttributeError, OSError):
        pass
    return max(n, 1), model_name


class Env:
    """Process-wide facts of a bench run: rank layout, device, communication switches."""

    def __init__(self, args):
        import torch.distributed as dist
        self.dist = dist
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if args.gpus > 1 and self.world == 1 and os.environ.get('ENSLAM_BENCH_FORCE_COMM') != '1':
            raise SystemExit(f"bench.py --gpus {args.gpus}: WORLD_SIZE is 1 (main() starts the ranks itself when it is unset)")
        if self.world != args.gpus and self.world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the hot path has no CPU implementation")
        # rehearsal on a one-GPU box: ENSLAM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and ENSLAM_BENCH_BACKEND=gloo
        # replaces RCCL (which refuses two ranks on one device); the measured runs use neither
        if os.environ.get('ENSLAM_BENCH_SHARE_GPU') == '1':
            local_rank = 0
        self.backend = os.environ.get('ENSLAM_BENCH_BACKEND', 'nccl')
        torch.cuda.set_device(local_rank)
        self.dev = torch.device('cuda', local_rank)
        # rehearsal of the communication path on ONE rank with real RCCL (collectives of a 1-rank group): not a measurement
        self.force_comm = self.world == 1 and os.environ.get('ENSLAM_BENCH_FORCE_COMM') == '1'
        if self.force_comm:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29777')
            dist.init_process_group(self.backend, rank=0, world_size=1, **({'device_id': self.dev} if self.backend == 'nccl' else {}))
        if self.world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if self.backend == 'nccl':
                dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.comm_on = self.world > 1 or self.force_comm

    def timed(self, fn, n):
        """n calls of fn bracketed by barrier + synchronize on both sides; MAX over ranks of the wall time."""
        dist = self.dist
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(n):
            out = fn()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    def close(self):
        if self.world > 1 or self.force_comm:
            self.dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process (one rank per GPU, RCCL)
    and hand back its exit code.  The child inherits stdout / stderr, so rank 0's JSON line appears as this command's own.
    Never a re-exec: this process has not initialised the GPU and simply waits."""
    import socket
    import subprocess
    with socket.socket() as sk:                    # a free rendezvous port on the loopback interface
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def run_workload(env, args, scene, rays, scaling, steps, warmup, want_events, want_cpu_baseline, variant=None, grid_layout=None):
    """Measure one configuration.  scaling 'weak': `rays` per rank (batch = world x rays); 'strong': `rays` per
    iteration, split over the ranks in contiguous blocks (parallel.shard_range)."""
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd import parallel as PAR
    world, rank, dev = env.world, env.rank, env.dev
    stage = args.stage

    sc = build_scene_cpu(scene, seed=0)                            # identical replicas on every rank
    if variant == 'surfaces':                                      # analytic room instead of the random depth / colour images
        from evennicer_slam_amd.synthetic import BoxRoom
        room = BoxRoom.for_bound(sc['bound'], margin=0.7, seed=0)
        c4 = torch.eye(4, dtype=torch.float64)
        c4[:3] = sc['c2w'].double()
        col, dep = room.render(c4, sc['cam'])
        sc['color_img'], sc['depth_img'] = col.float(), dep
    if scaling == 'weak':                                          # rank r's block has its own seed; batch = all blocks
        blocks = [make_rays(sc, rays, seed=1000 + r) for r in range(world)]
        batch = [torch.cat([b[i] for b in blocks]) for i in range(4)]
        lo, hi = rank * rays, (rank + 1) * rays
    else:                                                          # one batch, contiguous block per rank
        batch = list(make_rays(sc, rays, seed=1000))
        lo, hi = PAR.shard_range(rays, rank, world)
    n_local, n_batch = hi - lo, batch[0].shape[0]
    rays_cpu = [t[lo:hi].contiguous() for t in batch]
    model = sc['model'].to(dev)
    attach_bounds(model, sc['bound'])
    # Feature grids: the reference's tensors ([1,32,D,H,W], same values and indexing) in torch's channels_last_3d memory format by
    # default -- their storage is then the [V][32] layout the gathers read, nothing is converted per step and the gradients come
    # back in the same format (functional.is_native_grid; the one-line change at grid_init is in INTEGRATION.md).
    # --grid-layout contiguous: the reference's own strides (what a caller who changes nothing passes; reported under `also`).
    grid_layout = grid_layout or args.grid_layout
    mf = torch.channels_last_3d if grid_layout == 'channels_last_3d' else torch.contiguous_format
    grids = {k: v.to(dev).contiguous(memory_format=mf).requires_grad_(True) for k, v in sc['grids'].items()}
    ro, rd, gd, gc = [t.to(dev) for t in rays_cpu]
    mapper_grads = variant == 'mapper_grads'
    if not mapper_grads:
        ro.requires_grad_(True)
        rd.requires_grad_(True)
    slam = types.SimpleNamespace(nice=True, bound=sc['bound'], **sc['cam'])
    renderer = E.Renderer(sc['cfg'], None, slam)
    kinds = EF.stage_kinds(stage)
    leaves = [grids[E._lib.GRID_NAMES[k]] for k in kinds]
    for k in kinds:
        ps = list(getattr(model, E._lib.MLP_NAMES[k]).parameters())
        if mapper_grads and E._lib.MLP_NAMES[k] != 'color_decoder':
            # the reference mapper's pattern (Mapper.py:363-369, configs/nice_slam.yaml:51-52): grids and the COLOUR decoder are
            # optimised, the occupancy decoders are fixed, rays carry no gradient outside bundle adjustment
            for q in ps:
                q.requires_grad_(False)
            continue
        leaves += ps

    fit = None
    if variant == 'surfaces':
        fit = fit_map(renderer, grids, model, sc, dev, iters=int(os.environ.get('ENSLAM_BENCH_FIT_ITERS', '400')))
    comm_on, force_comm = env.comm_on, env.force_comm
    # Ray sharding hands every rank the WHOLE batch (parallel.ShardedRenderer): each rank holds all ranks' rays, so the
    # batch maxima of gt_depth and the union of the touched 64-voxel blocks are computed locally -- one marking launch
    # over all rays before the local step -- and the gradient SUM is the only collective of a step.
    # ENSLAM_BENCH_EXCHANGE_FLAGS=1: the ranks only know their own rays (MAX all-reduces of the depth maximum and of the
    # block flags instead; per-rank marking cost independent of the rank count).
    whole_batch = comm_on and stage != 'coarse' and os.environ.get('ENSLAM_BENCH_EXCHANGE_FLAGS') != '1'
    ro_all = rd_all = gd_all = None
    if whole_batch:
        ro_all, rd_all, gd_all = [batch[i].to(dev) for i in range(3)]
    dmax_static = None
    if comm_on and stage != 'coarse':
        dmax_static = PAR.global_depth_max(gd_all if whole_batch else gd, force=force_comm and not whole_batch)
        renderer.depth_max_override = dmax_static
    prep = {'flags': None, 'prepared': None}
    # The pre-step of iteration k + 1 (batch maxima, block marking of the WHOLE batch, prefix sum, bucket-size read) depends on
    # nothing iteration k computes: it runs on a side stream under iteration k's gradient collective (two sets of flag buffers;
    # it waits for k's local step, whose sampler still reads the batch maxima, and k + 1's local step waits for it).
    # OFF by default (ENSLAM_BENCH_OVERLAP_PRE=1 switches it on): in the 1-rank RCCL rehearsal, where the collective is 31 us of
    # pack / unpack glue and hides nothing, the two cross-stream waits cost more than the 48 us pre-step they move (0.334 vs
    # 0.314 ms per step, two alternating pairs); with a real collective of 45-130 us it should pay -- unmeasured.
    overlap_pre = whole_batch and os.environ.get('ENSLAM_BENCH_OVERLAP_PRE', '0') == '1'
    pipe = {'side': torch.cuda.Stream() if overlap_pre else None, 'sets': [{'flags': None}, {'flags': None}], 'k': 0, 'ready': None}

    def pre_body(slot):
        torch.amax(gd_all, dim=0, keepdim=True, out=dmax_static[0:1])
        torch.mul(dmax_static[0:1], 1.2, out=dmax_static[1:2])
        slot['flags'] = PAR.batch_block_flags(renderer, grids, model, ro_all, rd_all, gd_all, stage, out=slot['flags'],
                                              block_voxels=BUCKET_BLOCK)
        slot['prepared'] = PAR.PreparedFlags([slot['flags'][id(t)] for t in leaves if t.dim() == 5], block_voxels=BUCKET_BLOCK,
                                             coarse=[slot['flags'][('c64', id(t))] for t in leaves if t.dim() == 5] if BUCKET_BLOCK != 64 else None)

    def pre_launch_next():      # (called right behind the local step of the current iteration)
        nxt = pipe['sets'][(pipe['k'] + 1) & 1]
        pipe['side'].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pipe['side']):
            pre_body(nxt)
            pipe['ready'] = torch.cuda.Event()
            pipe['ready'].record()

    def pre():          # what precedes the local step, outside the graph
        if overlap_pre:
            if pipe['ready'] is None:           # first iteration: nothing was launched ahead
                pre_body(pipe['sets'][pipe['k'] & 1])
            else:
                pipe['k'] += 1
                torch.cuda.current_stream().wait_event(pipe['ready'])
            cur = pipe['sets'][pipe['k'] & 1]
            prep['flags'], prep['prepared'] = cur['flags'], cur['prepared']
        elif whole_batch:
            pre_body(prep)
        elif dmax_static is not None:       # batch-global sampler maxima over all shards (tiny MAX all-reduce)
            PAR.global_depth_max(gd, force=force_comm, out=dmax_static)

    seed_grad, comm = {}, {'bytes': 0}

    def local_step():
        # a mapper iteration follows an optimiser step: grids and decoders have changed IN PLACE (same storage, new version
        # counters), so the voxel-major copies and the packed decoders are rebuilt every step (no caching credit in the timed
        # region); what an in-place update leaves valid -- argument checks, pointer tables -- stays cached, as in a real loop
        torch._C._increment_version(leaves)
        for t in leaves:
            t.grad = None
        ro.grad = None
        rd.grad = None
        if args.torch_loss or args.separate_loss or stage == 'coarse':
            depth, var, color = renderer.render_batch_ray(grids, model, rd, ro, dev, stage, gt_depth=gd)
            if args.torch_loss:
                loss = mapper_loss(depth, color, gd, gc, stage)
            else:
                loss = E.losses.rgbd_loss(depth, color if stage == 'color' else None, gd, gc, 0.2)
        else:               # the same render and loss with the loss folded into the compositing launches
            loss, depth, var, color = renderer.render_batch_ray_rgbd_loss(grids, model, rd, ro, dev, stage, gd, gc, 0.2)
        if 'one' not in seed_grad:              # d(loss)/d(loss) = 1, allocated once (backward() would fill one per step)
            seed_grad['one'] = torch.ones_like(loss)
        with EF.engine_on_calling_thread():       # scoped (the library's loops do the same); see `autograd_engine`
            loss.backward(gradient=seed_grad['one'])
        return loss

    def post():         # one bucketed RCCL all-reduce of the leaf gradients
        if comm_on and whole_batch:
            comm['bytes'] = PAR.allreduce_gradients(leaves, block_flags=prep['flags'], force=force_comm,
                                                    prepared=prep['prepared'])
        elif comm_on:
            comm['bytes'] = PAR.allreduce_gradients(leaves, block_flags=renderer.state.last_block_flags(), force=force_comm)

    def step():
        pre()
        loss = local_step()
        if overlap_pre:
            pre_launch_next()
        post()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()

    # per-kernel timing of the dominant kernel with HIP events on the launch stream (eager passes of the same step):
    # once as shipped (work list of the tiles with non-zero gradient) and once walking every tile (the dense figure)
    events = events_dense = None
    active = None

    def burst(n):
        # The Python-driven step is host-bound (~1 ms of launches for 0.3 ms of kernels): timed launch by launch, every kernel
        # would start on an idle GPU with cold clocks and a drained cache, which the timed region (graph replays, kernels back
        # to back) never sees.  So the stream is first held by a spin kernel while the host enqueues 10 steps; the GPU then
        # runs them back to back and the events bracket the kernel in the conditions of the timed region.
        for lo in range(0, n, 10):
            torch.cuda._sleep(int(3.0e7))
            for _ in range(min(10, n - lo)):
                step()
        torch.cuda.synchronize()

    if want_events:
        n_ev = min(steps, 50)
        renderer.state.profile['decoder_bwd'] = []
        burst(n_ev)
        events = renderer.state.profile.pop('decoder_bwd', None)
        active = renderer.state.last_active_tile_fraction()
        if renderer.state.use_work_list and stage != 'coarse':
            renderer.state.use_work_list = False
            try:
                for _ in range(3):
                    step()
                renderer.state.profile['decoder_bwd'] = []
                burst(n_ev)
                events_dense = renderer.state.profile.pop('decoder_bwd', None)
            finally:
                renderer.state.use_work_list = True
            for _ in range(3):
                step()
            torch.cuda.synchronize()

    # What an UNCHANGED caller gets: the reference API only (render_batch_ray -> the caller's torch loss -> loss.backward()),
    # Python-driven, no graph capture, no fused-loss entry point (Mapper.py:548-575 around this Renderer).
    api = None
    if world == 1 and stage != 'coarse' and not args.no_api:
        def api_step():
            torch._C._increment_version(leaves)         # (an optimiser step happened: values changed in place)
            for t in leaves:
                t.grad = None
            ro.grad = None
            rd.grad = None
            depth, var, color = renderer.render_batch_ray(grids, model, rd, ro, dev, stage, gt_depth=gd)
            loss = mapper_loss(depth, color, gd, gc, stage)
            loss.backward()
            return loss
        for _ in range(5):
            api_step()
        n_api = min(steps, 100)
        # (i) nothing changed at all: PyTorch's default autograd threading (the engine's worker thread runs our backward)
        # (median of three repeats: the hop to the engine's worker thread makes single repeats scatter between 0.5 and 0.85 ms by box)
        reps = []
        for _r in range(3):
            api_el, _l = env.timed(api_step, n_api)
            del _l
            reps.append(api_el)
        api_el = sorted(reps)[1]
        api = (n_local * n_api / api_el, api_el / n_api * 1e3)
        # (ii) the caller wraps its backward in `with EF.engine_on_calling_thread():` (one line)
        from evennicer_slam_amd import functional as _EF

        def api_step_ct():
            with _EF.engine_on_calling_thread():
                return api_step()
        for _ in range(3):
            api_step_ct()
        reps = []
        for _r in range(3):
            api_el2, _l = env.timed(api_step_ct, n_api)
            del _l
            reps.append(api_el2)
        api_el2 = sorted(reps)[1]
        api = api + (n_local * n_api / api_el2, api_el2 / n_api * 1e3)

    mode = 'eager'
    gstep = None
    phases = None
    if args.eager:
        elapsed, loss = env.timed(step, steps)
        eager_elapsed, eager_steps = elapsed, steps
    else:
        eager_steps = min(steps, 50)
        eager_elapsed, _ = env.timed(step, eager_steps)
        del _
        from evennicer_slam_amd.graph import GraphedStep
        for t in leaves:
            t.grad = None
        ro.grad = None
        rd.grad = None
        _gcmod.collect()            # no autograd graph of an earlier (default-stream) step may stay alive
        try:
            gstep = GraphedStep(local_step)
        except Exception as exc:    # never lose the measurement to a capture problem: time the eager loop instead
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); timing the eager step", file=sys.stderr)
            gstep = None
        if gstep is None:
            elapsed, loss = env.timed(step, steps)
        else:
            phases = [0.0, 0.0, 0.0] if os.environ.get('ENSLAM_BENCH_PHASES') == '1' else None

            def graph_step():
                if phases is None:
                    pre()
                    out = gstep.replay()
                    if overlap_pre:
                        pre_launch_next()
                    post()
                    return out
                # diagnostic: host-synchronised time of the three phases (changes the timing; not a measurement)
                for i, fn in enumerate((pre, gstep.replay, post)):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r = fn()
                    torch.cuda.synchronize()
                    phases[i] += time.perf_counter() - t0
                    if i == 1:
                        out = r
                return out

            mode = 'hipgraph'
            if comm_on:
                # Replays and eager collectives interleave on one stream; if that ever behaves badly on a node (it did
                # when two ranks of a rehearsal shared one GPU), fall back to the Python-driven step.  Both modes do
                # the same work; the choice is made on max-over-ranks times, so every rank takes the same branch.
                trial = min(10, steps)
                for _ in range(3):          # first use of the kernels only this mode runs (their code objects load on first launch)
                    graph_step()
                tg, _l = env.timed(graph_step, trial)
                te, _l = env.timed(step, trial)
                del _l
                if rank == 0 and os.environ.get('ENSLAM_BENCH_PHASES'):
                    print(f"[bench] trial: graph {tg / trial * 1e3:.3f} ms/step, eager {te / trial * 1e3:.3f} ms/step", file=sys.stderr)
                if te < tg:
                    mode = 'eager'
                    for t in leaves:
                        t.grad = None
            elapsed, loss = env.timed(graph_step if mode == 'hipgraph' else step, steps)

    # where a multi-rank step spends its time (HIP events around the three parts of 20 extra steps, this rank's stream;
    # outside the timed region): what precedes the local step, the local step, the gradient collective with its glue
    comm_ms = None
    if comm_on:
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(20)]
        local = (lambda: gstep.replay()) if (mode == 'hipgraph' and gstep is not None) else local_step
        torch.cuda.synchronize()
        for e in ev:                # (the three parts in line, one after the other: the pre-step is NOT overlapped here)
            e[0].record()
            if whole_batch:
                pre_body(prep)
            else:
                pre()
            e[1].record()
            local()
            e[2].record()
            post()
            e[3].record()
        torch.cuda.synchronize()
        comm_ms = {"pre_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in ev])),
                   "local_step_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in ev])),
                   "gradient_allreduce_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))}

    S = 48 if stage != 'coarse' else 32
    total_rays = n_batch                                            # rays all ranks processed in one step
    pretty = 'RPG' if scene == 'recording4' else 'Replica'
    per = f"{rays} rays x {S} samples per GPU" if scaling == 'weak' else f"{rays} rays x {S} samples per iteration split over {world} GPU(s)"
    out = {
        "metric": "rendered rays/sec (fwd+bwd)", "value": total_rays * steps / elapsed, "unit": "rays/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{pretty} {scene} full 4-level grid, stage {stage}, {per}, render_batch_ray + mapper loss + "
                               f"backward (grads: grids, all decoder params, rays)",
                   "rays_per_gpu": n_local, "rays_per_step": total_rays, "samples_per_ray": S, "grid_memory_format": grid_layout,
                   "parallelism": "1 GPU" if world == 1 else f"ray-sharded dp{world}, one bucketed RCCL all-reduce of leaf grads "
                                                            f"(touched 64-voxel blocks only)"},
        "loss": float(loss.item()), "mode": mode, "loss_impl": "torch" if args.torch_loss else ("fused HIP (losses.rgbd_loss)" if args.separate_loss or stage == 'coarse'
                                                                  else "fused into the compositing launches (render_batch_ray_rgbd_loss)"),
        "eager_rays_per_s": total_rays * eager_steps / eager_elapsed,
        "autograd_engine": ("headline (hipGraph replay): no autograd at replay time; eager_rays_per_s and the capture: backward inside "
                            "`with functional.engine_on_calling_thread()` (scoped, restored on exit); api_rays_per_s: PyTorch's "
                            "default threading; api_calling_thread_rays_per_s: the scoped form"),
    }
    if api is not None:
        out["api_rays_per_s"], out["api_ms_per_step"] = api[0], api[1]
        out["api_repeats"] = "median of 3 x %d steps" % min(steps, 100)
        out["api_step"] = ("render_batch_ray + torch L1 losses + loss.backward(), Python-driven (no hipGraph, no fused-loss entry), "
                           "PyTorch's default autograd threading")
        out["api_calling_thread_rays_per_s"], out["api_calling_thread_ms_per_step"] = api[2], api[3]
        out["api_calling_thread_step"] = "the same with `with functional.engine_on_calling_thread(): loss.backward()`"
    if fit is not None:
        out["fit"] = fit
    if comm_on:
        out["comm"] = dict(comm_ms, bucket_bytes=int(comm['bytes']), mode=mode,
                           flags="marked locally from the whole batch" if whole_batch else "MAX all-reduce of per-rank flags",
                           pre_step="side stream, under the previous iteration's collective" if overlap_pre else "in line",
                           bucket_block_voxels=BUCKET_BLOCK if whole_batch else 64)
    if events:
        dur = np.array([a.elapsed_time(b) for a, b in events]) * 1e-3          # seconds
        avg = float(dur.mean())
        n_points = n_local * S
        flops = n_points * 2 * FLOP_FWD_PER_POINT[stage]                        # dX + dW of every linear layer
        step_flops = n_local * S * 3 * FLOP_FWD_PER_POINT[stage]
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes + WRITE_SIZE); they are NOT measured in this run
        traffic, tsrc = None, None
        for tname in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and scene == 'room0' and n_local == 1000 and stage == 'color':
                traffic = json.load(open(tpath)).get("decoder_bwd_split_kernel", {}).get("bytes_per_launch")
                tsrc = f"profiles/{tname} (separate --pmc passes of this command; not measured in this run)"
                break
        frac = flops / avg / PEAK_F32_MFMA
        out["roofline"] = {
            "kernel": "decoder_bwd_split_kernel", "bound": "mfma", "achieved": flops / avg / 1e12, "peak": PEAK_F32_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": frac, "traffic": traffic, "traffic_source": tsrc,
            "avg_launch_us": avg * 1e6, "launches": int(len(dur)),
            # `frac` prices the yardstick FLOPs of EVERY sample against the shipped kernel, which does not schedule the
            # tiles whose gradient is exactly zero (transmittance underflowed behind the room's boundary).  The two
            # figures below separate that algorithmic skipping from kernel efficiency:
            #   frac_executed = frac x active_tile_fraction : FLOPs of the tiles the kernel actually walked / time
            #   frac_dense    : the same kernel made to walk every tile (ENSLAM_WORK_LIST=0), yardstick FLOPs / its time
            "active_tile_fraction": active,
            "frac_executed": frac * active if active is not None else None,
            "step_frac": step_flops / PEAK_F32_MFMA / (elapsed / steps),
        }
        if events_dense:
            dd = float(np.mean([a.elapsed_time(b) for a, b in events_dense])) * 1e-3
            out["roofline"]["frac_dense"] = flops / dd / PEAK_F32_MFMA
            out["roofline"]["avg_launch_us_dense"] = dd * 1e6
    if want_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sc, rays_cpu, stage)
    if rank == 0 and phases is not None:
        print("[bench] phases ms/step: depth-max all-reduce %.3f, graph replay %.3f, gradient all-reduce %.3f"
              % tuple(p / steps * 1e3 for p in phases), file=sys.stderr)
    # release this workload's device memory before the next one is built
    del gstep, grids, model, leaves, renderer
    EF.clear_caches()
    _gcmod.collect()
    torch.cuda.empty_cache()
    return out


def _timed_plain(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(n):
        out = None
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def run_config3(dev, steps=20):
    """BASELINE config 3: the tracker's camera iteration with the event term on room0 at Replica resolution --
    200-pixel RGB-D batch + render_img_rescale (0.15 x 680 x 1200 = 18 360 rays x 48, gradients to the pose; src/Tracker.py:150)
    -> PyTorch-ROCm UNet_2heads(6,2,2) (seeded random weights: the checkpoint is not in the image) -> blurred-L2 event loss
    (kernel 9, balancer 0.025; Tracker.py:206-232) -> backward to the 7 pose numbers -> Adam.  Path alone and path + U-Net are
    timed separately, Python-driven with the reference's .item() syncs, and as one hipGraph per iteration."""
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd.mapper import FusedAdam
    sc = build_scene_cpu('room0', 0)
    model = sc['model'].to(dev)
    attach_bounds(model, sc['bound'])
    for q in model.parameters():
        q.requires_grad_(False)
    grids = {k: v.to(dev) for k, v in sc['grids'].items()}
    renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **CAM))
    H, W = CAM['H'], CAM['W']
    g = torch.Generator().manual_seed(1)
    depth_img = (torch.rand(H, W, generator=g) * 3.0 + 0.5).to(dev)
    color_img = torch.rand(H, W, 3, generator=g).to(dev)
    pre_color = torch.rand(H, W, 3, generator=g).to(dev)
    gt_event = torch.randint(0, 4, (H, W, 2), generator=g).float().to(dev)
    gt_mask = (gt_event.sum(-1) > 2).long()
    cfg = dict(sc['cfg'])
    cfg['tracking'] = {'device': str(dev), 'w_color_loss': 0.5, 'ignore_edge_W': 100, 'ignore_edge_H': 100, 'handle_dynamic': True,
                       'use_color_in_tracking': True}
    cfg['event'] = {'activate_events': True, 'blur': True, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                    'balancer': 0.025}
    torch.manual_seed(0)
    net = E.event.UNet_2heads(6, 2, 2)
    for q in net.parameters():
        q.requires_grad_(False)
    net = net.to(dev).eval()
    slam = types.SimpleNamespace(nice=True, bound=sc['bound'], renderer=renderer, event_net=net, low_gpu_mem=False, **CAM)
    trk = E.tracker.TrackerIteration(cfg, None, slam)
    trk.c, trk.decoders = grids, model
    ct = torch.tensor([1.0, 0.0, 0.0, 0.0, 3.0, 1.0, 0.0], device=dev, requires_grad=True)
    opt = FusedAdam([ct], lr=1e-3)
    SF = 0.15
    rays = int(H * SF) * int(W * SF)

    def full():
        return trk.optimize_cam_in_batch(ct, None, color_img, depth_img, gt_event, gt_mask, 200, opt, 0, 0, pre_color, rgbd=True,
                                         event=True, scale_factor=SF)

    def path_only():
        opt.zero_grad()
        col = trk._render_rescaled(ct, depth_img, SF)
        col.sum().backward()
        opt.step()

    def unet_only():
        x = torch.rand(1, 6, int(H * SF), int(W * SF), device=dev, requires_grad=True)
        e, m = net(x)
        (e.sum() + m.sum()).backward()

    for f in (full, path_only, unet_only):
        for _ in range(3):
            f()
    losses = [float(x) for x in full()[:2]]
    t_full, _o = _timed_plain(full, steps)
    t_path, _o = _timed_plain(path_only, steps)
    t_unet, _o = _timed_plain(unet_only, steps)
    del _o
    out = {"workload": f"Replica room0, tracker camera iteration with the event term: 200-ray RGB-D batch + {rays}-ray x 48 "
                       f"render_img_rescale (scale 0.15) with gradient to the pose + UNet_2heads(6,2,2) fp32 (random weights) + "
                       f"blurred-L2 event loss + Adam",
           "rays": rays, "path_ms": t_path * 1e3, "path_rays_per_s": rays / t_path, "unet_ms": t_unet * 1e3,
           "path_plus_unet_ms": t_full * 1e3, "loss_rgbd": losses[0], "loss_event": losses[1]}
    try:
        _gcmod.collect()
        git = E.tracker.GraphedCameraIteration(trk, ct, opt, color_img, depth_img, gt_event, gt_mask, pre_color, batch_size=200,
                                               rgbd=True, event=True, scale_factor=SF)
        for _ in range(3):
            git.step()
        tg, _o = _timed_plain(git.step, steps * 2)
        del _o
        out["graphed_ms"] = tg * 1e3
        del git
    except Exception as exc:            # the capture is an optimisation of the caller's loop, not the measurement
        out["graphed_ms"] = None
        out["graphed_error"] = f"{type(exc).__name__}: {exc}"
    del trk, net, grids, model, renderer
    EF.clear_caches()
    _gcmod.collect()
    torch.cuda.empty_cache()
    return out


def run_tracker_iter(dev, n_rays=200, steps=200):
    """The tracker's RGB-D camera iteration as the run harness issues it (tracker.TrackerIteration / GraphedCameraIteration:
    Tracker.py:141-197 with the reference's defaults -- in-bound prefilter and `handle_dynamic` median mask on -- plus the
    optimiser step) on room0, colour stage, map and decoders fixed: camera tensor -> pose -> n_rays random pixels -> render ->
    masked uncertainty-weighted loss -> backward to the 7 pose numbers -> Adam; one hipGraph per iteration (VERDICT r2
    item 7: the small-batch floor).  `graphed_us_no_median`: the same with handle_dynamic off."""
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd.mapper import FusedAdam
    sc = build_scene_cpu('room0', 0)
    model = sc['model'].to(dev)
    attach_bounds(model, sc['bound'])
    for q in model.parameters():
        q.requires_grad_(False)
    grids = {k: v.to(dev).contiguous(memory_format=torch.channels_last_3d) for k, v in sc['grids'].items()}
    slam = types.SimpleNamespace(nice=True, bound=sc['bound'], event_net=None, low_gpu_mem=False, **CAM)
    slam.renderer = E.Renderer(sc['cfg'], None, slam)
    H, W = CAM['H'], CAM['W']
    g = torch.Generator().manual_seed(1)
    depth_img = (torch.rand(H, W, generator=g) * 3.0 + 0.5).to(dev)
    color_img = torch.rand(H, W, 3, generator=g).to(dev)
    out = {"workload": f"Replica room0, tracker RGB-D camera iteration as the harness runs it (in-bound prefilter, handle_dynamic median "
                       f"mask, uncertainty-weighted loss), {n_rays} rays x 48, colour stage, fixed map, gradient to the pose + Adam",
           "rays": n_rays}
    for dyn in (True, False):
        cfg = dict(sc['cfg'])
        cfg['tracking'] = {'device': dev, 'w_color_loss': 0.5, 'ignore_edge_W': 100, 'ignore_edge_H': 100, 'handle_dynamic': dyn,
                           'use_color_in_tracking': True, 'lr': 1e-3, 'pixels': n_rays, 'iters': 10}
        cfg['event'] = {'activate_events': False, 'blur': False, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                        'balancer': 0.025}
        trk = E.tracker.TrackerIteration(cfg, None, slam)
        trk.c, trk.decoders = grids, model
        ct = torch.tensor([1.0, 0.0, 0.0, 0.0, 3.0, 1.0, 0.0], device=dev, requires_grad=True)
        opt = FusedAdam([ct], lr=1e-3)
        if dyn:                         # the Python-driven form of the same iteration (optimize_cam_in_batch: one host read per call)
            for _ in range(5):
                trk.optimize_cam_in_batch(ct, None, color_img, depth_img, None, None, n_rays, opt, 1, 0, None, rgbd=True, event=False)
            te, _l = _timed_plain(lambda: trk.optimize_cam_in_batch(ct, None, color_img, depth_img, None, None, n_rays, opt, 1, 0, None,
                                                                    rgbd=True, event=False)[0], 30)
            out["eager_us"] = te * 1e6
            opt.zero_grad()
            _gcmod.collect()
        gi = E.tracker.GraphedCameraIteration(trk, ct, opt, color_img, depth_img, batch_size=n_rays, rgbd=True, event=False)
        tg, losses = _timed_plain(gi.step, steps)
        key = "graphed_us" if dyn else "graphed_us_no_median"
        out[key] = tg * 1e6
        if dyn:
            out["rays_per_s"] = n_rays / tg
            out["loss"] = float(losses[0].item())
        del gi, trk, opt, losses
        _gcmod.collect()
    del grids, model, slam
    EF.clear_caches()
    _gcmod.collect()
    torch.cuda.empty_cache()
    return out


def run_slam_fps(dev, frames=10):
    """BASELINE's second metric ("Replica room0 tracking+mapping FPS") on SYNTHETIC data with random-weight networks -- datasets and
    pretrained weights are not in the image, so this times the schedule's compute, not accuracy: room0 grids, Replica camera, the
    shipped schedule (configs/Replica/replica.yaml + configs/nice_slam.yaml): per frame 10 camera iterations (event term every frame,
    the RGB-D term on every 5th), every 5th frame 60 mapper iterations of 1000 rays (colour stage, frustum-masked grids + colour
    decoder).  One process, one GPU: the reference's tracker and mapper processes alternate here.  Tracker:
    tracker.GraphedCameraIteration; mapper: MaskedGridOptimizer + FusedAdam, one hipGraph per iteration."""
    import copy
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd.graph import GraphedStep
    from evennicer_slam_amd.mapper import FusedAdam, MaskedGridOptimizer
    KEYS = ('grid_middle', 'grid_fine', 'grid_color')
    sc = build_scene_cpu('room0', 0)
    H, W = CAM['H'], CAM['W']
    g = torch.Generator().manual_seed(1)
    fr = []
    for _ in range(4):
        ev = torch.randint(0, 4, (H, W, 2), generator=g).float()
        fr.append(dict(depth=(torch.rand(H, W, generator=g) * 3.0 + 0.5).to(dev), color=torch.rand(H, W, 3, generator=g).to(dev),
                       event=ev.to(dev), mask=(ev.sum(-1) > 2).long().to(dev)))
    model = copy.deepcopy(sc['model']).to(dev)
    attach_bounds(model, sc['bound'])
    for name in ('coarse_decoder', 'middle_decoder', 'fine_decoder'):
        for q in getattr(model, name).parameters():
            q.requires_grad_(False)
    grids = {k: v.to(dev).contiguous(memory_format=torch.channels_last_3d) for k, v in sc['grids'].items()}
    masks = {}
    for k in KEYS:
        D, Hh, Ww = grids[k].shape[2:]
        m = torch.zeros(D, Hh, Ww, dtype=torch.bool)
        m[:, :, Ww // 4: 3 * Ww // 4] = True
        masks[k] = m.to(dev)
    renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **CAM))
    opt = MaskedGridOptimizer(grids, masks, keys=KEYS)
    dec_opt = FusedAdam(list(model.color_decoder.parameters()), lr=0.005)
    opt.set_lr({k: 0.005 for k in KEYS})
    c_map = opt.render_grids()
    m_ro, m_rd, m_gd, m_gc = [t.to(dev) for t in make_rays(sc, 1000, 1000)]
    one = {}

    def map_it():
        dec_opt.zero_grad()
        loss, _d, _v, _c = renderer.render_batch_ray_rgbd_loss(c_map, model, m_rd, m_ro, dev, 'color', m_gd, m_gc, 0.2)
        if 'one' not in one:
            one['one'] = torch.ones_like(loss)
        loss.backward(gradient=one['one'])
        dec_opt.step()
        opt.step()
        return loss

    for _ in range(3):
        map_it()
    dec_opt.zero_grad()
    _gcmod.collect()
    g_map = GraphedStep(map_it)
    t_model = copy.deepcopy(model)
    for q in t_model.parameters():
        q.requires_grad_(False)
    attach_bounds(t_model, sc['bound'])
    t_grids = {k: v.detach().clone() for k, v in grids.items()}
    cfg = dict(sc['cfg'])
    cfg['tracking'] = {'device': dev, 'w_color_loss': 0.5, 'ignore_edge_W': 100, 'ignore_edge_H': 100, 'handle_dynamic': True,
                       'use_color_in_tracking': True, 'iters': 10}
    cfg['event'] = {'activate_events': True, 'blur': True, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                    'balancer': 0.025}
    torch.manual_seed(0)
    net = E.event.UNet_2heads(6, 2, 2)
    for q in net.parameters():
        q.requires_grad_(False)
    net = net.to(dev).eval()
    slam = types.SimpleNamespace(nice=True, bound=sc['bound'], event_net=net, low_gpu_mem=False, **CAM)
    slam.renderer = E.Renderer(sc['cfg'], None, slam)
    trk = E.tracker.TrackerIteration(cfg, None, slam)
    trk.c, trk.decoders = t_grids, t_model
    ct = torch.tensor([1.0, 0.0, 0.0, 0.0, 3.0, 1.0, 0.0], device=dev, requires_grad=True)
    cam_opt = FusedAdam([ct], lr=1e-3)
    f0 = fr[0]
    kw = dict(batch_size=200, scale_factor=0.15)
    git_full = E.tracker.GraphedCameraIteration(trk, ct, cam_opt, f0['color'], f0['depth'], f0['event'], f0['mask'], f0['color'], rgbd=True,
                                                event=True, **kw)
    git_ev = E.tracker.GraphedCameraIteration(trk, ct, cam_opt, f0['color'], f0['depth'], f0['event'], f0['mask'], f0['color'], rgbd=False,
                                              event=True, **kw)

    def update_para_from_mapping():                     # Tracker.py:247-260, in place
        opt.write_back()
        with torch.no_grad():
            for k in KEYS:
                t_grids[k].copy_(grids[k])
            for pt, pm in zip(t_model.parameters(), model.parameters()):
                pt.copy_(pm)
        git_full.refresh_map()

    for _ in range(60):                                 # (a first mapping round outside the timed frames)
        g_map.replay()
    update_para_from_mapping()
    torch.cuda.synchronize()
    t_track = t_map = 0.0
    t_all = time.perf_counter()
    for i in range(1, frames + 1):
        f, prev = fr[i % 4], fr[(i - 1) % 4]
        a = time.perf_counter()
        git = git_full if i % 5 == 0 else git_ev
        git.set_frame(f['color'], f['depth'], f['event'], f['mask'], prev['color'])
        for _ in range(10):
            git.step()
        torch.cuda.synchronize()
        b = time.perf_counter()
        t_track += b - a
        if i % 5 == 0:                                  # mapping.every_frame 5
            for _ in range(60):
                g_map.replay()
            update_para_from_mapping()
            torch.cuda.synchronize()
            t_map += time.perf_counter() - b
    t_all = time.perf_counter() - t_all
    out = {"workload": "synthetic Replica-schedule run, room0, 1 GPU, one process: per frame 10 camera iterations (event term: 18360-ray "
                       "render + UNet_2heads with random weights + blurred-L2; RGB-D term every 5th frame), every 5th frame 60 mapper "
                       "iterations of 1000 rays (colour stage) + the tracker's map update; synthetic images, no dataset",
           "frames": frames, "frames_per_s": frames / t_all, "tracking_ms_per_frame": t_track / frames * 1e3,
           "mapping_ms_per_round": t_map / max(frames // 5, 1) * 1e3}
    del g_map, git_full, git_ev, trk, opt, dec_opt, grids, t_grids, model, t_model, net
    EF.clear_caches()
    _gcmod.collect()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS), help='measurement configuration (see module docstring)')
    ap.add_argument('--rays', type=int, default=None, help='override the configuration\'s ray count (per GPU if weak, per step if strong)')
    ap.add_argument('--scene', default=None, help='override the configuration\'s scene (room0, office0, recording4)')
    ap.add_argument('--scaling', default=None, choices=('weak', 'strong'))
    ap.add_argument('--stage', default='color')
    ap.add_argument('--no-secondary', action='store_true', help='default run: skip the extra config-4 measurement reported under "also"')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--no-api', action='store_true', help='skip the reference-API-only (Python-driven) timing `api_rays_per_s`')
    ap.add_argument('--variant', default=None, choices=('surfaces', 'mapper_grads'), help="primary workload on a map fitted to an analytic room | with the reference mapper's gradient set")
    ap.add_argument('--grid-layout', default='channels_last_3d', choices=('channels_last_3d', 'contiguous'),
                    help='memory format of the feature-grid tensors (same shape and values either way)')
    ap.add_argument('--torch-loss', action='store_true', help='compute the mapper loss with torch ops instead of the fused HIP loss')
    ap.add_argument('--separate-loss', action='store_true', help='render_batch_ray, then losses.rgbd_loss as its own launches')
    ap.add_argument('--eager', action='store_true', help='time the plain Python-driven step instead of hipGraph replays')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and os.environ.get('ENSLAM_BENCH_FORCE_COMM') != '1':
        sys.exit(spawn_ranks(args.gpus))        # (nothing in this process has touched the GPU yet)
    env = Env(args)
    spec = dict(CONFIGS[args.config])
    customised = args.rays is not None or args.scene is not None or args.scaling is not None
    scene = args.scene or spec['scene']
    rays = args.rays if args.rays is not None else spec['rays']
    scaling = args.scaling or spec['scaling']
    primary = run_workload(env, args, scene, rays, scaling, args.steps, args.warmup,
                           want_events=not args.no_kernel_events,
                           want_cpu_baseline=env.rank == 0 and env.world == 1 and not args.no_cpu_baseline, variant=args.variant)
    primary["config"]["baseline_config"] = spec['name'] if not customised else "custom"
    if args.config == 2 and not customised and not args.no_secondary and args.stage == 'color':
        s4 = CONFIGS[4]
        sec = run_workload(env, args, s4['scene'], s4['rays'], s4['scaling'], max(20, args.steps // 2), max(5, args.warmup // 2),
                           want_events=False, want_cpu_baseline=False)
        keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "mode", "loss", "comm", "eager_rays_per_s")
        primary["also"] = {"config4": dict({k: sec[k] for k in keep if k in sec}, workload=sec["config"]["workload"],
                                           rays_per_gpu=sec["config"]["rays_per_gpu"])}
        if env.world == 1:
            # further single-GPU measurements of the default run (each guarded: a failure is reported, never fatal)
            def guarded(name, fn):
                try:
                    primary["also"][name] = fn()
                except Exception as exc:
                    primary["also"][name] = {"error": f"{type(exc).__name__}: {exc}"}

            def surfaces():
                r = run_workload(env, args, scene, rays, scaling, max(50, args.steps // 2), max(5, args.warmup // 2),
                                 want_events=not args.no_kernel_events, want_cpu_baseline=False, variant='surfaces')
                o = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "mode", "loss", "fit", "api_rays_per_s") if k in r}
                rf = r.get("roofline") or {}
                o.update({k: rf.get(k) for k in ("active_tile_fraction", "avg_launch_us", "frac", "frac_executed", "step_frac")})
                o["workload"] = ("config 2 on a map with surfaces: room0 grids and decoders fitted (through this path) to an analytic "
                                 "room (box room + one box, smooth view-consistent colours) seen from the bench camera")
                return o

            def contiguous_grids():
                r = run_workload(env, args, scene, rays, scaling, max(50, args.steps // 2), max(5, args.warmup // 2),
                                 want_events=False, want_cpu_baseline=False, grid_layout='contiguous')
                o = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "mode", "loss", "api_rays_per_s", "api_ms_per_step",
                                       "eager_rays_per_s") if k in r}
                o["workload"] = ("config 2 with the feature grids in the reference's own strides (a caller that changes nothing, not even "
                                 "the memory format at grid_init): touched blocks converted to [V][32] before the forward and the "
                                 "gradients transposed back after the backward, every step")
                return o

            def mapper_grads():
                r = run_workload(env, args, scene, rays, scaling, max(50, args.steps // 2), max(5, args.warmup // 2),
                                 want_events=not args.no_kernel_events, want_cpu_baseline=False, variant='mapper_grads')
                o = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "mode", "loss", "eager_rays_per_s") if k in r}
                rf = r.get("roofline") or {}
                o.update({"decoder_bwd_" + k: rf.get(k) for k in ("avg_launch_us", "frac", "step_frac")})
                o["workload"] = ("config 2 with the gradients the reference's mapper asks for (Mapper.py:363-369): feature grids and the colour "
                                 "decoder's parameters only -- no gradients for the middle / fine decoders' parameters (their backward runs the "
                                 "light chain kernel), none for the rays (no ray-gradient role in the finish launch).  frac / step_frac stay "
                                 "on the agreed yardstick (3 x 103 306 FLOP per point), of which this step needs 2.30 x")
                # the same step with the feature-gradient scatter as a launch of its own (csrc/grid_scatter.hip; opt-in, read once per
                # process: a child process).  Wins on this random-init map, loses on a map with surfaces
                try:
                    import subprocess
                    cmd = [sys.executable, os.path.abspath(__file__), '--variant', 'mapper_grads', '--steps', str(max(50, args.steps // 2)),
                           '--warmup', str(max(5, args.warmup // 2)), '--no-secondary', '--no-cpu-baseline', '--no-api', '--no-kernel-events']
                    cp = subprocess.run(cmd, env=dict(os.environ, ENSLAM_DEFER_SCATTER='2'), capture_output=True, text=True, timeout=300)
                    line = [ln for ln in cp.stdout.splitlines() if ln.startswith('{')][-1]
                    c = json.loads(line)
                    o["deferred_scatter"] = {"switch": "ENSLAM_DEFER_SCATTER=2", "value": c["value"], "unit": c["unit"],
                                             "ms_per_step": c["ms_per_step"], "loss": c.get("loss")}
                except Exception as e:                                       # (never fails the bench line)
                    o["deferred_scatter"] = {"error": repr(e)[:200]}
                return o

            guarded("config2_mapper_grads", mapper_grads)
            if args.grid_layout != 'contiguous':
                guarded("config2_contiguous_grids", contiguous_grids)
            guarded("config2_surfaces", surfaces)
            guarded("config3", lambda: run_config3(env.dev, steps=max(10, min(30, args.steps // 10))))
            guarded("tracker_iter_200", lambda: run_tracker_iter(env.dev, 200, steps=max(50, args.steps)))
            guarded("slam_fps_synthetic", lambda: run_slam_fps(env.dev, frames=10))
    if env.rank == 0:
        print(json.dumps(primary), flush=True)
    env.close()


if __name__ == '__main__':
    main()
