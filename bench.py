#!/usr/bin/env python3
"""Benchmark of the volume-rendering hot path (BASELINE.json metric: rendered rays/sec, fwd+bwd).

One "step" = Renderer.render_batch_ray (sampling, gather, decoders, compositing) + the mapper's loss
(Mapper.py:553-562) + backward producing gradients for the three feature grids, EVERY decoder parameter
(the reference never freezes any, so its autograd computes them all) and the rays -- on synthetic data of
BASELINE configs[1]: Replica room0, full 4-level grid, stage `color`, 1000 rays x 48 samples per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; rays are sharded, i.e. every rank renders its
   own 1000-ray block (weak scaling), and leaf gradients are summed with one bucketed RCCL all-reduce.)

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the `roofline` and `cpu_baseline` objects).
"""
import argparse
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
GRID_LEN = {'coarse': 2, 'middle': 0.32, 'fine': 0.16, 'color': 0.16, 'bound_divisible': 0.32}


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
    # 64-core EPYC of the GPU box), so a few thread counts are timed and the best is the baseline.
    n = ro.shape[0]
    all_threads = torch.get_num_threads()
    results = {}
    try:
        for nt in sorted({1, 8, 32, all_threads}):
            if nt > all_threads:
                continue
            torch.set_num_threads(nt)
            step()                                          # warm-up at this thread count
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
    model_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            model_name = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    others = ", ".join(f"{k} thr {n / v[0]:.0f}" for k, v in sorted(results.items()))
    return {"value": n / med, "unit": "rays/s", "cores": best, "kind": "port", "cpu": model_name,
            "value_1_thread": n / results[1][0] if 1 in results else None,
            "sample": f"{n} rays x 48 samples, stage {stage}, fwd+loss+bwd, torch CPU oracle; median of {iters} iterations after "
                      f"1 warm-up at the best of the thread counts tried (rays/s: {others})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--rays', type=int, default=1000, help='rays per GPU per step')
    ap.add_argument('--stage', default='color')
    ap.add_argument('--scene', default='room0')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--torch-loss', action='store_true', help='compute the mapper loss with torch ops instead of the fused HIP loss')
    ap.add_argument('--separate-loss', action='store_true', help='render_batch_ray, then losses.rgbd_loss as its own launches')
    ap.add_argument('--eager', action='store_true', help='time the plain Python-driven step instead of hipGraph replays')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU implementation")
    # rehearsal on a one-GPU box: ENSLAM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and ENSLAM_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device); the measured runs use neither
    if os.environ.get('ENSLAM_BENCH_SHARE_GPU') == '1':
        local_rank = 0
    backend = os.environ.get('ENSLAM_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    # rehearsal of the communication path on ONE rank with real RCCL (collectives of a 1-rank group): not a measurement
    force_comm = world == 1 and os.environ.get('ENSLAM_BENCH_FORCE_COMM') == '1'
    if force_comm:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29777')
        dist.init_process_group(backend, rank=0, world_size=1, **({'device_id': dev} if backend == 'nccl' else {}))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from evennicer_slam_amd import parallel as PAR

    sc = build_scene_cpu(args.scene, seed=0)                      # identical replicas on every rank
    rays_cpu = make_rays(sc, args.rays, seed=1000 + rank)         # each rank renders its own block of the batch
    model = sc['model'].to(dev)
    attach_bounds(model, sc['bound'])
    grids = {k: v.to(dev).requires_grad_(True) for k, v in sc['grids'].items()}
    ro, rd, gd, gc = [t.to(dev) for t in rays_cpu]
    ro.requires_grad_(True)
    rd.requires_grad_(True)
    slam = types.SimpleNamespace(nice=True, bound=sc['bound'], **sc['cam'])
    renderer = E.Renderer(sc['cfg'], None, slam)
    stage = args.stage
    kinds = EF.stage_kinds(stage)
    leaves = [grids[E._lib.GRID_NAMES[k]] for k in kinds]
    for k in kinds:
        leaves += list(getattr(model, E._lib.MLP_NAMES[k]).parameters())

    comm_on = world > 1 or force_comm
    # Ray sharding hands every rank the WHOLE batch (parallel.ShardedRenderer): each rank holds all ranks' rays, so the
    # batch maxima of gt_depth and the union of the touched 64-voxel blocks are computed locally -- one marking launch
    # over all rays before the local step -- and the gradient SUM is the only collective of a step.
    # ENSLAM_BENCH_EXCHANGE_FLAGS=1: the ranks only know their own rays (MAX all-reduces of the depth maximum and of the
    # block flags instead).
    whole_batch = comm_on and stage != 'coarse' and os.environ.get('ENSLAM_BENCH_EXCHANGE_FLAGS') != '1'
    ro_all = rd_all = gd_all = None
    if whole_batch:
        others = [make_rays(sc, args.rays, seed=1000 + r) for r in range(world)]
        ro_all, rd_all, gd_all = [torch.cat([o[i] for o in others]).to(dev) for i in range(3)]
    dmax_static = None
    if comm_on and stage != 'coarse':
        dmax_static = PAR.global_depth_max(gd_all if whole_batch else gd, force=force_comm and not whole_batch)
        renderer.depth_max_override = dmax_static
    prep = {'flags': None, 'prepared': None}

    def pre():          # what precedes the local step, outside the graph
        if whole_batch:
            torch.amax(gd_all, dim=0, keepdim=True, out=dmax_static[0:1])
            torch.mul(dmax_static[0:1], 1.2, out=dmax_static[1:2])
            prep['flags'] = PAR.batch_block_flags(renderer, grids, model, ro_all, rd_all, gd_all, stage, out=prep['flags'])
            prep['prepared'] = PAR.PreparedFlags([prep['flags'][id(t)] for t in leaves if t.dim() == 5])
        elif dmax_static is not None:       # batch-global sampler maxima over all shards (tiny MAX all-reduce)
            PAR.global_depth_max(gd, force=force_comm, out=dmax_static)

    seed_grad, comm = {}, {'bytes': 0}

    def local_step():
        # a mapper iteration follows an optimiser step: grids and decoders have changed, so the voxel-major
        # copies and the packed decoders are rebuilt every step (no caching credit in the timed region)
        EF.clear_caches()
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
        loss.backward(gradient=seed_grad['one'])
        return loss

    def post():         # one bucketed RCCL all-reduce of the leaf gradients
        if comm_on and whole_batch:
            comm['bytes'] = PAR.allreduce_gradients(leaves, block_flags=prep['flags'], force=force_comm,
                                                    prepared=prep['prepared'])
        elif comm_on:
            comm['bytes'] = PAR.allreduce_gradients(leaves, block_flags=EF.last_block_flags(), force=force_comm)

    def step():
        pre()
        loss = local_step()
        post()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # per-kernel timing of the dominant kernel with HIP events on the launch stream (eager pass of the same step)
    events = None
    if not args.no_kernel_events:
        EF.PROFILE['decoder_bwd'] = []
        for _ in range(min(args.steps, 50)):
            step()
        torch.cuda.synchronize()
        events = EF.PROFILE.pop('decoder_bwd', None)

    # eager (Python-driven) rate, always reported; the timed region below is graph replay unless --eager
    def timed(fn, n):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    mode = 'eager'
    if args.eager:
        elapsed, loss = timed(step, args.steps)
        eager_elapsed, eager_steps = elapsed, args.steps
    else:
        eager_steps = min(args.steps, 50)
        eager_elapsed, _ = timed(step, eager_steps)
        del _
        from evennicer_slam_amd.graph import GraphedStep
        import gc as _gcmod
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
            elapsed, loss = timed(step, args.steps)
        else:
            phases = [0.0, 0.0, 0.0] if os.environ.get('ENSLAM_BENCH_PHASES') == '1' else None

            def graph_step():
                if phases is None:
                    pre()
                    out = gstep.replay()
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
                trial = min(10, args.steps)
                tg, _l = timed(graph_step, trial)
                te, _l = timed(step, trial)
                del _l
                if rank == 0 and os.environ.get('ENSLAM_BENCH_PHASES'):
                    print(f"[bench] trial: graph {tg / trial * 1e3:.3f} ms/step, eager {te / trial * 1e3:.3f} ms/step", file=sys.stderr)
                if te < tg:
                    mode = 'eager'
                    for t in leaves:
                        t.grad = None
            elapsed, loss = timed(graph_step if mode == 'hipgraph' else step, args.steps)

    S = 48 if stage != 'coarse' else 32
    n_points = args.rays * S
    out = {
        "metric": "rendered rays/sec (fwd+bwd)", "value": world * args.rays * args.steps / elapsed, "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{'RPG' if args.scene == 'recording4' else 'Replica'} {args.scene} full 4-level grid, stage {stage}, {args.rays} rays x {S} samples "
                               f"per GPU, render_batch_ray + mapper loss + backward (grads: grids, all decoder params, rays)",
                   "rays_per_gpu": args.rays, "samples_per_ray": S,
                   "parallelism": "1 GPU" if world == 1 else f"ray-sharded dp{world}, one bucketed RCCL all-reduce of leaf grads "
                                                            f"(touched 64-voxel blocks only: {comm['bytes'] / 1e6:.1f} MB per step)"},
        "loss": float(loss.item()), "mode": mode, "loss_impl": "torch" if args.torch_loss else ("fused HIP (losses.rgbd_loss)" if args.separate_loss or stage == 'coarse'
                                                                  else "fused into the compositing launches (render_batch_ray_rgbd_loss)"),
        "eager_rays_per_s": world * args.rays * eager_steps / eager_elapsed,
    }
    if events:
        dur = np.array([a.elapsed_time(b) for a, b in events]) * 1e-3          # seconds
        avg = float(dur.mean())
        flops = n_points * 2 * FLOP_FWD_PER_POINT[stage]                        # dX + dW of every linear layer
        step_flops = args.rays * S * 3 * FLOP_FWD_PER_POINT[stage]
        traffic = None          # HBM-side bytes per launch from the committed PMC passes (profiles/r01_traffic.json)
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and args.rays == 1000 and stage == 'color':
            traffic = json.load(open(tpath)).get("decoder_bwd_split_kernel", {}).get("bytes_per_launch")
        out["roofline"] = {
            "kernel": "decoder_bwd_split_kernel", "bound": "mfma", "achieved": flops / avg / 1e12, "peak": PEAK_F32_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": flops / avg / PEAK_F32_MFMA, "traffic": traffic,
            "avg_launch_us": avg * 1e6, "launches": int(len(dur)),
            "peak_measured": 138.9,     # TFLOP/s, v_mfma_f32_16x16x4_f32 micro-benchmark on the box (profiles/r01_peaks.txt)
            # `achieved` prices the yardstick FLOPs of EVERY sample; tiles whose gradient is exactly zero (transmittance
            # underflowed behind the room's boundary) are not scheduled at all -- this is the share that was
            "active_tile_fraction": EF.last_active_tile_fraction(),
            "step_frac": step_flops / PEAK_F32_MFMA / (elapsed / args.steps),
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sc, rays_cpu, stage)
    if rank == 0 and not args.eager and mode == 'hipgraph' and gstep is not None and phases is not None:
        print("[bench] phases ms/step: depth-max all-reduce %.3f, graph replay %.3f, gradient all-reduce %.3f"
              % tuple(p / args.steps * 1e3 for p in phases), file=sys.stderr)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or force_comm:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
