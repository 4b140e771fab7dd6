"""The tracker's RGB-D camera iteration as the harness runs it (tracker.GraphedCameraIteration: TrackerIteration._rgbd_loss with
the in-bound prefilter of Tracker.py:164-174 and, with DYN=1, the median mask of :180-182 -- the reference's default
`handle_dynamic: True`) on room0, one hipGraph per iteration.  RAYS (200), DYN (1 / 0), STEPS (300)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import evennicer_slam_amd as E
from evennicer_slam_amd.mapper import FusedAdam

dev = torch.device('cuda', 0)
N = int(os.environ.get('RAYS', 200))
sc = bench.build_scene_cpu('room0', 0)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
for p in model.parameters(): p.requires_grad_(False)
LAYOUT = os.environ.get('LAYOUT', 'channels_last_3d')
mf = torch.channels_last_3d if LAYOUT == 'channels_last_3d' else torch.contiguous_format
grids = {k: v.to(dev).contiguous(memory_format=mf) for k, v in sc['grids'].items()}
slam = types.SimpleNamespace(nice=True, bound=sc['bound'], event_net=None, low_gpu_mem=False, **bench.CAM)
slam.renderer = E.Renderer(sc['cfg'], None, slam)
H, W = bench.CAM['H'], bench.CAM['W']
g = torch.Generator().manual_seed(1)
depth_img = (torch.rand(H, W, generator=g) * 3.0 + 0.5).to(dev)
color_img = torch.rand(H, W, 3, generator=g).to(dev)
for dyn in ([1, 0] if 'DYN' not in os.environ else [int(os.environ['DYN'])]):
    cfg = dict(sc['cfg'])
    cfg['tracking'] = {'device': dev, 'w_color_loss': 0.5, 'ignore_edge_W': 100, 'ignore_edge_H': 100, 'handle_dynamic': bool(dyn),
                       'use_color_in_tracking': True, 'lr': 1e-3, 'pixels': N, 'iters': 10}
    cfg['event'] = {'activate_events': False, 'blur': False, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0, 'balancer': 0.025}
    trk = E.tracker.TrackerIteration(cfg, None, slam)
    trk.c, trk.decoders = grids, model
    ct = torch.tensor([1.0, 0.0, 0.0, 0.0, 3.0, 1.0, 0.0], device=dev, requires_grad=True)
    opt = FusedAdam([ct], lr=1e-3)
    gi = E.tracker.GraphedCameraIteration(trk, ct, opt, color_img, depth_img, batch_size=N, rgbd=True, event=False)
    for _ in range(5):
        gi.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(os.environ.get('STEPS', 300))
    for _ in range(n):
        out = gi.step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print(f"tracker RGB-D iteration as the harness runs it, room0, {N} rays x 48, handle_dynamic={bool(dyn)}, in-bound prefilter on, {LAYOUT} grids: "
          f"hipGraph {t * 1e6:.1f} us/iter, loss {float(out[0].item()):.3f}", flush=True)
    del gi
