"""Measurement config 3: the tracker's camera iteration with the event term on room0 at Replica
resolution -- 200-pixel RGB-D batch + `render_img_rescale` (0.15 x 680 x 1200 = 18 360 rays x 48, gradients to the
pose) -> PyTorch-ROCm UNet_2heads(6,2,2) (seeded random weights: the pretrained checkpoint is not in the image) ->
blurred-L2 event loss (kernel 9, balancer 0.025) -> backward to the 7 pose numbers -> Adam.
Reports the path alone (render + backward of the rescaled image) and path + U-Net + losses."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import evennicer_slam_amd as E
from evennicer_slam_amd.mapper import FusedAdam

dev = torch.device('cuda', 0)
DEV = 'cuda:0'
sc = bench.build_scene_cpu('room0', 0)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
for p in model.parameters(): p.requires_grad_(False)
grids = {k: v.to(dev) for k, v in sc['grids'].items()}
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
H, W, fx, fy, cx, cy = (bench.CAM[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
g = torch.Generator().manual_seed(1)
depth_img = (torch.rand(H, W, generator=g) * 3.0 + 0.5).to(dev)
color_img = torch.rand(H, W, 3, generator=g).to(dev)
pre_color = torch.rand(H, W, 3, generator=g).to(dev)
gt_event = torch.randint(0, 4, (H, W, 2), generator=g).float().to(dev)
gt_mask = (gt_event.sum(-1) > 2).long()
cfg = dict(sc['cfg'])
cfg['tracking'] = {'device': DEV, 'w_color_loss': 0.5, 'ignore_edge_W': 100, 'ignore_edge_H': 100, 'handle_dynamic': True,
                   'use_color_in_tracking': True}
cfg['event'] = {'activate_events': True, 'blur': True, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                'balancer': 0.025}
torch.manual_seed(0)
net = E.event.UNet_2heads(6, 2, 2)
for p in net.parameters(): p.requires_grad_(False)
net = net.to(dev).eval()
if os.environ.get('CHANNELS_LAST', '0') == '1':
    net = net.to(memory_format=torch.channels_last)
slam = types.SimpleNamespace(nice=True, bound=sc['bound'], renderer=renderer, event_net=net, low_gpu_mem=False, **bench.CAM)
trk = E.tracker.TrackerIteration(cfg, None, slam)
trk.c, trk.decoders = grids, model
ct = torch.tensor([1.0, 0.0, 0.0, 0.0, 3.0, 1.0, 0.0], device=dev, requires_grad=True)
opt = FusedAdam([ct], lr=1e-3)
SF = 0.15


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _i in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


def full(): return trk.optimize_cam_in_batch(ct, None, color_img, depth_img, gt_event, gt_mask, 200, opt, 0, 0, pre_color, rgbd=True, event=True, scale_factor=SF)
def rgbd_only(): return trk.optimize_cam_in_batch(ct, None, color_img, depth_img, None, None, 200, opt, 0, 0, None, rgbd=True, event=False)
def path_only():
    opt.zero_grad()
    col = trk._render_rescaled(ct, depth_img, SF)
    col.sum().backward()
    opt.step()
def unet_only():
    x = torch.rand(1, 6, int(H * SF), int(W * SF), device=dev, requires_grad=True)
    e, m = net(x)
    (e.sum() + m.sum()).backward()

n = int(os.environ.get('STEPS', 30))
for f in (full, rgbd_only, path_only, unet_only):
    for _i in range(3): f()
r = [x for x in full()[:3]]          # floats only: tensors of an eager iteration (autograd graph alive) must not survive into a capture
t_full, t_rgbd, t_path, t_unet = timed(full, n), timed(rgbd_only, n), timed(path_only, n), timed(unet_only, n)
rays = int(H * SF) * int(W * SF)
print(f"config 3, room0, eager: full iteration (200-ray RGB-D + {rays}-ray event render + U-Net + losses + Adam) {t_full * 1e3:.2f} ms; "
      f"RGB-D part alone {t_rgbd * 1e3:.2f} ms; event render path alone (fwd+bwd to the pose) {t_path * 1e3:.2f} ms "
      f"({rays / t_path / 1e6:.2f} M rays/s); U-Net fwd+bwd alone {t_unet * 1e3:.2f} ms; losses rgbd {r[0]:.2f} event {r[1]:.2f} mask {r[2]:.3f}")
git = None
if os.environ.get('GRAPH', '1') == '1':
    import gc
    gc.collect()
    git = E.tracker.GraphedCameraIteration(trk, ct, opt, color_img, depth_img, gt_event, gt_mask, pre_color, batch_size=200,
                                           rgbd=True, event=True, scale_factor=SF)
    git_ev = E.tracker.GraphedCameraIteration(trk, ct, opt, color_img, depth_img, gt_event, gt_mask, pre_color, batch_size=200,
                                              rgbd=False, event=True, scale_factor=SF)
if git is not None:
    for _i in range(3): git.step(); git_ev.step()
    tg, tge = timed(git.step, n * 3), timed(git_ev.step, n * 3)
    lr_, le_, lm_ = git.step()
    tf = timed(lambda: git.set_frame(color_img, depth_img, gt_event, gt_mask, pre_color), 20)
    print(f"config 3, one hipGraph per iteration (static-shape formulation): RGB-D + event {tg * 1e3:.2f} ms, event only "
          f"(frames without RGB-D) {tge * 1e3:.2f} ms, per-frame set_frame {tf * 1e3:.2f} ms; losses rgbd {lr_.item():.2f} event {le_.item():.2f}")
