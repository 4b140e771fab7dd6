"""Secondary metric (BASELINE.json: "Replica room0 tracking+mapping FPS") on SYNTHETIC data with random-weight
networks -- the datasets and pretrained weights are not in the image, so this measures the schedule's compute, not
accuracy: room0 grids, Replica camera, the shipped schedule (configs/Replica/replica.yaml:24-27,34-36 +
configs/nice_slam.yaml): per frame 10 camera iterations (event term every frame, RGB-D term when a depth frame is
available: every 5th frame, `event.rgbd_every_frame`), every 5th frame 60 mapper iterations of 1000 rays (all in the
colour stage, the most expensive one; the 1500 iterations of frame 0 are reported separately).

One process, one GPU: the tracker and mapper of the reference run as separate processes; here they alternate.
Tracker: tracker.GraphedCameraIteration (one hipGraph per iteration, map followed through refresh_map after each
mapping round like Tracker.update_para_from_mapping).  Mapper: MaskedGridOptimizer + FusedAdam, one hipGraph per
iteration, write_back after the round."""
import copy, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import evennicer_slam_amd as E
from evennicer_slam_amd.mapper import MaskedGridOptimizer, FusedAdam
from evennicer_slam_amd.graph import GraphedStep

dev = torch.device('cuda', 0)
DEV = 'cuda:0'
FRAMES = int(os.environ.get('FRAMES', 20))
KEYS = ('grid_middle', 'grid_fine', 'grid_color')
sc = bench.build_scene_cpu('room0', 0)
H, W, fx, fy, cx, cy = (bench.CAM[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
g = torch.Generator().manual_seed(1)
frames = []
for _ in range(4):                                          # a few distinct synthetic frames, cycled
    d = torch.rand(H, W, generator=g) * 3.0 + 0.5
    ev = torch.randint(0, 4, (H, W, 2), generator=g).float()
    frames.append(dict(depth=d.to(dev), color=torch.rand(H, W, 3, generator=g).to(dev), event=ev.to(dev),
                       mask=(ev.sum(-1) > 2).long().to(dev)))

# ---- mapper side (owns the map)
model = copy.deepcopy(sc['model']).to(dev); bench.attach_bounds(model, sc['bound'])
for name in ('coarse_decoder', 'middle_decoder', 'fine_decoder'):          # fix_fine: True; middle / coarse never optimised
    for p in getattr(model, name).parameters():
        p.requires_grad_(False)
grids = {k: v.to(dev).contiguous(memory_format=torch.channels_last_3d if os.environ.get('LAYOUT', 'channels_last_3d') == 'channels_last_3d' else torch.contiguous_format) for k, v in sc['grids'].items()}
masks = {}
for k in KEYS:
    D, Hh, Ww = grids[k].shape[2:]
    m = torch.zeros(D, Hh, Ww, dtype=torch.bool)
    m[:, :, Ww // 4: 3 * Ww // 4] = True
    masks[k] = m.to(dev)
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
opt = MaskedGridOptimizer(grids, masks, keys=KEYS)
dec_opt = FusedAdam(list(model.color_decoder.parameters()), lr=0.005)
opt.set_lr({k: 0.005 for k in KEYS})
c_map = opt.render_grids()
m_ro, m_rd, m_gd, m_gc = [t.to(dev) for t in bench.make_rays(sc, 1000, 1000)]
one = {}


def map_it():
    dec_opt.zero_grad()
    loss, depth, var, color = renderer.render_batch_ray_rgbd_loss(c_map, model, m_rd, m_ro, dev, 'color', m_gd, m_gc, 0.2)
    if 'one' not in one: one['one'] = torch.ones_like(loss)
    loss.backward(gradient=one['one'])
    dec_opt.step()
    opt.step()
    return loss


for _ in range(3): map_it()
import gc; dec_opt.zero_grad(); gc.collect()
g_map = GraphedStep(map_it)

# ---- tracker side (its own copies of the map, Tracker.py:247-260)
t_model = copy.deepcopy(model)
for p in t_model.parameters(): p.requires_grad_(False)
bench.attach_bounds(t_model, sc['bound'])
t_grids = {k: v.detach().clone() for k, v in grids.items()}
cfg = dict(sc['cfg'])
cfg['tracking'] = {'device': DEV, 'w_color_loss': 0.5, 'ignore_edge_W': 100, 'ignore_edge_H': 100, 'handle_dynamic': True,
                   'use_color_in_tracking': True}
cfg['event'] = {'activate_events': True, 'blur': True, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                'balancer': 0.025}
torch.manual_seed(0)
net = E.event.UNet_2heads(6, 2, 2)
for p in net.parameters(): p.requires_grad_(False)
net = net.to(dev).eval()
t_renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
slam = types.SimpleNamespace(nice=True, bound=sc['bound'], renderer=t_renderer, event_net=net, low_gpu_mem=False, **bench.CAM)
trk = E.tracker.TrackerIteration(cfg, None, slam)
trk.c, trk.decoders = t_grids, t_model
ct = torch.tensor([1.0, 0.0, 0.0, 0.0, 3.0, 1.0, 0.0], device=dev, requires_grad=True)
cam_opt = FusedAdam([ct], lr=1e-3)
f0 = frames[0]
kw = dict(batch_size=200, scale_factor=0.15)
git_full = E.tracker.GraphedCameraIteration(trk, ct, cam_opt, f0['color'], f0['depth'], f0['event'], f0['mask'], f0['color'],
                                            rgbd=True, event=True, **kw)
git_ev = E.tracker.GraphedCameraIteration(trk, ct, cam_opt, f0['color'], f0['depth'], f0['event'], f0['mask'], f0['color'],
                                          rgbd=False, event=True, **kw)


def update_para_from_mapping():                              # Tracker.py:247-260, in place
    opt.write_back()
    with torch.no_grad():
        for k in KEYS:
            t_grids[k].copy_(grids[k])
        for pt, pm in zip(t_model.parameters(), model.parameters()):
            pt.copy_(pm)
    git_full.refresh_map()


def mapping_round(n):
    for _ in range(n): g_map.replay()


def sync(): torch.cuda.synchronize()


# first frame: 1500 mapper iterations (iters_first)
sync(); t0 = time.perf_counter(); mapping_round(1500); update_para_from_mapping(); sync()
t_first = time.perf_counter() - t0
t_track = t_map = 0.0
sync(); t_all = time.perf_counter()
for i in range(1, FRAMES + 1):
    f, prev = frames[i % 4], frames[(i - 1) % 4]
    rgbd = i % 5 == 0
    a = time.perf_counter()
    git = git_full if rgbd else git_ev
    git.set_frame(f['color'], f['depth'], f['event'], f['mask'], prev['color'])
    for _ in range(10): git.step()
    sync(); b = time.perf_counter(); t_track += b - a
    if i % 5 == 0:                                          # mapping.every_frame 5
        mapping_round(60); update_para_from_mapping()
        sync(); t_map += time.perf_counter() - b
sync(); t_all = time.perf_counter() - t_all
print(f"synthetic Replica-schedule run, room0, 1 GPU: {FRAMES} frames in {t_all * 1e3:.1f} ms = {FRAMES / t_all:.1f} frames/s "
      f"(tracking {t_track / FRAMES * 1e3:.1f} ms/frame = 10 iterations; mapping {t_map / max(FRAMES // 5, 1) * 1e3:.1f} ms per round "
      f"of 60 iterations incl. write-back and the tracker's map update); first-frame mapping (1500 iterations) {t_first * 1e3:.0f} ms")
