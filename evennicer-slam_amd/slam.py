"""Single-process run harness: the reference's `run.py` -> `EvenNICER_SLAM.run` starts a tracker, a
mapper and a coarse-mapper PROCESS that hand the map over through shared tensors (src/EvenNICER_SLAM.py:278-332).  Here
the same per-frame schedule runs in one process on one GPU, tracker and mapper alternating on the HIP path:

  frame 0         pose = ground truth (Tracker.py:283-285); `mapping.iters_first` joint iterations on the first frame
  frame i > 0     tracker: constant-speed or last-pose initialisation (Tracker.py:295-303), `tracking.iters` camera
                  iterations (`TrackerIteration.optimize_cam_in_batch`, RGB-D term; the event term when an event network
                  is given), the candidate with the smallest loss is kept (:321-330)
  every `mapping.every_frame`-th frame: `mapping.iters` joint iterations (`mapper.MapperIteration`: frustum-masked grids,
                  colour decoder, local BA over the window of keyframes -- the reference's `global` selection: random
                  keyframes + the latest one + the current frame, Mapper.py:280-303) and `update_para_from_mapping`
                  then -- `cfg['coarse']` -- the COARSE mapper's round on the same frame (the reference's third process,
                  EvenNICER_SLAM.py:303-311: `Mapper(..., coarse_mapper=True)` runs the same schedule with stage `coarse`
                  only, every voxel of `grid_coarse` optimisable, no depth guidance, no BA; Mapper.py:133-136,326-328,
                  460-461,550,797-798)
  keyframes       every `mapping.keyframe_every` frames (Mapper.py:687-692)
  end             `Logger.log` checkpoint in the reference's format, ATE through `eval_ate.evaluate_checkpoint`

`SLAM.prefit_decoders` stands in for the pretrained ConvONet decoders the reference loads (EvenNICER_SLAM.load_pretrain;
the checkpoints are not in the image): decoders and grids are fitted jointly to a few ground-truth-posed frames of the
sequence, the decoders are kept, the grids are re-initialised -- the run itself then optimises grids (and the colour decoder)
only, as the reference does.

Not here: mesh extraction, visualiser, wandb."""
import os
import time
import types

import numpy as np
import torch

from . import functional as EF
from .common import get_camera_from_tensor, get_tensor_from_camera
from .decoder import get_model
from .eval_ate import Logger, evaluate_checkpoint
from .mapper import FusedAdam, MapperIteration
from .renderer import Renderer
from .scene import GRID_KEYS, grid_init, scene_bound
from .tracker import TrackerIteration

MAP_KEYS = ('grid_middle', 'grid_fine', 'grid_color')


def frustum_mask(c2w, depth, shape, bound, cam):
    """Frustum feature selection (Mapper.get_mask_from_c2w, src/Mapper.py:114-186) in torch: voxels of a [D,H,W] grid
    that project into the image in front of the measured depth (+0.5 m), plus those within 0.5 m of the camera.
    Returns bool [D,H,W].  (The reference samples the depth with cv2.remap INTER_LINEAR and a zero border; the same
    bilinear lookup is F.grid_sample(align_corners=True, padding zeros).)"""
    H, W, fx, fy, cx, cy = (cam[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
    dev = depth.device
    D_, H_, W_ = shape
    X, Y, Z = torch.meshgrid(torch.linspace(float(bound[0][0]), float(bound[0][1]), W_),
                             torch.linspace(float(bound[1][0]), float(bound[1][1]), H_),
                             torch.linspace(float(bound[2][0]), float(bound[2][1]), D_), indexing='ij')
    pts = torch.stack([X, Y, Z], -1).reshape(-1, 3).to(dev)
    c2w4 = torch.eye(4, device=dev, dtype=torch.float64)
    c2w4[:3] = c2w[:3].double().to(dev)
    w2c = torch.linalg.inv(c2w4)
    cam_pts = (w2c[:3, :3] @ pts.double().T + w2c[:3, 3:4]).T
    cam_pts[:, 0] *= -1
    K = torch.tensor([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]], dtype=torch.float64, device=dev)
    uv = (K @ cam_pts.T).T
    z = uv[:, 2:3] + 1e-5
    uv = (uv[:, :2] / z).float()
    gx = 2 * uv[:, 0] / (W - 1) - 1
    gy = 2 * uv[:, 1] / (H - 1) - 1
    depths = torch.nn.functional.grid_sample(depth[None, None].float(), torch.stack([gx, gy], -1)[None, :, None, :],
                                             mode='bilinear', padding_mode='zeros', align_corners=True).reshape(-1, 1)
    mask = (uv[:, 0] < W) & (uv[:, 0] > 0) & (uv[:, 1] < H) & (uv[:, 1] > 0)
    depths = torch.where(depths == 0, depths.max(), depths)
    mask = mask & (0 <= -z[:, 0]) & (-z[:, 0] <= depths[:, 0].double() + 0.5)
    near = ((pts - c2w[:3, 3].to(dev, pts.dtype)[None]) ** 2).sum(1) < 0.25
    return (mask | near).reshape(W_, H_, D_).permute(2, 1, 0).contiguous()


class SLAM:
    def __init__(self, cfg, dataset, output, device='cuda:0', event_net=None, verbose=False, static_shapes=False):
        self.cfg, self.dataset, self.output, self.device, self.verbose = cfg, dataset, output, device, verbose
        self.static_shapes = static_shapes
        self.scale = cfg['scale']
        self.H, self.W = cfg['cam']['H'] - 2 * cfg['cam']['crop_edge'], cfg['cam']['W'] - 2 * cfg['cam']['crop_edge']
        self.fx, self.fy = cfg['cam']['fx'], cfg['cam']['fy']
        self.cx, self.cy = cfg['cam']['cx'] - cfg['cam']['crop_edge'], cfg['cam']['cy'] - cfg['cam']['crop_edge']
        self.cam = dict(H=self.H, W=self.W, fx=self.fx, fy=self.fy, cx=self.cx, cy=self.cy)
        self.bound = scene_bound(cfg['mapping']['bound'], self.scale, cfg['grid_len']['bound_divisible'])   # load_bound
        self.shared_decoders = get_model(cfg).to(device)
        self.shared_decoders.bound = self.bound
        for name in ('middle_decoder', 'fine_decoder', 'color_decoder'):
            getattr(self.shared_decoders, name).bound = self.bound
        self.shared_decoders.coarse_decoder.bound = self.bound * cfg['model']['coarse_bound_enlarge']
        # channels_last_3d: the reference's grid tensors stored voxel-major (read in place by the kernels; cfg['grid_memory_format']
        # = 'contiguous' keeps the reference's strides)
        mf = None if (cfg.get('grid_memory_format', 'channels_last_3d') == 'contiguous' or str(device) == 'cpu') else torch.channels_last_3d
        self.shared_c = grid_init(self.bound, cfg['grid_len'], cfg['model']['c_dim'], cfg['model']['coarse_bound_enlarge'], device,
                                  memory_format=mf)
        n = len(dataset)
        self.n_img = n
        self.estimate_c2w_list = torch.zeros((n, 4, 4))
        self.gt_c2w_list = torch.zeros((n, 4, 4))
        self.ckptsdir = os.path.join(output, 'ckpts')
        os.makedirs(self.ckptsdir, exist_ok=True)
        self.nice = True
        self.renderer = Renderer(cfg, None, self)
        self.event_net = event_net
        self.low_gpu_mem = False
        self.logger = Logger(cfg, None, self)
        self.tracker = TrackerIteration(cfg, None, self)
        self.keyframe_dict, self.keyframe_list = [], []
        self.timing = dict(track=0.0, map=0.0)

    # ---------------------------------------------------------------- tracker side
    def update_para_from_mapping(self):
        """Tracker.update_para_from_mapping (Tracker.py:247-259): the tracker's own copies of the map."""
        import copy
        self.tracker.decoders = copy.deepcopy(self.shared_decoders)
        for p in self.tracker.decoders.parameters():
            p.requires_grad_(False)
        self.tracker.c = {k: v.detach().clone() for k, v in self.shared_c.items()}

    def track(self, idx, gt_color, gt_depth, gt_event, gt_mask, pre_gt_color):
        t = getattr(self, '_track_cfg', None) or self.cfg['tracking']
        pre = self.estimate_c2w_list[idx - 1].to(self.device)
        if t.get('const_speed_assumption', True) and idx - 2 >= 0:         # Tracker.py:295-301
            pre = pre.float()
            delta = pre @ self.estimate_c2w_list[idx - 2].to(self.device).float().inverse()
            est = delta @ pre
        else:
            est = pre
        use_event = self.event_net is not None and self.cfg['event'].get('activate_events', False)
        if t.get('graphed', False) and int(t['iters']) > 0:
            return self._track_graphed(idx, est, gt_color, gt_depth, gt_event, gt_mask, pre_gt_color, t, use_event)
        camera_tensor = get_tensor_from_camera(est.detach()).to(self.device).requires_grad_(True)
        opt = FusedAdam([camera_tensor], lr=t['lr'])
        best, best_loss, first_loss = camera_tensor.detach().clone(), None, None
        for it in range(t['iters']):
            out = self.tracker.optimize_cam_in_batch(camera_tensor, est, gt_color, gt_depth, gt_event, gt_mask, t['pixels'], opt,
                                                     idx, it, pre_gt_color, rgbd=True, event=use_event,
                                                     scale_factor=self.cfg['event'].get('scale_factor', 0.1))
            loss = out[0] + (out[1] if (use_event and out[1] is not None) else 0.0)
            if it == 0:
                first_loss = loss
            if best_loss is None or loss < best_loss:                       # Tracker.py:321-330 (candidate with the least loss)
                best_loss, best = loss, camera_tensor.detach().clone()
        c2w = torch.eye(4, device=self.device)
        c2w[:3] = get_camera_from_tensor(best)
        if self.verbose:
            gt = self.gt_c2w_list[idx][:3, 3].to(self.device)
            print(f'frame {idx}: tracking loss {first_loss} -> {best_loss}; translation error init '
                  f'{float((est[:3, 3] - gt).norm()):.4f} -> {float((c2w[:3, 3] - gt).norm()):.4f} m', flush=True)
        return c2w

    def _track_graphed(self, idx, est, gt_color, gt_depth, gt_event, gt_mask, pre_gt_color, t, use_event):
        """`track` with the camera iterations as replays of ONE hipGraph (tracker.GraphedCameraIteration, captured at the first tracked
        frame and kept): per frame the initial pose goes into the captured camera tensor, the optimiser is reset, the images are
        copied in (set_frame draws the frame's pixels ahead) and the map the mapper has just published is copied into the captured
        tensors (refresh_map); the losses stay on the device and the least-loss candidate (Tracker.py:321-330) is picked by one
        argmin behind the last iteration -- no host round trip per iteration."""
        from .tracker import GraphedCameraIteration
        iters, n = int(t['iters']), int(t['pixels'])
        sf = self.cfg['event'].get('scale_factor', 0.1)
        g = getattr(self, '_gtrack', None)
        init = get_tensor_from_camera(est.detach()).to(self.device).float()
        if g is None or g['key'] != (iters, n, use_event):
            ct = init.clone().requires_grad_(True)
            opt = FusedAdam([ct], lr=t['lr'])
            git = GraphedCameraIteration(self.tracker, ct, opt, gt_color, gt_depth, gt_event if use_event else None,
                                         gt_mask if use_event else None, pre_gt_color if use_event else None, batch_size=n, rgbd=True,
                                         event=use_event, scale_factor=sf, n_draws=iters)
            g = self._gtrack = dict(key=(iters, n, use_event), ct=ct, opt=opt, git=git,
                                    cams=torch.empty((iters, 7), dtype=torch.float32, device=self.device),
                                    losses=torch.empty(iters, dtype=torch.float64, device=self.device))
        ct, opt, git = g['ct'], g['opt'], g['git']
        with torch.no_grad():
            ct.copy_(init.reshape(ct.shape))
        opt.reset()
        opt.set_lr(t['lr'])
        git.refresh_map()                                   # (update_para_from_mapping replaced the tracker's map objects)
        git.set_frame(gt_color, gt_depth, gt_event if use_event else None, gt_mask if use_event else None,
                      pre_gt_color if use_event else None)
        for it in range(iters):
            l_rgbd, l_event, _l_mask = git.step()
            g['losses'][it].copy_(l_rgbd + l_event if use_event else l_rgbd)
            g['cams'][it].copy_(ct.detach().reshape(-1))         # the candidate of this loss: the pose AFTER the step (Tracker.py:321-330)
        k = torch.argmin(g['losses'])
        best = g['cams'][k]
        c2w = torch.eye(4, device=self.device)
        c2w[:3] = get_camera_from_tensor(best)
        if self.verbose:
            gt = self.gt_c2w_list[idx][:3, 3].to(self.device)
            print(f'frame {idx}: tracking loss {float(g["losses"][0])} -> {float(g["losses"][k])} (graphed); translation error init '
                  f'{float((est[:3, 3] - gt).norm()):.4f} -> {float((c2w[:3, 3] - gt).norm()):.4f} m', flush=True)
        return c2w

    # ---------------------------------------------------------------- mapper side
    def map(self, idx, gt_color, gt_depth, cur_c2w, iters):
        m = self.cfg['mapping']
        window = m.get('mapping_window_size', 5)
        frames = []
        if self.keyframe_dict:                                               # 'global' selection, Mapper.py:280-303
            n_old = len(self.keyframe_dict) - 1
            pick = list(np.random.permutation(n_old)[:max(window - 2, 0)]) if n_old > 0 else []
            pick = sorted(set(int(p) for p in pick) | {len(self.keyframe_dict) - 1})
            oldest = min(pick)
            for k in pick:
                kf = self.keyframe_dict[k]
                frames.append(dict(depth=kf['depth'], color=kf['color'], c2w=kf['est_c2w'], fixed=(k == oldest), key=k))
        frames.append(dict(depth=gt_depth, color=gt_color.float(), c2w=cur_c2w, fixed=False, key=-1))
        masks = None
        if m.get('frustum_feature_selection', True):                         # Mapper.py:330-361
            masks = {k: frustum_mask(cur_c2w, gt_depth, tuple(self.shared_c[k].shape[2:]), self.bound, self.cam) for k in MAP_KEYS}
        ba = bool(m.get('BA', False)) and len(self.keyframe_dict) > 4 and idx > 0      # Mapper.py:700: BA once 4 keyframes exist
        cfg = dict(self.cfg)
        # the first frame is mapped with lr_first_factor (Mapper.py:794-796: lr_factor = lr_first_factor with iters_first)
        cfg['mapping'] = dict(m, BA=ba, lr_factor=m.get('lr_first_factor', m['lr_factor']) if idx == 0 else m['lr_factor'])
        it = MapperIteration(cfg, self.renderer, self.shared_c, self.shared_decoders, frames, self.cam, masks=masks, keys=MAP_KEYS,
                             static_shapes=self.static_shapes)
        loss = None
        for j in range(iters):
            loss = it.step(j, iters)
        cams = it.finish()
        if self.cfg.get('coarse', False):
            # the coarse mapper's round on this frame (the reference's third process): stage `coarse`, every voxel of grid_coarse,
            # the same selected frames at their current (fixed) poses
            cfg_c = dict(self.cfg)
            cfg_c['mapping'] = dict(cfg['mapping'], BA=False)
            frames_c = [dict(f, fixed=True) for f in frames]
            itc = MapperIteration(cfg_c, self.renderer, self.shared_c, self.shared_decoders, frames_c, self.cam, masks=None,
                                  keys=('grid_coarse',), static_shapes=self.static_shapes)
            for j in range(iters):
                self.last_coarse_loss = itc.step(j, iters)
            itc.finish()
        if ba:                                                               # Mapper.py:644-660: poses back to the lists
            for f, ct in zip(frames, cams):
                if ct is None:
                    continue
                c2w = torch.eye(4, device=self.device)
                c2w[:3] = get_camera_from_tensor(ct)
                if f['key'] == -1:
                    cur_c2w = c2w
                else:
                    self.keyframe_dict[f['key']]['est_c2w'] = c2w
        return cur_c2w, (float(loss.item()) if loss is not None else None)

    # ---------------------------------------------------------------- the run
    def prefit_decoders(self, indices, iters=300, pixels=None, lr_grid=0.02, lr_dec=2e-3, seed=0):
        """Stand-in for `load_pretrain` (no checkpoint ships with the image): fit ALL decoders and the grids jointly to the
        frames `indices` of the dataset at their ground-truth poses (random pixels -> render -> the mapper's loss -> Adam,
        colour stage, through the HIP path), keep the decoders, re-initialise the grids.  Returns the final loss."""
        from .common import get_samples
        from .losses import rgbd_loss
        dev = self.device
        pixels = pixels or self.cfg['mapping']['pixels']
        items = [self.dataset[i] for i in indices]
        frames = [(it[1].float(), it[2], it[-1][:3].to(dev)) for it in items]
        grids = {k: v.detach().clone().requires_grad_(True) for k, v in self.shared_c.items()}
        dec = self.shared_decoders
        params = [q for q in dec.parameters()]
        for q in params:
            q.requires_grad_(True)
        opt = torch.optim.Adam([{'params': [grids[k] for k in MAP_KEYS], 'lr': lr_grid}, {'params': params, 'lr': lr_dec}])
        gen = torch.cuda.get_rng_state(dev)
        torch.manual_seed(seed)
        n = max(pixels // len(frames), 1)
        loss = None
        for _ in range(iters):
            ro, rd, gd, gc = [], [], [], []
            for color, depth, c2w in frames:
                o, d, dep, col = get_samples(0, self.H, 0, self.W, n, self.H, self.W, self.fx, self.fy, self.cx, self.cy, c2w,
                                             depth, color, dev)
                ro.append(o.float()); rd.append(d.float()); gd.append(dep.float()); gc.append(col.float())
            ro, rd, gd, gc = torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)
            opt.zero_grad(set_to_none=True)
            depth, _u, color = self.renderer.render_batch_ray(grids, dec, rd, ro, dev, 'color', gt_depth=gd)
            loss = rgbd_loss(depth, color, gd, gc, self.cfg['mapping']['w_color_loss'])
            with EF.engine_on_calling_thread():
                loss.backward()
            opt.step()
        for q in params:
            q.grad = None
        torch.cuda.set_rng_state(gen, dev)
        return float(loss.item()) if loss is not None else None

    def run(self, max_frames=None, tracking_iters=None):
        """tracking_iters overrides cfg['tracking']['iters'] (0: poses stay at their constant-speed initialisation -- the
        baseline an ATE improvement is measured against)."""
        m, t = self.cfg['mapping'], dict(self.cfg['tracking'])
        if tracking_iters is not None:
            t['iters'] = int(tracking_iters)
        self._track_cfg = t
        n = self.n_img if max_frames is None else min(max_frames, self.n_img)
        pre_color = None
        for idx in range(n):
            item = self.dataset[idx]
            if len(item) == 6:
                _, gt_color, gt_depth, gt_event, gt_mask, gt_c2w = item
            else:
                (_, gt_color, gt_depth, gt_c2w), gt_event, gt_mask = item, None, None
            self.gt_c2w_list[idx] = gt_c2w.clone().cpu()
            t0 = time.perf_counter()
            if idx == 0 or t.get('gt_camera', False):
                c2w = gt_c2w.clone()
            else:
                c2w = self.track(idx, gt_color.float(), gt_depth, gt_event, gt_mask, pre_color)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            self.timing['track'] += t1 - t0
            if idx == 0 or idx % m['every_frame'] == 0 or idx == n - 1:
                iters = m['iters_first'] if idx == 0 else m['iters']
                c2w, loss = self.map(idx, gt_color, gt_depth, c2w, iters)
                self.update_para_from_mapping()
                if self.verbose:
                    print(f'frame {idx}: mapping loss {loss}')
                if idx % m['keyframe_every'] == 0 or idx == n - 2:           # Mapper.py:687-692
                    self.keyframe_list.append(idx)
                    self.keyframe_dict.append(dict(gt_c2w=gt_c2w.cpu(), idx=idx, color=gt_color.float(), depth=gt_depth,
                                                   est_c2w=c2w.clone()))
            torch.cuda.synchronize()
            self.timing['map'] += time.perf_counter() - t1
            self.estimate_c2w_list[idx] = c2w.detach().cpu()
            pre_color = gt_color.float()
        ckpt = self.logger.log(n - 1, self.keyframe_dict, self.keyframe_list, selected_keyframes=None)
        return dict(ckpt=ckpt, frames=n, fps=n / max(self.timing['track'] + self.timing['map'], 1e-9), timing=dict(self.timing))

    def evaluate(self, ckpt):
        return evaluate_checkpoint(ckpt, scale=self.scale)
