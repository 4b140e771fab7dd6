"""Analytic, view-consistent RGB-D scenes (no dataset ships with the image): an axis-aligned room seen from inside with one
box standing in it.  Depth and colour of every pixel follow from ray / box intersections, so the frames of a camera
trajectory are renderings of ONE fixed geometry -- what the run harness needs to show that tracking converges
and what `bench.py` uses for its second scene variant (a map with real surfaces instead of random-init occupancy).

Conventions are the reference's: pixel (i, j) -> direction [(i - cx) / fx, -(j - cy) / fy, -1] rotated by c2w[:3, :3]
(src/common.py:74-89); `depth` is the ray parameter of that un-normalised direction, i.e. the z-buffer depth the datasets
store (src/utils/datasets.py:107-113)."""
import math

import numpy as np
import torch


class BoxRoom:
    """Room = inside of the box [room_lo, room_hi]; one solid box [box_lo, box_hi] in it.  Colour is a smooth function of the
    3-D hit point (three sinusoids per channel), so it is consistent across views and learnable by the colour grid."""

    def __init__(self, room_lo, room_hi, box_lo=None, box_hi=None, seed=0):
        self.room_lo = torch.tensor(room_lo, dtype=torch.float64)
        self.room_hi = torch.tensor(room_hi, dtype=torch.float64)
        self.box_lo = torch.tensor(box_lo, dtype=torch.float64) if box_lo is not None else None
        self.box_hi = torch.tensor(box_hi, dtype=torch.float64) if box_hi is not None else None
        g = torch.Generator().manual_seed(seed)
        self.freq = (torch.rand(3, 3, generator=g, dtype=torch.float64) * 1.6 + 0.8)       # rad / m, per channel and axis
        self.phase = torch.rand(3, generator=g, dtype=torch.float64) * 2 * math.pi

    @staticmethod
    def for_bound(bound, margin=0.7, seed=0):
        """A room `margin` metres inside a scene bound [3,2] with a box on its floor (z is up in none of the reference's
        scenes in particular; the box simply sits against the low-y wall)."""
        b = torch.as_tensor(bound, dtype=torch.float64)
        lo, hi = b[:, 0] + margin, b[:, 1] - margin
        ext = hi - lo
        box_lo = lo + ext * torch.tensor([0.55, 0.0, 0.25], dtype=torch.float64)
        box_hi = lo + ext * torch.tensor([0.75, 0.35, 0.55], dtype=torch.float64)
        return BoxRoom(lo.tolist(), hi.tolist(), box_lo.tolist(), box_hi.tolist(), seed=seed)

    def intersect(self, rays_o, rays_d):
        """Ray parameter t >= 0 of the first surface hit (float64 [N]) for origins inside the room."""
        o, d = rays_o.double(), rays_d.double()
        dev = o.device
        d = torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
        lo, hi = self.room_lo.to(dev), self.room_hi.to(dev)
        t_wall = torch.maximum((lo - o) / d, (hi - o) / d).min(dim=-1).values
        if self.box_lo is None:
            return t_wall
        blo, bhi = self.box_lo.to(dev), self.box_hi.to(dev)
        ta, tb = (blo - o) / d, (bhi - o) / d
        t1 = torch.minimum(ta, tb).max(dim=-1).values
        t2 = torch.maximum(ta, tb).min(dim=-1).values
        hit = (t1 < t2) & (t1 > 0)
        return torch.where(hit & (t1 < t_wall), t1, t_wall)

    def color_at(self, pts):
        p = pts.double()
        dev = p.device
        return 0.5 + 0.45 * torch.sin(p @ self.freq.to(dev).T + self.phase.to(dev))

    def render(self, c2w, cam, device='cpu'):
        """(color float64 [H,W,3] in [0.05, 0.95], depth float32 [H,W]) of the view c2w ([3|4,4])."""
        H, W, fx, fy, cx, cy = (cam[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
        c2w = torch.as_tensor(c2w, dtype=torch.float64, device=device)
        j, i = torch.meshgrid(torch.arange(H, dtype=torch.float64, device=device),
                              torch.arange(W, dtype=torch.float64, device=device), indexing='ij')
        dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1).reshape(-1, 3)
        rays_d = dirs @ c2w[:3, :3].T
        rays_o = c2w[:3, 3].expand_as(rays_d)
        t = self.intersect(rays_o, rays_d)
        pts = rays_o + rays_d * t[:, None]
        return self.color_at(pts).reshape(H, W, 3), t.float().reshape(H, W)


def look_at(eye, target, up=(0., 0., 1.)):
    """c2w float64 [4,4] of a camera at `eye` looking at `target` in the reference's axes (camera looks along -z, y up)."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, u, -f, eye
    return torch.from_numpy(c2w)


def trajectory(room, n_frames, step=0.02, yaw_deg=0.4):
    """n_frames camera-to-world poses (float32 [4,4]) on a gentle arc inside the room: `step` metres and `yaw_deg` degrees of
    change in the viewing direction per frame, looking towards the box side of the room."""
    lo, hi = room.room_lo.numpy(), room.room_hi.numpy()
    c = 0.5 * (lo + hi)
    ext = hi - lo
    start = c - ext * np.array([0.25, 0.10, 0.05])
    poses = []
    for k in range(n_frames):
        a = math.radians(yaw_deg) * k
        eye = start + step * k * np.array([math.cos(0.3), math.sin(0.3), 0.15])
        tgt = c + ext * np.array([0.20 * math.cos(a) + 0.10, -0.25 + 0.20 * math.sin(a), 0.02 * math.sin(2 * a)])
        poses.append(look_at(eye, tgt).float())
    return poses


def demo_config(inp, evf, cam, device='cuda:0', env=None):
    """Configuration of the run harness for the tiny analytic room of `write_demo_sequence` (the reference's schedule and
    learning rates, configs/nice_slam.yaml, at a size that runs in seconds; environment-style overrides through `env`).
    One deliberate difference: lr_first_factor is 1, not 5 -- the decoders here are fitted to a handful of frames
    (SLAM.prefit_decoders), not ConvONet-pretrained, and a middle-grid learning rate of 0.5 on frame 0 left the map at its
    initial loss in about half of the runs (measured: tools/run_synthetic_slam.py, LR_FIRST=5)."""
    env = env or {}
    stage = {'coarse': dict(decoders_lr=0.0, coarse_lr=0.001, middle_lr=0.0, fine_lr=0.0, color_lr=0.0),
             'middle': dict(decoders_lr=0.0, coarse_lr=0.0, middle_lr=0.1, fine_lr=0.0, color_lr=0.0),
             'fine': dict(decoders_lr=0.0, coarse_lr=0.0, middle_lr=0.005, fine_lr=0.005, color_lr=0.0),
             'color': dict(decoders_lr=0.005, coarse_lr=0.0, middle_lr=0.005, fine_lr=0.005, color_lr=0.005)}
    return {
        'dataset': 'replica_event', 'scale': 1, 'occupancy': True, 'coarse': True, 'data': {'dim': 3, 'input_folder': inp, 'event_folder': evf},
        'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 32, 'N_surface': 16, 'N_importance': 0},
        'model': {'c_dim': 32, 'coarse_bound_enlarge': 2, 'pos_embedding_method': 'fourier'},
        'grid_len': {'coarse': 0.8, 'middle': 0.32, 'fine': 0.16, 'color': 0.16, 'bound_divisible': 0.32},
        'cam': dict(cam, png_depth_scale=6553.5, crop_edge=0),
        'mapping': {'bound': [[-1.0, 1.1], [-0.9, 0.8], [-0.7, 0.6]], 'w_color_loss': 0.2, 'lr_factor': 1, 'lr_first_factor': float(env.get('LR_FIRST', 1)),
                    'BA': False, 'BA_cam_lr': 0.001, 'middle_iter_ratio': 0.4, 'fine_iter_ratio': 0.6, 'fix_fine': True,
                    'fix_color': False, 'pixels': int(env.get('MAP_PIXELS', 600)), 'iters_first': int(env.get('ITERS_FIRST', 400)),
                    'iters': int(env.get('MAP_ITERS', 30)), 'every_frame': int(env.get('EVERY', 2)), 'keyframe_every': 6,
                    'mapping_window_size': 5, 'frustum_feature_selection': True, 'stage': stage},
        'tracking': {'device': device, 'w_color_loss': 0.5, 'ignore_edge_W': 4, 'ignore_edge_H': 4, 'handle_dynamic': True,
                     'use_color_in_tracking': True, 'lr': float(env.get('TRACK_LR', 0.002)), 'pixels': int(env.get('TRACK_PIXELS', 1000)),
                     'iters': int(env.get('TRACK_ITERS', 40)), 'const_speed_assumption': True, 'gt_camera': False,
                     'graphed': bool(int(env.get('TRACK_GRAPHED', 0)))},
        'event': {'activate_events': False, 'blur': True, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                  'balancer': 0.025, 'scale_factor': 0.5},
    }


def write_demo_sequence(root, n, cam, step=0.012, yaw_deg=0.5):
    """n frames of a BoxRoom inside the demo bound along `trajectory`, written in the Replica_event layout (all-zero events).
    Returns ((input_folder, event_folder), poses)."""
    from . import datasets as D
    from .scene import scene_bound
    bound = scene_bound([[-1.0, 1.1], [-0.9, 0.8], [-0.7, 0.6]], 1.0, 0.32)
    room = BoxRoom.for_bound(bound, margin=0.12, seed=1)
    poses = trajectory(room, n, step=step, yaw_deg=yaw_deg)
    frames, events = [], []
    for i, c2w in enumerate(poses):
        col, dep = room.render(c2w.double(), cam)
        frames.append((col.numpy(), dep.numpy()))
        if i > 0:
            events.append(np.zeros((cam['H'], cam['W'], 2), dtype=np.uint8))
    return D.write_replica_event_sequence(root, frames, [p.numpy() for p in poses], 6553.5, events), poses
