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
