"""CPU restatement of one camera iteration of the tracker, RGB-D part.  TEST INFRASTRUCTURE ONLY.
Pinned by tests/golden/tiny_tracker_iter.npz (produced by the reference's own statements,
tests/golden/make_golden_tracker.py).

Follows src/Tracker.py:141 (camera tensor -> pose; src/common.py:189-229), :161-162 (get_samples), :164-174 (rays
whose exit from the bound comes before the measured depth are dropped), :175-179 (render, uncertainty detached),
:184-195 (depth mask, uncertainty-weighted L1 depth + w * L1 colour), :197 (backward)."""
import torch

from . import render_oracle as R


def quad2rotation(quad):
    """common.py:189-212: rotation of an unnormalised quaternion (real, i, j, k)."""
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2.0 / (quad * quad).sum(-1)
    rot = torch.zeros(quad.shape[0], 3, 3, dtype=quad.dtype)
    rot[:, 0, 0] = 1 - two_s * (qj ** 2 + qk ** 2)
    rot[:, 0, 1] = two_s * (qi * qj - qk * qr)
    rot[:, 0, 2] = two_s * (qi * qk + qj * qr)
    rot[:, 1, 0] = two_s * (qi * qj + qk * qr)
    rot[:, 1, 1] = 1 - two_s * (qi ** 2 + qk ** 2)
    rot[:, 1, 2] = two_s * (qj * qk - qi * qr)
    rot[:, 2, 0] = two_s * (qi * qk - qj * qr)
    rot[:, 2, 1] = two_s * (qj * qk + qi * qr)
    rot[:, 2, 2] = 1 - two_s * (qi ** 2 + qj ** 2)
    return rot


def camera_from_tensor(t):
    """common.py:215-228 for a single 7-vector."""
    x = t[None]
    return torch.cat([quad2rotation(x[:, :4]), x[:, 4:, None]], 2)[0]


def inside_prefilter(rays_o, rays_d, gt_depth, bound):
    """Tracker.py:166-170 / Mapper.py:537-543."""
    with torch.no_grad():
        o = rays_o.detach().unsqueeze(-1)
        d = rays_d.detach().unsqueeze(-1)
        t = (bound.unsqueeze(0) - o) / d
        t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
        return t >= gt_depth


def tracker_loss(depth, uncertainty, color, gt_depth, gt_color, w_color):
    """Tracker.py:179-195 with handle_dynamic off and use_color_in_tracking on."""
    uncertainty = uncertainty.detach()
    mask = gt_depth > 0
    loss = (torch.abs(gt_depth - depth) / torch.sqrt(uncertainty + 1e-10))[mask].sum()
    return loss + w_color * torch.abs(gt_color - color)[mask].sum()


def camera_iteration(params, grids, bound, camera_tensor, gt_depth_img, gt_color_img, cam, edge, n, w_color, idx=None):
    """Returns (loss, depth, uncertainty, color, inside_mask); call .backward() on the loss for d loss / d camera_tensor."""
    H, W, fx, fy, cx, cy = cam
    He, We = edge
    c2w = camera_from_tensor(camera_tensor)
    ro, rd, gd, gc = R.sample_pixels(He, H - He, We, W - We, n, c2w, gt_depth_img, gt_color_img, fx, fy, cx, cy, idx=idx)
    inside = inside_prefilter(ro, rd, gd, bound.to(ro.dtype))
    ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
    depth, var, color = R.render_batch_ray(params, grids, rd, ro, 'color', bound, gt_depth=gd)
    return tracker_loss(depth, var, color, gd, gc, w_color), depth, var, color, inside
