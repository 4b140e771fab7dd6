"""Tracker camera-iteration glue (RGB-D part).

Reference (src/Tracker.py:141-197): every camera iteration turns the 7-vector camera tensor into a pose
(`get_camera_from_tensor`, src/common.py:189-229), samples pixels and builds their rays (`get_samples`, :160-169),
renders them and back-propagates an uncertainty-weighted loss to the 7 numbers.  In PyTorch the quaternion algebra
and its autograd are ~200 launches on tensors of 1-9 elements -- several times the cost of rendering 200 rays.

`rays_from_camera_tensor` does pose + rays in one launch and their backward (ray gradients -> 7 numbers) in another;
`get_samples_from_camera_tensor` mirrors `get_samples` with the camera tensor in place of c2w (same single RNG
draw); `losses.tracker_loss` is the fused loss.  The plain-torch route (`common.get_camera_from_tensor` +
`common.get_samples`) stays available and gives the same numbers."""
import ctypes

import torch

from . import _lib as L
from .functional import _ptr, _require_hip, _stream


class _PoseRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, camera_tensor, i, j, fx, fy, cx, cy):
        _require_hip(camera_tensor, "camera_tensor")
        ct = camera_tensor.detach().contiguous().float().reshape(-1)
        if ct.numel() != 7:
            raise L.EnslamError(f"camera tensor must have 7 elements (quaternion, translation), got {tuple(camera_tensor.shape)}")
        pi = i.detach().contiguous().float().reshape(-1)
        pj = j.detach().contiguous().float().reshape(-1)
        n = pi.numel()
        ro = torch.empty((n, 3), dtype=torch.float32, device=ct.device)
        rd = torch.empty((n, 3), dtype=torch.float32, device=ct.device)
        L.check(L.lib().enslam_pose_rays_fwd(n, _ptr(ct), _ptr(pi), _ptr(pj), ctypes.c_float(fx), ctypes.c_float(fy),
                                             ctypes.c_float(cx), ctypes.c_float(cy), _ptr(ro), _ptr(rd), _stream()),
                "enslam_pose_rays_fwd")
        ctx.keep = (ct, pi, pj, float(fx), float(fy), float(cx), float(cy), tuple(camera_tensor.shape))
        ctx.set_materialize_grads(False)
        return ro, rd

    @staticmethod
    def backward(ctx, g_ro, g_rd):
        ct, pi, pj, fx, fy, cx, cy, shape = ctx.keep
        if g_ro is None and g_rd is None:
            return (None,) * 7
        gro = g_ro.detach().contiguous().float() if g_ro is not None else None
        grd = g_rd.detach().contiguous().float() if g_rd is not None else None
        g = torch.empty(7, dtype=torch.float32, device=ct.device)
        L.check(L.lib().enslam_pose_rays_bwd(pi.numel(), _ptr(ct), _ptr(pi), _ptr(pj), ctypes.c_float(fx), ctypes.c_float(fy),
                                             ctypes.c_float(cx), ctypes.c_float(cy), _ptr(gro), _ptr(grd), _ptr(g), _stream()),
                "enslam_pose_rays_bwd")
        return g.reshape(shape), None, None, None, None, None, None


def rays_from_camera_tensor(camera_tensor, i, j, fx, fy, cx, cy):
    """rays_o, rays_d float32 [n,3] through pixels (i = column, j = row) of the camera
    [qr,qi,qj,qk, tx,ty,tz]; differentiable in the camera tensor."""
    return _PoseRays.apply(camera_tensor, i, j, fx, fy, cx, cy)


def get_samples_from_camera_tensor(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, camera_tensor, depth, color, device):
    """`common.get_samples` with the camera tensor in place of c2w: n random pixels of the window (the same single
    torch.randint draw), their depth / colour samples and their rays."""
    cols = torch.linspace(W0, W1 - 1, W1 - W0, device=device)
    rows = torch.linspace(H0, H1 - 1, H1 - H0, device=device)
    ww = W1 - W0
    idx = torch.randint((H1 - H0) * ww, (n,), device=device)
    col, row = idx % ww, idx // ww
    d = depth[H0:H1, W0:W1][row, col]
    c = color[H0:H1, W0:W1][row, col]
    rays_o, rays_d = rays_from_camera_tensor(camera_tensor, cols[col], rows[row], fx, fy, cx, cy)
    return rays_o, rays_d, d, c
