"""Fused mapper loss (first piece of the mapper glue).

`rgbd_loss(depth, color, gt_depth, gt_color, w_color)` = sum_{gt_depth>0} |gt_depth - depth| + w_color * sum |gt_color - color|
(reference: src/Mapper.py:553-562; pass color=None outside the colour stage).  The torch formulation issues ~24 tiny
kernels and a host sync (boolean-mask indexing); this is one HIP kernel each way and graph-capturable."""
import ctypes

import torch

from . import _lib as L
from .functional import _ptr, _require_hip, _stream


class _RgbdLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, color, gt_depth, gt_color, w_color):
        _require_hip(depth, "depth")
        d = depth.detach().contiguous().double()
        c = color.detach().contiguous().float() if color is not None else None
        gd = gt_depth.detach().contiguous().float()
        gc = gt_color.detach().contiguous().float() if color is not None else None
        loss = torch.empty(1, dtype=torch.float64, device=depth.device)
        L.check(L.lib().enslam_rgbd_loss_fwd(d.shape[0], _ptr(d), _ptr(c), _ptr(gd), _ptr(gc), ctypes.c_float(w_color),
                                             _ptr(loss), _stream()), "enslam_rgbd_loss_fwd")
        ctx.keep = (d, c, gd, gc, float(w_color))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d, c, gd, gc, w = ctx.keep
        g1 = g.detach().double().reshape(1).contiguous()
        g_depth = torch.empty_like(d)
        g_color = torch.empty_like(c) if c is not None else None
        L.check(L.lib().enslam_rgbd_loss_bwd(d.shape[0], _ptr(d), _ptr(c), _ptr(gd), _ptr(gc), ctypes.c_float(w), _ptr(g1),
                                             _ptr(g_depth), _ptr(g_color), _stream()), "enslam_rgbd_loss_bwd")
        return g_depth, g_color, None, None, None


def rgbd_loss(depth, color, gt_depth, gt_color, w_color=0.2):
    if color is None:
        gt_color = None
    return _RgbdLoss.apply(depth, color, gt_depth, gt_color, w_color)


class _TrackerLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, uncertainty, color, gt_depth, gt_color, w_color):
        _require_hip(depth, "depth")
        d = depth.detach().contiguous().double()
        u = uncertainty.detach().contiguous().double()
        c = color.detach().contiguous().float() if color is not None else None
        gd = gt_depth.detach().contiguous().float()
        gc = gt_color.detach().contiguous().float() if color is not None else None
        loss = torch.empty(1, dtype=torch.float64, device=depth.device)
        L.check(L.lib().enslam_tracker_loss_fwd(d.shape[0], _ptr(d), _ptr(u), _ptr(c), _ptr(gd), _ptr(gc),
                                                ctypes.c_float(w_color), _ptr(loss), _stream()), "enslam_tracker_loss_fwd")
        ctx.keep = (d, u, c, gd, gc, float(w_color))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d, u, c, gd, gc, w = ctx.keep
        g1 = g.detach().double().reshape(1).contiguous()
        g_depth = torch.empty_like(d)
        g_color = torch.empty_like(c) if c is not None else None
        L.check(L.lib().enslam_tracker_loss_bwd(d.shape[0], _ptr(d), _ptr(u), _ptr(c), _ptr(gd), _ptr(gc), ctypes.c_float(w),
                                                _ptr(g1), _ptr(g_depth), _ptr(g_color), _stream()), "enslam_tracker_loss_bwd")
        return g_depth, None, g_color, None, None, None


def tracker_loss(depth, uncertainty, color, gt_depth, gt_color, w_color=0.5, use_color=True):
    """Tracker.py:179-195 (handle_dynamic off): sum over rays with gt_depth > 0 of |gt_depth - depth| / sqrt(uncertainty
    + 1e-10), plus w_color * |gt_color - color| over the same rays when use_color (use_color_in_tracking).  The
    uncertainty is treated as a constant, as the reference detaches it (:179)."""
    if not use_color:
        color = gt_color = None
    return _TrackerLoss.apply(depth, uncertainty, color, gt_depth, gt_color, w_color)
