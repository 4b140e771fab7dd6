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


def image_rays_from_camera_tensor(camera_tensor, H, W, new_H, new_W, fx, fy, cx, cy, device):
    """Rays through the (new_H, new_W) strided pixel centres of the image (`common.get_rays_rescale`, the ray set of
    `render_img_rescale`) straight from the camera tensor: one launch, gradients to the 7 numbers in one more."""
    cols = torch.linspace(0, W - 1, new_W, device=device)
    rows = torch.linspace(0, H - 1, new_H, device=device)
    return rays_from_camera_tensor(camera_tensor, cols.repeat(new_H), rows.repeat_interleave(new_W), fx, fy, cx, cy)


class TrackerIteration(object):
    """The camera-iteration half of the reference's `Tracker` (src/Tracker.py:23-245): same configuration keys, same
    `optimize_cam_in_batch` signature and return values; the frame loop, visualiser, logging and the mapper hand-shake
    stay with the caller.  `slam` supplies `bound, renderer, event_net, H, W, fx, fy, cx, cy` (and optionally `nice`,
    `low_gpu_mem`); `self.c` / `self.decoders` are assigned by the caller like `Tracker.update_para_from_mapping` does.

    Differences a caller can see: none in the numbers.  Internally pose -> rays and the loss are fused launches and
    both loss terms are back-propagated in one pass (their graphs only share the camera tensor)."""

    def __init__(self, cfg, args, slam):
        self.cfg, self.args = cfg, args
        t, e = cfg['tracking'], cfg['event']
        self.device = t['device']
        self.w_color_loss = t['w_color_loss']
        self.ignore_edge_W, self.ignore_edge_H = t['ignore_edge_W'], t['ignore_edge_H']
        self.handle_dynamic = t['handle_dynamic']
        self.use_color_in_tracking = t['use_color_in_tracking']
        self.nice = getattr(slam, 'nice', True)
        self.low_gpu_mem = getattr(slam, 'low_gpu_mem', False)
        self.bound = slam.bound
        self.renderer = slam.renderer
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = slam.H, slam.W, slam.fx, slam.fy, slam.cx, slam.cy
        self.event_net = getattr(slam, 'event_net', None)
        self.activate_events = e['activate_events']
        self.blur = e['blur']
        self.kernel_sizes, self.kernel_weights = e['kernel_sizes'], e['kernel_weights']
        self.unblurred_weight = e['unblurred_weight']
        self.c = None
        self.decoders = None

    def _render_rescaled(self, camera_tensor, gt_depth, scale_factor):
        """`renderer.render_img_rescale(c, decoders, get_camera_from_tensor(camera_tensor), ...)`'s colour image."""
        from .event import resize_bilinear
        H, W = self.H, self.W
        new_H, new_W = int(H * scale_factor), int(W * scale_factor)
        rays_o, rays_d = image_rays_from_camera_tensor(camera_tensor, H, W, new_H, new_W, self.fx, self.fy, self.cx, self.cy,
                                                       self.device)
        gd = resize_bilinear(gt_depth[None].float(), (new_H, new_W)).reshape(-1) if gt_depth is not None else None
        _, _, color = self.renderer._render_chunks(self.c, self.decoders, rays_o, rays_d, self.device, 'color', gd)
        return color.reshape(new_H, new_W, 3)

    def optimize_cam_in_batch(self, camera_tensor, pre_c2w, gt_color, gt_depth, gt_event, gt_mask, batch_size, optimizer,
                              idx, iter, pre_gt_color, rgbd=True, event=True, scale_factor=0.1):
        """One camera iteration (Tracker.py:104-245): sample pixels, render, RGB-D loss and/or event loss, backward,
        optimiser step.  Returns the reference's tuple:
        blur off: (loss_rgbd, loss_event, loss_mask, gt_event, full_event, gt_mask, P(event)[h,w]);
        blur on : (loss_rgbd, loss_event, loss_mask, gt_event, full_event, gts_blurred, preds_blurred, term_values,
                   gt_mask, P(event)[h,w]).  With `event=False` the event entries are None (the reference raises a
        NameError on its undefined lists in that case when blur is on)."""
        from . import event as EV
        from .losses import tracker_loss
        device = self.device
        H, W, fx, fy, cx, cy = self.H, self.W, self.fx, self.fy, self.cx, self.cy
        optimizer.zero_grad()
        full_event = event_mask = None
        gts_event_list, preds_event_list, losses_event_list = [], [], []
        if event:
            if self.event_net is None:
                raise RuntimeError("event=True needs slam.event_net")
            g = gt_event.permute(2, 0, 1)
            _, h, w = g.shape
            size = (int(scale_factor * h), int(scale_factor * w))
            if size[0] <= 0 or size[1] <= 0:
                raise AssertionError('Scale is too small, resized images would have no pixels')
            gt_event = EV.resize_nearest(g, size).permute(1, 2, 0)                                  # :133-134
            gt_mask = EV.resize_nearest(gt_mask[None, :, :], size).permute(1, 2, 0)                 # :137
            full_color_previous = EV.resize_nearest(pre_gt_color.permute(2, 0, 1), size).permute(1, 2, 0)   # :146
            if self.low_gpu_mem:
                torch.cuda.empty_cache()
            full_color_current = self._render_rescaled(camera_tensor, gt_depth, scale_factor)      # :150
            full_event, event_mask = EV.inference_event(net=self.event_net, img1=full_color_previous,
                                                        img2=full_color_current, device=device, scale_factor=1.0,
                                                        out_threshold=0.5)                           # :153
        total = None
        loss_rgbd = None
        if rgbd:
            Wedge, Hedge = self.ignore_edge_W, self.ignore_edge_H
            ro, rd, b_depth, b_color = get_samples_from_camera_tensor(Hedge, H - Hedge, Wedge, W - Wedge, batch_size, H, W,
                                                                      fx, fy, cx, cy, camera_tensor, gt_depth, gt_color, device)
            if self.nice:                                                                           # :164-174
                with torch.no_grad():
                    t = (self.bound.unsqueeze(0).to(device) - ro.detach().unsqueeze(-1)) / rd.detach().unsqueeze(-1)
                    t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
                    inside = t >= b_depth
                rd, ro, b_depth, b_color = rd[inside], ro[inside], b_depth[inside], b_color[inside]
            depth, uncertainty, color = self.renderer.render_batch_ray(self.c, self.decoders, rd, ro, device, stage='color',
                                                                       gt_depth=b_depth)
            uncertainty = uncertainty.detach()
            gd_loss = b_depth
            if self.handle_dynamic:                                                                 # :180-182
                with torch.no_grad():
                    tmp = torch.abs(b_depth - depth) / torch.sqrt(uncertainty + 1e-10)
                    keep = tmp < 10 * tmp.median()
                    gd_loss = torch.where(keep, b_depth, torch.zeros_like(b_depth))                 # the loss keeps gt_depth > 0
            loss_rgbd = tracker_loss(depth, uncertainty, color, gd_loss, b_color, self.w_color_loss,
                                     use_color=self.use_color_in_tracking)                           # :187-195
            total = loss_rgbd
        loss_event = loss_mask = None
        if event:
            loss_event, gts_event_list, preds_event_list, terms = EV.event_loss(                    # :206-221
                gt_event, full_event, self.blur, self.kernel_sizes, self.unblurred_weight, self.kernel_weights)
            losses_event_list = terms
            loss_mask = torch.nn.functional.cross_entropy(event_mask, gt_mask.permute(2, 0, 1).long())      # :224-225
            loss_event = loss_event * self.cfg['event']['balancer']                                 # :228-229
            if self.activate_events:
                total = loss_event if total is None else total + loss_event.to(total.dtype)
        if total is not None and total.requires_grad:
            total.backward()                                                                        # :197-199,231-232
        optimizer.step()
        optimizer.zero_grad()
        item = lambda x: None if x is None else float(x.item())
        loss_rgbd_item, loss_event_item, loss_mask_item = item(loss_rgbd), item(loss_event), item(loss_mask)
        p_event = event_mask[0][1] if event_mask is not None else None
        if event and not self.blur:
            return loss_rgbd_item, loss_event_item, loss_mask_item, gt_event, full_event, gt_mask, p_event
        losses_event_list = [float(x.item()) if torch.is_tensor(x) else float(x) for x in losses_event_list]
        return (loss_rgbd_item, loss_event_item, loss_mask_item, gt_event, full_event, gts_event_list, preds_event_list,
                losses_event_list, gt_mask, p_event)
