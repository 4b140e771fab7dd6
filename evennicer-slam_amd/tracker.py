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
import os

import torch

from . import _lib as L
from . import functional as EF
from .functional import _ptr, _require_hip, _stream


# the RGB-D term of a camera iteration through the fused launches (TrackerIteration._rgbd_loss_fused); ENSLAM_TRACKER_FUSED=0 keeps
# the launch-per-step route (gather, pose, prefilter and median in torch ops, losses.tracker_loss) -- same numbers
FUSED_ITERATION = os.environ.get('ENSLAM_TRACKER_FUSED', '1') == '1'


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


class _TrackerRays(torch.autograd.Function):
    """Head of the camera iteration in one launch (enslam_tracker_rays): pixel samples of the drawn indices, rays of the camera
    tensor, in-bound mask and the sampler's batch maxima; backward = enslam_pose_rays_bwd (ray gradients -> the 7 numbers)."""

    @staticmethod
    def forward(ctx, camera_tensor, idx, H0, W0, ww, depth, color, fx, fy, cx, cy, bound6, prefilter, draw_counter=None):
        _require_hip(camera_tensor, "camera_tensor")
        ct = camera_tensor.detach().contiguous().float().reshape(-1)
        if ct.numel() != 7:
            raise L.EnslamError(f"camera tensor must have 7 elements (quaternion, translation), got {tuple(camera_tensor.shape)}")
        # idx: int64 [n] -- or, with draw_counter (int32 [1] on the device), [n_draws, n] drawn ahead: row counter % n_draws is used
        n_draws = int(idx.shape[0]) if draw_counter is not None else 0
        n, dev = int(idx.shape[-1]), ct.device
        if idx.dtype != torch.int64 or idx.dim() != (2 if draw_counter is not None else 1):
            raise L.EnslamError(f"pixel indices must be int64 [n] (or [n_draws, n] with a draw counter), got {tuple(idx.shape)} {idx.dtype}")
        depth, color, idx = depth.contiguous(), color.contiguous(), idx.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        flat = torch.empty(2 * n + 6 * n + n + 3 * n + 2, **f32)            # one allocation: pix_i | pix_j | ro | rd | gd | gc | dmax
        pi, pj, ro, rd, gd, gc, dmax = flat.split([n, n, 3 * n, 3 * n, n, 3 * n, 2])
        ro, rd, gc = ro.view(n, 3), rd.view(n, 3), gc.view(n, 3)
        inside = torch.empty(n, dtype=torch.uint8, device=dev) if prefilter else None
        L.check(L.lib().enslam_tracker_rays(n, _ptr(ct), _ptr(idx), int(H0), int(W0), int(ww), int(depth.shape[1]), int(depth.shape[0]), _ptr(depth), _ptr(color),
                                            int(color.dtype == torch.float64), ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_float(cx),
                                            ctypes.c_float(cy), bound6, int(bool(prefilter)), _ptr(pi), _ptr(pj), _ptr(ro), _ptr(rd), _ptr(gd),
                                            _ptr(gc), _ptr(inside), _ptr(dmax), _ptr(draw_counter), n_draws, _stream()), "enslam_tracker_rays")
        ctx.keep = (ct, pi, pj, float(fx), float(fy), float(cx), float(cy), tuple(camera_tensor.shape))
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(gd, gc, dmax)
        if inside is not None:
            ctx.mark_non_differentiable(inside)
        return ro, rd, gd, gc, inside, dmax

    @staticmethod
    def backward(ctx, g_ro, g_rd, *_unused):
        ct, pi, pj, fx, fy, cx, cy, shape = ctx.keep
        if g_ro is None and g_rd is None:
            return (None,) * 14
        gro = g_ro.detach().contiguous().float() if g_ro is not None else None
        grd = g_rd.detach().contiguous().float() if g_rd is not None else None
        g = torch.empty(7, dtype=torch.float32, device=ct.device)
        L.check(L.lib().enslam_pose_rays_bwd(pi.numel(), _ptr(ct), _ptr(pi), _ptr(pj), ctypes.c_float(fx), ctypes.c_float(fy),
                                             ctypes.c_float(cx), ctypes.c_float(cy), _ptr(gro), _ptr(grd), _ptr(g), _stream()),
                "enslam_pose_rays_bwd")
        return (g.reshape(shape),) + (None,) * 13


def rays_from_camera_tensor(camera_tensor, i, j, fx, fy, cx, cy):
    """rays_o, rays_d float32 [n,3] through pixels (i = column, j = row) of the camera
    [qr,qi,qj,qk, tx,ty,tz]; differentiable in the camera tensor."""
    return _PoseRays.apply(camera_tensor, i, j, fx, fy, cx, cy)


def get_samples_from_camera_tensor(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, camera_tensor, depth, color, device):
    """`common.get_samples` with the camera tensor in place of c2w: n random pixels of the window (the same single
    torch.randint draw), their depth / colour samples and their rays."""
    from .common import get_sample_uv           # (same draw; one fused launch for the pixel samples on the GPU)
    i, j, d, c = get_sample_uv(H0, H1, W0, W1, n, depth, color, device=device)
    rays_o, rays_d = rays_from_camera_tensor(camera_tensor, i, j, fx, fy, cx, cy)
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
        # (int64 [n_draws, n] pixel indices, int32 [1] counter) when the draws of a frame are made ahead (GraphedCameraIteration)
        self.draws = None

    def draw_ahead(self, n_draws, batch_size, out=None):
        """Pixel indices of n_draws coming iterations in ONE torch.randint (the same distribution as the per-iteration draw of
        common.get_sample_uv, a different consumption of the generator's stream) + a device counter starting at 0."""
        n_pix = (self.H - 2 * self.ignore_edge_H) * (self.W - 2 * self.ignore_edge_W)
        if out is None:
            out = (torch.empty((n_draws, batch_size), dtype=torch.int64, device=self.device),
                   torch.zeros(1, dtype=torch.int32, device=self.device))
        draw = torch.randint(n_pix, tuple(out[0].shape), device=out[0].device)
        if tuple(draw.shape) != tuple(out[0].shape):    # (an injected draw of one iteration's size: the same pixels every iteration)
            draw = draw.reshape(-1, out[0].shape[1]).expand_as(out[0])
        out[0].copy_(draw)
        out[1].zero_()
        return out

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

    def _bound_on(self, device):
        """the scene bound on the rays' device (copied once: a host-to-device copy cannot be captured in a hipGraph)"""
        b = getattr(self, '_bound_dev', None)
        if b is None or b.device != device:
            b = self._bound_dev = self.bound.to(device)
        return b

    def prepare_event_frame(self, gt_event, gt_mask, pre_gt_color, scale_factor):
        """The per-frame part of the event term (Tracker.py:129-137,146): ground-truth events, event mask and the
        previous colour image at the event resolution.  Constant over the camera iterations of a frame."""
        from . import event as EV
        g = gt_event.permute(2, 0, 1)
        _, h, w = g.shape
        size = (int(scale_factor * h), int(scale_factor * w))
        if size[0] <= 0 or size[1] <= 0:
            raise AssertionError('Scale is too small, resized images would have no pixels')
        return (EV.resize_nearest(g, size).permute(1, 2, 0), EV.resize_nearest(gt_mask[None, :, :], size).permute(1, 2, 0),
                EV.resize_nearest(pre_gt_color.permute(2, 0, 1), size).permute(1, 2, 0))

    def _rgbd_loss(self, camera_tensor, gt_color, gt_depth, batch_size, static_shapes):
        """RGB-D term of one iteration (Tracker.py:160-195).  static_shapes: no boolean indexing (hipGraph capture) --
        rays the reference drops are rendered but carry no loss, and the sampler's batch maxima are taken over the
        kept rays, so the kept rays get exactly the samples of the reference's filtered batch."""
        from .losses import tracker_loss
        device = self.device
        H, W, fx, fy, cx, cy = self.H, self.W, self.fx, self.fy, self.cx, self.cy
        Wedge, Hedge = self.ignore_edge_W, self.ignore_edge_H
        if FUSED_ITERATION and self._fused_ok(camera_tensor, gt_color, gt_depth, batch_size):
            return self._rgbd_loss_fused(camera_tensor, gt_color, gt_depth, batch_size)
        ro, rd, b_depth, b_color = get_samples_from_camera_tensor(Hedge, H - Hedge, Wedge, W - Wedge, batch_size, H, W,
                                                                  fx, fy, cx, cy, camera_tensor, gt_depth, gt_color, device)
        inside = None
        if self.nice:                                                                               # :164-174
            with torch.no_grad():
                t = (self._bound_on(ro.device).unsqueeze(0) - ro.detach().unsqueeze(-1)) / rd.detach().unsqueeze(-1)
                t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
                inside = t >= b_depth
            if not static_shapes:
                rd, ro, b_depth, b_color = rd[inside], ro[inside], b_depth[inside], b_color[inside]
                inside = None
        prev = self.renderer.depth_max_override
        if inside is not None:
            m = torch.where(inside, b_depth.float(), b_depth.new_zeros(()).float()).max().reshape(1)
            self.renderer.depth_max_override = torch.cat([m, m * 1.2]).contiguous()
        try:
            depth, uncertainty, color = self.renderer.render_batch_ray(self.c, self.decoders, rd, ro, device, stage='color',
                                                                       gt_depth=b_depth)
        finally:
            self.renderer.depth_max_override = prev
        uncertainty = uncertainty.detach()
        keep = inside
        if self.handle_dynamic:                                                                     # :180-182
            with torch.no_grad():
                tmp = torch.abs(b_depth - depth) / torch.sqrt(uncertainty + 1e-10)
                if inside is None:
                    med = tmp.median()
                else:               # median (lower middle, like torch.median) over the kept rays, without a dynamic shape
                    srt = torch.sort(torch.where(inside, tmp, torch.full_like(tmp, float('inf')))).values
                    med = srt.gather(0, ((inside.sum() - 1) // 2).clamp(min=0).reshape(1))[0]
                dyn = tmp < 10 * med
                keep = dyn if keep is None else (keep & dyn)
        gd_loss = b_depth if keep is None else torch.where(keep, b_depth, torch.zeros_like(b_depth))   # the loss keeps gt_depth > 0
        return tracker_loss(depth, uncertainty, color, gd_loss, b_color, self.w_color_loss,
                            use_color=self.use_color_in_tracking)                                   # :187-195

    def _fused_ok(self, camera_tensor, gt_color, gt_depth, batch_size):
        from . import functional as EF
        return (camera_tensor.is_cuda and camera_tensor.numel() == 7 and torch.is_tensor(gt_depth) and torch.is_tensor(gt_color)
                and gt_depth.is_cuda and gt_color.is_cuda and gt_depth.dtype == torch.float32 and gt_depth.dim() == 2
                and gt_color.dim() == 3 and gt_color.shape[2] == 3 and tuple(gt_color.shape[:2]) == tuple(gt_depth.shape)
                and gt_color.dtype in (torch.float32, torch.float64) and not gt_depth.requires_grad and not gt_color.requires_grad
                and self.renderer.tracker_loss_ok(batch_size, gt_depth) and self.renderer.depth_max_override is None)

    def _rgbd_loss_fused(self, camera_tensor, gt_color, gt_depth, batch_size):
        """The RGB-D term (Tracker.py:160-195) in five launches forward -- the randint draw, enslam_tracker_rays (pixels, rays,
        in-bound mask, batch maxima), sampler, decoders, enslam_render_tracker_loss_fwd's tail (compositing + median mask + loss +
        unit gradients) -- and two backward (light decoder backward with the ray gradients, pose chain).  Same numbers as the
        route below it: the rays the reference drops are rendered but masked, the sampler's maxima and the median are taken over
        the kept rays."""
        from . import functional as EF
        H, W = self.H, self.W
        We, He = self.ignore_edge_W, self.ignore_edge_H
        ww = (W - We) - We
        counter = None
        if self.draws is not None and self.draws[0].shape[1] == batch_size:     # indices of the whole frame drawn ahead (captured iterations)
            idx, counter = self.draws
        else:
            idx = torch.randint(((H - He) - He) * ww, (batch_size,), device=camera_tensor.device)   # the draw of common.get_sample_uv
        ro, rd, gd, gc, inside, dmax = _TrackerRays.apply(camera_tensor, idx, He, We, ww, gt_depth, gt_color, self.fx, self.fy, self.cx,
                                                          self.cy, EF.bound6(self.bound), bool(self.nice), counter)
        prev = self.renderer.depth_max_override
        self.renderer.depth_max_override = dmax
        try:
            loss, _d, _u, _c = self.renderer.render_batch_ray_tracker_loss(self.c, self.decoders, rd, ro, self.device, 'color', gd, gc,
                                                                           self.w_color_loss, inside=inside,
                                                                           handle_dynamic=self.handle_dynamic,
                                                                           use_color=self.use_color_in_tracking)
        finally:
            self.renderer.depth_max_override = prev
        return loss

    def iteration_losses(self, camera_tensor, gt_color, gt_depth, frame, batch_size, rgbd=True, event=True,
                         scale_factor=0.1, static_shapes=False):
        """Loss tensors of one camera iteration, nothing synchronised: dict with `total` (what is back-propagated;
        None when nothing is), `rgbd`, `event` (balanced), `mask`, `full_event`, `event_mask`, `gts_blurred`,
        `preds_blurred`, `terms`.  `frame` = prepare_event_frame(...) (needed when event)."""
        from . import event as EV
        out = dict(total=None, rgbd=None, event=None, mask=None, full_event=None, event_mask=None, gts_blurred=[],
                   preds_blurred=[], terms=[])
        if event:
            if self.event_net is None:
                raise RuntimeError("event=True needs slam.event_net")
            gt_event, gt_mask, full_color_previous = frame
            if self.low_gpu_mem and not static_shapes:
                torch.cuda.empty_cache()
            full_color_current = self._render_rescaled(camera_tensor, gt_depth, scale_factor)      # :150
            out['full_event'], out['event_mask'] = EV.inference_event(
                net=self.event_net, img1=full_color_previous, img2=full_color_current, device=self.device, scale_factor=1.0,
                out_threshold=0.5)                                                                  # :153
        if rgbd:
            out['rgbd'] = out['total'] = self._rgbd_loss(camera_tensor, gt_color, gt_depth, batch_size, static_shapes)
        if event:
            loss_event, out['gts_blurred'], out['preds_blurred'], out['terms'] = EV.event_loss(     # :206-221
                gt_event, out['full_event'], self.blur, self.kernel_sizes, self.unblurred_weight, self.kernel_weights)
            out['mask'] = torch.nn.functional.cross_entropy(out['event_mask'], gt_mask.permute(2, 0, 1).long())   # :224-225
            out['event'] = loss_event * self.cfg['event']['balancer']                               # :228-229
            if self.activate_events:
                out['total'] = out['event'] if out['total'] is None else out['total'] + out['event'].to(out['total'].dtype)
        return out

    def optimize_cam_in_batch(self, camera_tensor, pre_c2w, gt_color, gt_depth, gt_event, gt_mask, batch_size, optimizer,
                              idx, iter, pre_gt_color, rgbd=True, event=True, scale_factor=0.1):
        """One camera iteration (Tracker.py:104-245): sample pixels, render, RGB-D loss and/or event loss, backward,
        optimiser step.  Returns the reference's tuple:
        blur off: (loss_rgbd, loss_event, loss_mask, gt_event, full_event, gt_mask, P(event)[h,w]);
        blur on : (loss_rgbd, loss_event, loss_mask, gt_event, full_event, gts_blurred, preds_blurred, term_values,
                   gt_mask, P(event)[h,w]).  With `event=False` the event entries are None (the reference raises a
        NameError on its undefined lists in that case when blur is on)."""
        optimizer.zero_grad()
        frame = None
        if event:
            frame = self.prepare_event_frame(gt_event, gt_mask, pre_gt_color, scale_factor)
            gt_event, gt_mask = frame[0], frame[1]
        o = self.iteration_losses(camera_tensor, gt_color, gt_depth, frame, batch_size, rgbd, event, scale_factor)
        if o['total'] is not None and o['total'].requires_grad:
            with EF.engine_on_calling_thread():
                o['total'].backward()                                                               # :197-199,231-232
        optimizer.step()
        optimizer.zero_grad()
        item = lambda x: None if x is None else float(x.item())
        loss_rgbd_item, loss_event_item, loss_mask_item = item(o['rgbd']), item(o['event']), item(o['mask'])
        p_event = o['event_mask'][0][1] if o['event_mask'] is not None else None
        if event and not self.blur:
            return loss_rgbd_item, loss_event_item, loss_mask_item, gt_event, o['full_event'], gt_mask, p_event
        terms = [float(x.item()) if torch.is_tensor(x) else float(x) for x in o['terms']]
        return (loss_rgbd_item, loss_event_item, loss_mask_item, gt_event, o['full_event'], o['gts_blurred'],
                o['preds_blurred'], terms, gt_mask, p_event)


class GraphedCameraIteration(object):
    """The camera iterations of one frame as replays of ONE hipGraph (zero_grad -> losses -> backward -> optimiser
    step), using the static-shape formulation of `TrackerIteration.iteration_losses`.  `optimizer` must be capturable
    (`mapper.FusedAdam`).  Per frame: `set_frame(images...)` copies the images into the graph's input buffers and
    prepares the event-resolution ground truth; per iteration: `step()` returns the (device) loss tensors."""

    def __init__(self, trk, camera_tensor, optimizer, gt_color, gt_depth, gt_event=None, gt_mask=None, pre_gt_color=None,
                 batch_size=200, rgbd=True, event=True, scale_factor=0.1, warmup=3, n_draws=None):
        from .graph import GraphedStep
        self.trk, self.event, self.scale_factor = trk, event, scale_factor
        self.gt_color, self.gt_depth = gt_color.clone(), gt_depth.clone()
        self.frame = None
        if event:
            self.frame = tuple(t.clone() for t in trk.prepare_event_frame(gt_event, gt_mask, pre_gt_color, scale_factor))
        # pixel draws of a whole frame made ahead, eagerly, in set_frame(): a randint inside the graph costs its launch and two
        # fill launches per replay (torch refreshes the captured generator's seed and offset in front of every replay)
        self.n_draws = int(n_draws if n_draws is not None else trk.cfg['tracking'].get('iters', 10))
        self._draws = trk.draw_ahead(self.n_draws, batch_size) if (rgbd and FUSED_ITERATION) else None
        self._zero = None

        def it():
            optimizer.zero_grad()
            o = trk.iteration_losses(camera_tensor, self.gt_color, self.gt_depth, self.frame, batch_size, rgbd, event,
                                     scale_factor, static_shapes=True)
            if 'one' not in self.__dict__:
                self.one = torch.ones_like(o['total'])
            with EF.engine_on_calling_thread():
                o['total'].backward(gradient=self.one)
            optimizer.step()
            if self._zero is None:
                self._zero = o['total'].new_zeros(())
            return tuple(self._zero if o[k] is None else o[k].detach() for k in ('rgbd', 'event', 'mask'))

        trk.draws = self._draws                          # (only while the iteration is recorded: eager calls keep their own draw)
        try:
            self.graph = GraphedStep(it, warmup=warmup)
        finally:
            trk.draws = None
        if self._draws is not None:
            self._draws[1].zero_()                       # (the warm-up iterations consumed rows)
        self._steps_since_draw = 0
        # The captured iteration reads the device-side forms (voxel-major copies, packed decoders) of exactly these
        # objects: the graph owns them from here on (refresh_map copies a replaced map INTO them).
        self._map_c = {k: trk.c[k] for k in ('grid_middle', 'grid_fine', 'grid_color')}
        self._map_decoders = trk.decoders

    def set_frame(self, gt_color, gt_depth, gt_event=None, gt_mask=None, pre_gt_color=None):
        self.gt_color.copy_(gt_color)
        self.gt_depth.copy_(gt_depth)
        if self._draws is not None:
            self.trk.draw_ahead(self.n_draws, self._draws[0].shape[1], out=self._draws)      # fresh pixels for the new frame
        self._steps_since_draw = 0
        if self.event:
            for dst, src in zip(self.frame, self.trk.prepare_event_frame(gt_event, gt_mask, pre_gt_color, self.scale_factor)):
                dst.copy_(src)

    def refresh_map(self, c=None, decoders=None):
        """Make the captured iteration see a new map.  `c` / `decoders` default to what the tracker holds now
        (`trk.c`, `trk.decoders`).  The reference's `Tracker.update_para_from_mapping` (Tracker.py:247-259) REPLACES
        its map per frame (`self.c[key] = val.clone()`, `self.decoders = copy.deepcopy(shared_decoders)`); an in-place
        update (`trk.c[key].copy_(val)`) works as well.  Either way the new values are copied into the tensors and the
        decoder module the graph was captured with, the tracker is pointed back at those objects, and their cached
        voxel-major / packed forms are rewritten in place.  Returns the number of device buffers rewritten (>= 1 unless
        nothing changed); raises when the new map does not have the captured shapes."""
        from . import _lib as L
        from .functional import refresh_in_place
        trk = self.trk
        c = trk.c if c is None else c
        decoders = trk.decoders if decoders is None else decoders
        with torch.no_grad():
            for k, cap in self._map_c.items():
                new = c[k]
                if new is not cap:
                    if tuple(new.shape) != tuple(cap.shape):
                        raise L.EnslamError(f"refresh_map: {k} has shape {tuple(new.shape)}, the captured iteration was recorded "
                                            f"with {tuple(cap.shape)}; capture a new GraphedCameraIteration")
                    cap.copy_(new)
            if decoders is not self._map_decoders:
                mine = dict(self._map_decoders.named_parameters())
                theirs = dict(decoders.named_parameters())
                if set(mine) != set(theirs):
                    raise L.EnslamError("refresh_map: the new decoders do not have the captured module's parameters")
                for name, p in mine.items():
                    p.copy_(theirs[name])
                for name in ('bound',):                                  # plain attributes the renderer reads
                    for sub in ('', 'coarse_decoder', 'middle_decoder', 'fine_decoder', 'color_decoder'):
                        src = getattr(decoders, sub) if sub else decoders
                        dst = getattr(self._map_decoders, sub) if sub else self._map_decoders
                        if hasattr(src, name):
                            setattr(dst, name, getattr(src, name))
        if trk.c is None:
            trk.c = {}
        for k, cap in self._map_c.items():
            trk.c[k] = cap
        trk.decoders = self._map_decoders
        return refresh_in_place(trk.c, trk.decoders, 'color')

    def step(self):
        # The captured iteration takes row `counter % n_draws` of the pixel indices drawn ahead: once every row has been used
        # (more step() calls since set_frame() than n_draws) fresh rows are drawn -- one eager randint -- instead of silently
        # reusing the same pixel batches.
        if self._draws is not None and self._steps_since_draw >= self.n_draws:
            self.trk.draw_ahead(self.n_draws, self._draws[0].shape[1], out=self._draws)
            self._steps_since_draw = 0
        self._steps_since_draw += 1
        return self.graph.replay()
