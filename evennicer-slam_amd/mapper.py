"""Mapper inner-iteration glue on the device layout.

Reference (src/Mapper.py): per `optimize_map` call the frustum-masked part of every grid becomes a compact leaf
`val_grad = val[mask]` (:343-361) optimised by `torch.optim.Adam` with per-stage learning rates (:396-419, :466-473);
EVERY iteration re-materialises the full grid (`val[mask] = val_grad`, :448-458 -- an index_put over up to 22.8 MB
per grid, whose backward gathers the dense gradient back), steps Adam (:573-575) and writes the result back
(:596-602).

Here the grid lives in the kernels' voxel-major layout for the whole call: the render kernels read it and add their
gradients into a persistent accumulator directly (functional.VoxelMajorGrid), and ONE launch per iteration applies
Adam to the masked voxels of all grids and clears the accumulators (enslam_adam_masked).  No per-iteration layout
conversion, index_put, dense-gradient gather, block marking or clearing.  `write_back()` returns the optimised
voxels to the caller's `[1,32,D,H,W]` tensors (`val[mask] = val_grad`, :596-602)."""
import ctypes

import torch

from . import _lib as L
from .functional import (VoxelMajorGrid, _native_vm, _ptr, _require_hip, _stream, engine_on_calling_thread, is_native_grid,
                         note_raw_write)

GRID_KEYS = ('grid_coarse', 'grid_middle', 'grid_fine', 'grid_color')


class MaskedGridOptimizer:
    """Adam over the masked voxels of feature grids, state and parameters in voxel-major layout.

    c      : dict key -> float32 [1,32,D,H,W] HIP tensor (the shared map `self.c` of the reference)
    masks  : dict key -> bool tensor [D,H,W] or the reference's channel-repeated [1,32,D,H,W] (Mapper.py:343-346);
             a missing key / None optimises every voxel (frustum_feature_selection off, :330-341)
    keys   : the grids this mapper optimises (coarse mapper: ('grid_coarse',); else middle, fine, color -- :326-328)
    """

    def __init__(self, c, masks=None, keys=('grid_middle', 'grid_fine', 'grid_color'), betas=(0.9, 0.999), eps=1e-8):
        lib = L.lib()
        self.c, self.keys = c, tuple(keys)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.grids, self.mask, self.mask5, self.m, self.v = {}, {}, {}, {}, {}
        self.native = set()
        masks = masks or {}
        dev = None
        for key in self.keys:
            g = c[key]
            _require_hip(g, key)
            if g.dim() != 5 or g.shape[0] != 1 or g.shape[1] != 32 or g.dtype != torch.float32:
                raise L.EnslamError(f"{key}: expected float32 [1,32,D,H,W], got {tuple(g.shape)} {g.dtype}")
            dev = g.device
            D, H, W = (int(x) for x in g.shape[2:])
            V = D * H * W
            if is_native_grid(g):
                # a channels_last_3d grid IS [V,32] in memory: the optimiser works on the caller's tensor itself (its masked
                # voxels change in place every step, as `val[mask] = val_grad; c[key] = val` does in the reference's loop,
                # Mapper.py:448-458) and write_back() has nothing left to copy
                vm = _native_vm(g)
                self.native.add(key)
            else:
                vm = torch.empty((V, 32), dtype=torch.float32, device=dev)
                L.check(lib.enslam_grid_to_voxel_major(_ptr(g.detach().contiguous()), _ptr(vm), V, _stream()),
                        "enslam_grid_to_voxel_major")
            self.grids[key] = VoxelMajorGrid((D, H, W), vm, torch.zeros((V, 32), dtype=torch.float32, device=dev))
            mk = masks.get(key)
            if mk is not None:
                mk = mk.to(dev)
                if mk.dim() == 5:                       # channel-repeated mask of the reference: any channel
                    mk = mk[0, 0]
                if tuple(mk.shape) != (D, H, W):
                    raise L.EnslamError(f"{key}: mask shape {tuple(mk.shape)} does not match the grid {(D, H, W)}")
                self.mask5[key] = mk.bool()
                self.mask[key] = mk.reshape(-1).to(torch.uint8).contiguous()
            else:
                self.mask5[key] = self.mask[key] = None
            self.m[key] = torch.zeros((V, 32), dtype=torch.float32, device=dev)
            self.v[key] = torch.zeros((V, 32), dtype=torch.float32, device=dev)
        n = len(self.keys)
        # device scalars the launch reads: learning rate (float64) and step count (int32) of each grid
        self.lr_t = torch.zeros(n, dtype=torch.float64, device=dev)
        self.step_t = torch.zeros(n, dtype=torch.int32, device=dev)
        self._lr_host = [None] * n
        self.steps = [0] * n                           # host mirror of step_t

    # ---- what the renderer gets in place of the reference's `c`
    def render_grids(self, extra=None):
        """dict to pass as `c` to Renderer.render_batch_ray: optimised grids in the device layout, every other key of
        the caller's `c` (or `extra`) unchanged."""
        out = dict(self.c)
        if extra:
            out.update(extra)
        out.update(self.grids)
        return out

    def set_lr(self, lrs):
        """lrs: dict key -> learning rate for the coming steps (cfg['mapping']['stage'][stage][...]*lr_factor,
        Mapper.py:466-473).  Only a changed value costs a (tiny, stream-ordered) copy."""
        host = [float(lrs.get(k, 0.0)) for k in self.keys]
        if host != self._lr_host:
            self.lr_t.copy_(torch.tensor(host, dtype=torch.float64), non_blocking=False)
            self._lr_host = host

    def step(self, lrs=None):
        """optimizer.step() + optimizer.zero_grad() for the grids (Mapper.py:575, 594): Adam on the masked voxels of
        every grid that has received a gradient so far (torch.optim.Adam skips parameters whose grad is None; the
        stages only ever add grids), then all accumulators are cleared."""
        if lrs is not None:
            self.set_lr(lrs)
        lib = L.lib()
        n = len(self.keys)
        inc = [1 if self.grids[k].has_grad else 0 for k in self.keys]
        if any(inc):
            if all(inc):
                self.step_t += 1
            else:
                self.step_t += torch.tensor(inc, dtype=torch.int32, device=self.step_t.device)
            self.steps = [s + i for s, i in zip(self.steps, inc)]
        arr = lambda: (ctypes.c_void_p * n)()
        p, g, m, v, mk, lr, st = arr(), arr(), arr(), arr(), arr(), arr(), arr()
        vs = (ctypes.c_int64 * n)()
        for i, k in enumerate(self.keys):
            G = self.grids[k]
            p[i], g[i], m[i], v[i] = G.vm.data_ptr(), G.grad_vm.data_ptr(), self.m[k].data_ptr(), self.v[k].data_ptr()
            mk[i] = self.mask[k].data_ptr() if self.mask[k] is not None else None
            vs[i] = G.vm.shape[0]
            lr[i] = self.lr_t.data_ptr() + 8 * i
            st[i] = self.step_t.data_ptr() + 4 * i
        L.check(lib.enslam_adam_masked(n, p, g, m, v, mk, vs, lr, st, self.betas[0], self.betas[1], self.eps, _stream()),
                "enslam_adam_masked")
        if self.native:                                 # the caller's own tensors have just changed in place
            note_raw_write([self.c[k] for k in self.keys if k in self.native])

    def grid(self, key):
        """Current values of one grid as a fresh float32 [1,32,D,H,W] tensor (layout conversion; for inspection)."""
        G = self.grids[key]
        if key in self.native:
            return self.c[key].detach().contiguous()
        out = torch.empty((1, 32) + G.dims, dtype=torch.float32, device=G.vm.device)
        L.check(L.lib().enslam_grid_from_voxel_major(_ptr(G.vm), _ptr(out), G.vm.shape[0], _stream()),
                "enslam_grid_from_voxel_major")
        return out

    def write_back(self):
        """val[mask] = val_grad (Mapper.py:596-602): optimised voxels into the caller's tensors, in place."""
        with torch.no_grad():
            for k in self.keys:
                if k in self.native:
                    continue                            # (optimised in place)
                new = self.grid(k)
                if self.mask5[k] is None:
                    self.c[k].copy_(new)
                else:
                    self.c[k].copy_(torch.where(self.mask5[k][None, None], new, self.c[k]))
        return self.c


class FusedAdam:
    """torch.optim.Adam (defaults) for a list of small dense parameters -- the decoders the mapper optimises
    (Mapper.py:363-369, group 0 of :409) -- as ONE launch per step; learning rate and step count live on the
    device, so a captured step follows `set_lr` without re-capture.  Parameters whose `.grad` is None are skipped
    only if that holds for all of them (the reference's decoder group always receives gradients together)."""

    def __init__(self, params, lr=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FusedAdam: empty parameter list")
        if len(self.params) > 72:
            raise L.EnslamError("FusedAdam: at most 72 tensors per optimiser")
        dev = self.params[0].device
        for p in self.params:
            _require_hip(p, "parameter")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise L.EnslamError("FusedAdam: parameters must be contiguous float32")
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        sizes = [p.numel() for p in self.params]
        self._m = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self._v = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.exp_avg, self.exp_avg_sq = list(self._m.split(sizes)), list(self._v.split(sizes))
        self.lr_t = torch.full((1,), float(lr), dtype=torch.float64, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self._lr_host = float(lr)

    def set_lr(self, lr):
        if float(lr) != self._lr_host:
            self.lr_t.fill_(float(lr))
            self._lr_host = float(lr)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def reset(self):
        """Back to the state of a freshly constructed optimiser (moments and step count zero; the learning rate stays): what
        the reference gets by building a new torch.optim.Adam per frame (Tracker.py:303-306) -- without new device buffers, so a
        captured iteration that reads them keeps working."""
        self._m.zero_()
        self._v.zero_()
        self.step_t.zero_()

    def step(self, lr=None):
        if lr is not None:
            self.set_lr(lr)
        grads = [p.grad for p in self.params]
        if all(g is None for g in grads):
            return
        n = len(self.params)
        P, G, M, V = ((ctypes.c_void_p * n)() for _ in range(4))
        numel = (ctypes.c_int64 * n)()
        for i, (p, g) in enumerate(zip(self.params, grads)):
            if g is None:
                raise L.EnslamError("FusedAdam.step: some parameters of the group have no gradient")
            g = g if g.is_contiguous() else g.contiguous()
            P[i], G[i], M[i], V[i] = p.data_ptr(), g.data_ptr(), self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr()
            numel[i] = p.numel()
        # raw kernel writes: tell torch (and the packed-decoder cache) they changed -- now, and on every replay when
        # this call is being captured (graph.GraphedStep.replay)
        note_raw_write(self.params)
        if n == 1 and numel[0] <= 1024:                 # one workgroup (a camera tensor): the launch counts the step itself
            L.check(L.lib().enslam_adam_tensors_step(n, P, G, M, V, numel, _ptr(self.lr_t), _ptr(self.step_t), self.betas[0],
                                                     self.betas[1], self.eps, _stream()), "enslam_adam_tensors_step")
            return
        self.step_t += 1
        L.check(L.lib().enslam_adam_tensors(n, P, G, M, V, numel, _ptr(self.lr_t), _ptr(self.step_t), self.betas[0],
                                            self.betas[1], self.eps, _stream()), "enslam_adam_tensors")


class MapperIteration:
    """The joint-optimisation loop of `Mapper.optimize_map` (src/Mapper.py:326-641, RGB-D part) on the device layout:
    frustum-masked grids in `MaskedGridOptimizer`, the optimised decoders and -- with `BA` -- the camera tensors of the
    optimised frames in `FusedAdam`s (the reference's one torch.optim.Adam with a parameter group each, :396-407).

      frames  list of dicts, in the reference's `optimize_frame` order (selected keyframes ..., current frame last):
              {'depth': [H,W], 'color': [H,W,3], 'c2w': [3|4,4] estimated pose, 'fixed': bool}; with BA every frame
              that is not `fixed` (the reference fixes the oldest keyframe, :376-377) gets a camera tensor
              (`camera_tensors[i]`: given, or `common.get_tensor_from_camera(c2w)`), optimised in the colour stage
              with `BA_cam_lr` (:486-490) and turned into rays by the fused pose -> ray launch of `tracker.py`
      keys    grids this mapper optimises ((`grid_coarse`,) for the coarse mapper, :326-328)

    `step(joint_iter, num_joint_iters)` is one iteration (:448-602): stage and learning rates of the schedule, one
    `get_samples` draw of `pixels // len(frames)` pixels per frame (:502-535), the in-bound prefilter (:537-547), render,
    loss, backward, optimiser steps.  `static_shapes=True` keeps the ray count fixed (no boolean indexing, no host
    synchronisation): rays the reference drops are rendered but carry no loss, the sampler's batch maxima are taken over
    the kept rays -- the kept rays get exactly the samples, losses and gradients of the reference's filtered batch, and
    the whole iteration, per-iteration pixel draw included, can be captured in one hipGraph (`graphed()`).
    `finish()` writes the grids back (`val[mask] = val_grad`, :596-602) and returns the camera tensors."""

    def __init__(self, cfg, renderer, c, decoders, frames, cam, masks=None, keys=('grid_middle', 'grid_fine', 'grid_color'),
                 camera_tensors=None, device=None, static_shapes=False):
        from .common import get_tensor_from_camera
        m = cfg['mapping']
        self.cfg, self.renderer, self.c, self.decoders = cfg, renderer, c, decoders
        self.keys = tuple(keys)
        self.coarse_mapper = self.keys == ('grid_coarse',)
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = (cam[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
        self.device = device if device is not None else c[self.keys[0]].device
        self.w_color_loss, self.lr_factor = m['w_color_loss'], m['lr_factor']
        self.BA, self.BA_cam_lr = bool(m.get('BA', False)) and not self.coarse_mapper, m.get('BA_cam_lr', 0.0)
        self.middle_iter_ratio, self.fine_iter_ratio = m['middle_iter_ratio'], m['fine_iter_ratio']
        self.fix_fine, self.fix_color = m.get('fix_fine', True), m.get('fix_color', False)
        self.pixels = m['pixels']
        self.static_shapes = static_shapes
        self.frames = frames
        self.grid_opt = MaskedGridOptimizer(c, masks, keys=self.keys)
        dec_params = []                                                     # :363-369
        if not self.fix_fine:
            dec_params += list(decoders.fine_decoder.parameters())
        if not self.fix_color:
            dec_params += list(decoders.color_decoder.parameters())
        self.dec_opt = FusedAdam(dec_params, lr=0.0) if dec_params else None
        self.camera_tensors = [None] * len(frames)
        if self.BA:                                                         # :374-390
            for i, f in enumerate(frames):
                if f.get('fixed', False):
                    continue
                t = camera_tensors[i] if camera_tensors is not None and camera_tensors[i] is not None else \
                    get_tensor_from_camera(f['c2w'])
                self.camera_tensors[i] = t.detach().to(self.device, torch.float32).clone().requires_grad_(True)
        cams = [t for t in self.camera_tensors if t is not None]
        self.cam_opt = FusedAdam(cams, lr=0.0) if cams else None
        self._bound_dev = renderer.bound.to(self.device)
        self._one = None
        self.last = {}

    def stage_of(self, joint_iter, num_joint_iters):                       # :460-467
        if self.coarse_mapper:
            return 'coarse'
        if joint_iter <= int(num_joint_iters * self.middle_iter_ratio):
            return 'middle'
        if joint_iter <= int(num_joint_iters * self.fine_iter_ratio):
            return 'fine'
        return 'color'

    def sample_batch(self):
        """One `get_samples` draw per frame (:502-535) -> concatenated rays_o, rays_d, gt_depth, gt_color (float32)."""
        from .common import get_samples
        from .tracker import get_samples_from_camera_tensor
        H, W = self.H, self.W
        n = self.pixels // len(self.frames)
        ro, rd, gd, gc = [], [], [], []
        for f, ct in zip(self.frames, self.camera_tensors):
            if ct is not None:
                o, d, dep, col = get_samples_from_camera_tensor(0, H, 0, W, n, H, W, self.fx, self.fy, self.cx, self.cy, ct,
                                                                f['depth'], f['color'], self.device)
            else:
                o, d, dep, col = get_samples(0, H, 0, W, n, H, W, self.fx, self.fy, self.cx, self.cy, f['c2w'], f['depth'],
                                             f['color'], self.device)
            ro.append(o.float()); rd.append(d.float()); gd.append(dep.float()); gc.append(col.float())
        return torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)

    def step(self, joint_iter, num_joint_iters, stage=None):
        """One joint iteration; returns the (device) loss tensor.  `stage` overrides the schedule (captured steps)."""
        from .losses import rgbd_loss
        stage = stage or self.stage_of(joint_iter, num_joint_iters)
        st = self.cfg['mapping']['stage'][stage]
        f = self.lr_factor
        if self.dec_opt is not None:                                        # :469-490
            self.dec_opt.set_lr(st['decoders_lr'] * f)
            self.dec_opt.zero_grad()
        if self.cam_opt is not None:
            if stage == 'color':
                self.cam_opt.set_lr(self.BA_cam_lr)
            self.cam_opt.zero_grad()
        ro, rd, gd, gc = self.sample_batch()
        keep = None
        with torch.no_grad():                                               # :537-547
            t = (self._bound_dev.unsqueeze(0) - ro.detach().unsqueeze(-1)) / rd.detach().unsqueeze(-1)
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            inside = t >= gd
        r = self.renderer
        prev = r.depth_max_override
        if self.static_shapes:
            keep = inside
            if not self.coarse_mapper:
                mx = torch.where(inside, gd, gd.new_zeros(())).max().reshape(1)
                r.depth_max_override = torch.cat([mx, mx * 1.2]).contiguous()
        else:
            rd, ro, gd, gc = rd[inside], ro[inside], gd[inside], gc[inside]
        try:
            depth, unc, color = r.render_batch_ray(self.grid_opt.render_grids(), self.decoders, rd, ro, self.device, stage,
                                                   gt_depth=None if self.coarse_mapper else gd)
        finally:
            r.depth_max_override = prev
        use_color = stage == 'color'
        if keep is None:
            loss = rgbd_loss(depth, color if use_color else None, gd, gc, self.w_color_loss)          # :553-562
        else:
            # dropped rays: no depth term (gt_depth 0) and no colour term (target = the rendered colour itself: |0|, sign 0)
            gd_l = torch.where(keep, gd, torch.zeros_like(gd))
            gc_l = torch.where(keep[:, None], gc, color.detach()) if use_color else gc
            loss = rgbd_loss(depth, color if use_color else None, gd_l, gc_l, self.w_color_loss)
        if self._one is None:
            self._one = torch.ones_like(loss)
        with engine_on_calling_thread():
            loss.backward(gradient=self._one)                               # :573
        if self.dec_opt is not None:
            self.dec_opt.step()                                             # :575 (one optimizer.step() in the reference)
        if self.cam_opt is not None:
            self.cam_opt.step()
        self.grid_opt.step({k: st[k[5:] + '_lr'] * f for k in self.keys})
        self.last = dict(stage=stage, inside=inside, n_rays=ro.shape[0])
        return loss.detach()

    def graphed(self, stage, warmup=3):
        """The iteration of one stage as ONE hipGraph (static shapes): `replay()` runs an iteration with fresh pixels."""
        from .graph import GraphedStep
        if not self.static_shapes:
            raise L.EnslamError("MapperIteration.graphed needs static_shapes=True")
        st = self.cfg['mapping']['stage'][stage]
        # learning rates are device scalars: set them before capture, they stay adjustable between replays
        self.grid_opt.set_lr({k: st[k[5:] + '_lr'] * self.lr_factor for k in self.keys})
        return GraphedStep(lambda: self.step(0, 1, stage=stage), warmup=warmup)

    def finish(self):
        """Write the optimised voxels back into the caller's grids; returns the camera tensors (None for fixed frames)."""
        self.grid_opt.write_back()
        return [None if t is None else t.detach() for t in self.camera_tensors]
