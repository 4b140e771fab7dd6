"""`Renderer` with the reference's constructor and method signatures (src/utils/Renderer.py:6-360),
backed by the HIP library.  Callers (Tracker.py:150,175; Mapper.py:548,567,591; Mesher.py:548;
Visualizer.py:79,244) use it unchanged."""
import torch
import torch.nn.functional as F

from . import _lib as L
from . import functional as EF
from .common import get_rays, get_rays_rescale


class Renderer(object):
    FUSED_LOSS_MAX_RAYS = 32768         # the tile-mode limit of enslam_render_fwd (csrc/render_fwd.hip)

    def __init__(self, cfg, args, slam, points_batch_size=500000, ray_batch_size=100000):
        self.ray_batch_size = ray_batch_size
        self.points_batch_size = points_batch_size
        r = cfg['rendering']
        self.lindisp, self.perturb = r['lindisp'], r['perturb']
        self.N_samples, self.N_surface, self.N_importance = r['N_samples'], r['N_surface'], r['N_importance']
        self.scale = cfg['scale']
        self.occupancy = cfg['occupancy']
        self.nice = slam.nice
        self.bound = slam.bound
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = slam.H, slam.W, slam.fx, slam.fy, slam.cx, slam.cy
        if not self.nice or not self.occupancy:
            raise NotImplementedError("the HIP path implements the NICE / occupancy configuration "
                                      "(configs/nice_slam.yaml: occupancy True); iMAP mode is out of scope")
        self._tvals = {}
        # (PyTorch's autograd threading is left alone here: the loops of this package scope the single-thread engine around
        # their own backward calls -- functional.engine_on_calling_thread -- and a caller can do the same.)
        # ray-sharded callers (parallel.ShardedRenderer) set this to functional.batch_depth_max(whole batch)
        self.depth_max_override = None
        # what render calls leave behind for this renderer's caller (touched-block flags, work-list size, profiling hooks)
        self.state = EF.RenderState()

    # ------------------------------------------------------------------ helpers
    def _t_vals(self, device, n_lin, n_surf):
        key = (str(device), n_lin, n_surf)
        if key not in self._tvals:
            t_lin = torch.linspace(0., 1., steps=n_lin, device=device)
            t_surf = torch.linspace(0., 1., steps=n_surf, device=device).double() if n_surf > 0 else None
            self._tvals[key] = (t_lin, t_surf)
        return self._tvals[key]

    def _coarse_bound(self, decoders):
        cd = getattr(decoders, 'coarse_decoder', None)
        b = getattr(cd, 'bound', None) if cd is not None else None
        return b if b is not None else self.bound

    # ------------------------------------------------------------------ API
    def eval_points(self, p, decoders, c=None, stage='color', device='cuda:0'):
        """occupancy (and colour) of points p [N,3]; out-of-bound points get occ = 100."""
        rets = []
        for pi in torch.split(p, self.points_batch_size):
            rets.append(EF.eval_points(pi, decoders, c, stage, self.bound, apply_mask=True,
                                       coarse_bound=self._coarse_bound(decoders)))
        return torch.cat(rets, dim=0)

    def render_batch_ray(self, c, decoders, rays_d, rays_o, device, stage, gt_depth=None):
        """(depth f64 [N], uncertainty f64 [N], color f32 [N,3]) -- note rays_d comes before rays_o."""
        return self._render(c, decoders, rays_d, rays_o, device, stage, gt_depth, None)

    def render_batch_ray_rgbd_loss(self, c, decoders, rays_d, rays_o, device, stage, gt_depth, gt_color, w_color=0.2):
        """`render_batch_ray` and the mapper's loss on its outputs (Mapper.py:548-562:
        sum_{gt_depth>0} |gt_depth - depth| + [stage == 'color'] w_color * sum |gt_color - color|) with the loss folded
        into the compositing launches, forward and backward.  Returns (loss f64 scalar, depth, uncertainty, color); only
        the loss carries gradient.  Same numbers as `losses.rgbd_loss(*render_batch_ray(...))` up to the summation
        order of the loss value."""
        if gt_depth is None or stage == 'coarse':
            raise ValueError("render_batch_ray_rgbd_loss needs gt_depth and a depth-guided stage (middle, fine, color)")
        gd = gt_depth.detach().contiguous().float().reshape(-1)
        gc = gt_color.detach().contiguous().float().reshape(-1, 3) if (stage == 'color' and gt_color is not None) else None
        if rays_o.shape[0] > self.FUSED_LOSS_MAX_RAYS:
            # full-image batches run the one-wave-per-ray forward with compositing inside the kernel: separate loss launches
            from .losses import rgbd_loss
            depth, var, color = self._render(c, decoders, rays_d, rays_o, device, stage, gt_depth, None)
            loss = rgbd_loss(depth, color if gc is not None else None, gd, gc, w_color)
            return loss, depth.detach(), var.detach(), color.detach()
        return self._render(c, decoders, rays_d, rays_o, device, stage, gt_depth, (gd, gc, float(w_color)))

    def tracker_loss_ok(self, n_rays, gt_depth):
        """whether render_batch_ray_tracker_loss serves this batch (else: render_batch_ray + losses.tracker_loss)"""
        S = self.N_samples + self.N_surface
        return (gt_depth is not None and 0 < n_rays <= min(self.FUSED_LOSS_MAX_RAYS, L.lib().enslam_tracker_tail_max_rays())
                and self.N_importance == 0 and S % 16 == 0 and S <= 64)

    def render_batch_ray_tracker_loss(self, c, decoders, rays_d, rays_o, device, stage, gt_depth, gt_color, w_color=0.5, inside=None,
                                      handle_dynamic=True, use_color=True):
        """`render_batch_ray` and the TRACKER's RGB-D loss on its outputs (Tracker.py:176-195) with the loss, its median mask and
        its gradient folded into one launch behind the decoders:
            tmp  = |gt_depth - depth| / sqrt(uncertainty.detach() + 1e-10)
            keep = inside & (tmp < 10 * median(tmp[inside]))     (handle_dynamic; `inside`: uint8 / bool mask of the rays the
                                                                  in-bound prefilter of :164-174 keeps, None = every ray)
            loss = tmp[keep & (gt_depth > 0)].sum() + w_color * |gt_color - color|[keep & (gt_depth > 0)].sum()   (use_color)
        Returns (loss f64 scalar, depth, uncertainty, color); only the loss carries gradient.  Batches of up to 4096 rays, colour
        / fine / middle stage (`tracker_loss_ok`)."""
        if stage == 'coarse' or not self.tracker_loss_ok(rays_o.shape[0], gt_depth):
            raise ValueError("render_batch_ray_tracker_loss: batch not served (see tracker_loss_ok); use render_batch_ray + losses.tracker_loss")
        gd = gt_depth.detach().contiguous().float().reshape(-1)
        gc = gt_color.detach().contiguous().float().reshape(-1, 3) if (use_color and stage == 'color' and gt_color is not None) else None
        ins = None
        if inside is not None:
            ins = inside.detach().reshape(-1)
            ins = (ins if ins.dtype == torch.uint8 else ins.to(torch.uint8)).contiguous()
        return self._render(c, decoders, rays_d, rays_o, device, stage, gt_depth, (gd, gc, float(w_color), ins, bool(handle_dynamic), 'tracker'))

    def _render(self, c, decoders, rays_d, rays_o, device, stage, gt_depth, loss, z_given=None, s_valid=None):
        if self.N_importance > 0 and z_given is None:
            if loss is not None:
                raise NotImplementedError("render_batch_ray_rgbd_loss with N_importance > 0: use render_batch_ray + losses.rgbd_loss")
            return self._render_hierarchical(c, decoders, rays_d, rays_o, device, stage, gt_depth)
        if stage not in L.STAGE:
            raise ValueError(f"unknown stage {stage!r}")
        EF._require_hip(rays_o, "rays")
        if stage == 'coarse':
            gt_depth = None
        N = rays_o.shape[0]
        if gt_depth is not None:
            gt_depth = gt_depth.reshape(-1)
            if N == 0:
                raise RuntimeError("render_batch_ray: empty ray batch with gt_depth (the reference's "
                                   "torch.max over an empty tensor raises here too)")
        n_lin, n_surf = self.N_samples, (self.N_surface if gt_depth is not None else 0)
        S = n_lin + n_surf
        if z_given is None and S > 64:
            raise NotImplementedError(f"N_samples + N_surface = {S}: the kernels composite at most 64 samples per ray")
        dev = rays_o.device
        if z_given is None and S % 16 != 0 and N > 0:
            # sample counts that are not whole 16-sample tiles: the sampler runs on its own and the render takes the distances
            # as given, the last tile padded (the pad is evaluated but neither composited nor given gradient)
            if loss is not None:
                raise NotImplementedError("render_batch_ray_rgbd_loss needs N_samples + N_surface to be a multiple of 16")
            with torch.no_grad():
                t_r = torch.rand((N, n_lin), device=dev) if self.perturb > 0. else None
                z = EF.sample_rays(rays_o, rays_d, gt_depth, self.bound, n_lin, n_surf, self.lindisp, t_r,
                                   depth_max=self.depth_max_override if gt_depth is not None else None)
                S_pad = -(-S // 16) * 16
                z = torch.cat([z, z[:, -1:].expand(N, S_pad - S)], -1).contiguous()
            return self._render(c, decoders, rays_d, rays_o, device, stage, gt_depth, None, z_given=z, s_valid=S)
        t_lin, t_surf = self._t_vals(dev, n_lin, self.N_surface)
        t_rand = torch.rand((N, n_lin), device=dev) if (self.perturb > 0. and z_given is None) else None
        kinds = EF.stage_kinds(stage)
        decs = {k: getattr(decoders, L.MLP_NAMES[k]) for k in kinds}
        plan = EF.RenderPlan(stage, self.bound, self._coarse_bound(decoders), n_lin, n_surf, self.lindisp, t_lin,
                             t_surf, kinds, decs, depth_max=self.depth_max_override if gt_depth is not None else None)
        plan.loss = loss
        plan.state = self.state
        plan.z_given, plan.s_valid = z_given, s_valid
        grids = []
        for k in kinds:
            g = c[L.GRID_NAMES[k]]
            if isinstance(g, EF.VoxelMajorGrid):            # grid held in the device layout (mapper.MaskedGridOptimizer)
                plan.vm[k] = g
                g = g.anchor
            grids.append(g)
        params = []
        for k in kinds:
            params += EF.decoder_params(decs[k], k)
        if N == 0:
            z = rays_o.new_zeros((0,))
            return z.double(), z.double(), rays_o.new_zeros((0, 3))
        return EF.render(plan, rays_o, rays_d, gt_depth, t_rand, grids, params)

    HIERARCHICAL_MAX_RAYS = 32768        # 64-sample rays run on the tile-per-wave forward (its ray limit)

    def _render_hierarchical(self, c, decoders, rays_d, rays_o, device, stage, gt_depth):
        """`render_batch_ray` with N_importance > 0 (Renderer.py:182-197): a first pass over the N_samples + N_surface
        distances gives the weights, `sample_pdf` (common.py:19-63, torch ops, deterministic when perturb == 0) draws
        N_importance more distances from them, and the sorted union is rendered.  The reference detaches the new distances
        and returns only the second pass's outputs, so no gradient flows through the first pass: it runs forward-only
        (sampler, `eval_points`, compositing kernels); the second pass is the differentiable HIP render on GIVEN distances,
        its last 16-sample tile padded (the pad is evaluated but neither composited nor given gradient)."""
        from .common import sample_pdf
        if stage == 'coarse':
            gt_depth = None
        N = rays_o.shape[0]
        if N == 0:
            z = rays_o.new_zeros((0,))
            return z.double(), z.double(), rays_o.new_zeros((0, 3))
        if N > self.HIERARCHICAL_MAX_RAYS:
            # chunked internally; the sampler's batch maxima (Renderer.py:110,145) span the whole call, so they are taken here
            # once and handed to every chunk (render_img / render_img_rescale pass up to ray_batch_size = 100000 rays)
            prev = self.depth_max_override
            if gt_depth is not None and prev is None:
                self.depth_max_override = EF.batch_depth_max(gt_depth.reshape(-1))
            try:
                outs = [self._render_hierarchical(c, decoders, rays_d[i:i + self.HIERARCHICAL_MAX_RAYS],
                                                  rays_o[i:i + self.HIERARCHICAL_MAX_RAYS], device, stage,
                                                  None if gt_depth is None else gt_depth.reshape(-1)[i:i + self.HIERARCHICAL_MAX_RAYS])
                        for i in range(0, N, self.HIERARCHICAL_MAX_RAYS)]
            finally:
                self.depth_max_override = prev
            return tuple(torch.cat(t, 0) for t in zip(*outs))
        dev = rays_o.device
        n_surf = self.N_surface if gt_depth is not None else 0
        gd = gt_depth.reshape(-1) if gt_depth is not None else None
        with torch.no_grad():
            t_rand = torch.rand((N, self.N_samples), device=dev) if self.perturb > 0. else None
            z1 = EF.sample_rays(rays_o, rays_d, gd, self.bound, self.N_samples, n_surf, self.lindisp, t_rand,
                                depth_max=self.depth_max_override if gd is not None else None)
            pts, _ = EF.ray_points(rays_o.detach(), rays_d.detach(), z1, self.bound)
            raw1 = EF.eval_points(pts, decoders, c, stage, self.bound, apply_mask=True, coarse_bound=self._coarse_bound(decoders))
            _, _, _, weights = EF.composite(raw1.view(N, z1.shape[1], 4), z1)
            z_mid = .5 * (z1[..., 1:] + z1[..., :-1])
            z_samples = sample_pdf(z_mid, weights[..., 1:-1], self.N_importance, det=(self.perturb == 0.), device=dev)
            z2, _ = torch.sort(torch.cat([z1, z_samples.to(z1.dtype)], -1), -1)
            S2 = z2.shape[1]
            S_pad = -(-S2 // 16) * 16
            if S_pad > 64:
                raise NotImplementedError(f"N_samples + N_importance + N_surface = {S2}: at most 64 samples per ray")
            if S_pad > S2:
                z2 = torch.cat([z2, z2[:, -1:].expand(N, S_pad - S2)], -1)
            z2 = z2.contiguous()
        return self._render(c, decoders, rays_d, rays_o, device, stage, gt_depth, None, z_given=z2, s_valid=S2)

    def _render_chunks(self, c, decoders, rays_o, rays_d, device, stage, gt_depth):
        depth_l, var_l, col_l = [], [], []
        for i in range(0, rays_d.shape[0], self.ray_batch_size):
            gd = None if gt_depth is None else gt_depth[i:i + self.ray_batch_size]
            d, u, col = self.render_batch_ray(c, decoders, rays_d[i:i + self.ray_batch_size],
                                              rays_o[i:i + self.ray_batch_size], device, stage, gt_depth=gd)
            depth_l.append(d.double())
            var_l.append(u.double())
            col_l.append(col)
        return torch.cat(depth_l, 0), torch.cat(var_l, 0), torch.cat(col_l, 0)

    def render_img(self, c, decoders, c2w, device, stage, gt_depth=None):
        """Full-resolution depth / uncertainty / colour images, no gradient (Renderer.py:201-256)."""
        with torch.no_grad():
            H, W = self.H, self.W
            rays_o, rays_d = get_rays(H, W, self.fx, self.fy, self.cx, self.cy, c2w, device)
            gd = gt_depth.reshape(-1) if gt_depth is not None else None
            depth, var, color = self._render_chunks(c, decoders, rays_o.reshape(-1, 3), rays_d.reshape(-1, 3),
                                                    device, stage, gd)
            return depth.reshape(H, W), var.reshape(H, W), color.reshape(H, W, 3)

    def render_img_rescale(self, c, decoders, c2w, device, stage, gt_depth=None, scale_factor=0.1):
        """Image rendered at (int(H*s), int(W*s)) pixel centres WITH gradient (Renderer.py:258-319)."""
        H, W = self.H, self.W
        new_H, new_W = int(H * scale_factor), int(W * scale_factor)
        rays_o, rays_d = get_rays_rescale(H, W, new_H, new_W, self.fx, self.fy, self.cx, self.cy, c2w, device)
        gd = None
        if gt_depth is not None:
            # torchvision Resize(BILINEAR) on a tensor == F.interpolate(bilinear, align_corners=False)
            # (antialias only applies when asked for on tensors in the pinned torchvision).
            gd = F.interpolate(gt_depth[None, None].float(), size=(new_H, new_W), mode='bilinear',
                               align_corners=False).reshape(-1)
        depth, var, color = self._render_chunks(c, decoders, rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), device,
                                                stage, gd)
        return depth.reshape(new_H, new_W), var.reshape(new_H, new_W), color.reshape(new_H, new_W, 3)

    def regulation(self, c, decoders, rays_d, rays_o, gt_depth, device, stage='color'):
        raise NotImplementedError("Renderer.regulation is the iMAP (occupancy=False) free-space regulariser "
                                  "(Renderer.py:322-360); not part of the NICE hot path")
