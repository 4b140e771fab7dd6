"""CPU oracle for the volume-rendering hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch CPU ops, float64 where the reference is
float64) of the algorithm the HIP kernels implement.  It is imported only by
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg, and only as
the checker / the timed CPU baseline.  The product path (evennicer_slam_amd)
never imports it and never falls back to it.

Parity pin: tests/test_oracle_golden.py checks every function here against the
fixtures in tests/golden/*.npz, which were produced by running the reference's
own Python (tests/golden/make_golden.py) in the build container.

Reference functions restated (file:line in /root/reference):
  scene_bound            <- src/EvenNICER_SLAM.py:170-175   (load_bound rounding)
  grid_shapes            <- src/EvenNICER_SLAM.py:236-273   (grid_init shapes)
  pixel_rays / sample_pixels / image_rays
                         <- src/common.py:74-89,92-107,130-142,160-169,300-340
  sample_depths          <- src/utils/Renderer.py:83-171     (z sampling + sort)
  inside_bound           <- src/utils/Renderer.py:44-47
  trilinear              <- src/common.py:342-357 + src/conv_onet/models/decoder.py:168-175
                            (F.grid_sample; explicit form follows ATen GridSampler.h)
  fourier_embed          <- src/conv_onet/models/decoder.py:26-30
  decode_xyz / decode_feat / decode_stage
                         <- src/conv_onet/models/decoder.py:177-203,262-274,312-342
  eval_points            <- src/utils/Renderer.py:24-62
  composite              <- src/common.py:256-297 (occupancy branch)
  render_batch_ray       <- src/utils/Renderer.py:64-199 (incl. the N_importance second pass, :182-197)
  inverse_cdf_samples    <- src/common.py:19-63 (sample_pdf)
"""
import numpy as np
import torch
import torch.nn.functional as F

STAGE_GRIDS = {
    'coarse': ('grid_coarse',),
    'middle': ('grid_middle',),
    'fine': ('grid_fine', 'grid_middle'),
    'color': ('grid_fine', 'grid_middle', 'grid_color'),
}
STAGE_DECODERS = {
    'coarse': ('coarse_decoder',),
    'middle': ('middle_decoder',),
    'fine': ('fine_decoder', 'middle_decoder'),
    'color': ('fine_decoder', 'color_decoder', 'middle_decoder'),
}


# ----------------------------------------------------------------------------- scene geometry
def scene_bound(cfg_bound, scale=1.0, bound_divisible=0.32):
    """Rounded scene bound, float64 [3,2].  The upper edge is
    lo + float32(k * divisible): the reference multiplies an int32 tensor by a
    python float, which yields float32, before adding the float64 lower edge."""
    b = torch.as_tensor(np.asarray(cfg_bound, dtype=np.float64) * scale).clone()
    k = torch.trunc((b[:, 1] - b[:, 0]) / bound_divisible).to(torch.int32) + 1
    step32 = k.to(torch.float32) * torch.tensor(bound_divisible, dtype=torch.float32)
    b[:, 1] = step32.double() + b[:, 0]
    return b


def grid_shapes(bound, grid_len, coarse_enlarge=2):
    """{'grid_*': [D,H,W]} (z,y,x order) from bound and grid_len dict coarse/middle/fine/color."""
    ext = (bound[:, 1] - bound[:, 0])
    out = {}
    for key in ('coarse', 'middle', 'fine', 'color'):
        e = ext * coarse_enlarge if key == 'coarse' else ext
        nx, ny, nz = [int(v) for v in (e / grid_len[key]).tolist()]
        out['grid_' + key] = [nz, ny, nx]
    return out


# ----------------------------------------------------------------------------- rays
def pixel_rays(i, j, c2w, fx, fy, cx, cy):
    """Rays through pixel centres (i = column, j = row).  rays_d = R @ dir, rays_o = t."""
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    rays_d = (dirs.reshape(-1, 1, 3) * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def sample_pixels(H0, H1, W0, W1, n, c2w, depth, color, fx, fy, cx, cy, idx=None):
    """n random pixels of the window; `idx` may be given to bypass the RNG.
    The RNG draw is torch.randint(window_size, (n,)) on the default generator."""
    ww, hh = W1 - W0, H1 - H0
    if idx is None:
        idx = torch.randint(hh * ww, (n,))
    col = torch.linspace(W0, W1 - 1, ww)[idx % ww]
    row = torch.linspace(H0, H1 - 1, hh)[idx // ww]
    d = depth[H0:H1, W0:W1].reshape(-1)[idx]
    c = color[H0:H1, W0:W1].reshape(-1, 3)[idx]
    ro, rd = pixel_rays(col, row, c2w, fx, fy, cx, cy)
    return ro, rd, d, c


def image_rays(H, W, new_H, new_W, c2w, fx, fy, cx, cy):
    """All rays of an image resampled to (new_H,new_W) pixel centres (strided, not averaged)."""
    col = torch.linspace(0, W - 1, new_W)
    row = torch.linspace(0, H - 1, new_H)
    jj, ii = torch.meshgrid(row, col, indexing='ij')
    ro, rd = pixel_rays(ii.reshape(-1), jj.reshape(-1), c2w, fx, fy, cx, cy)
    return ro.reshape(new_H, new_W, 3), rd.reshape(new_H, new_W, 3)


# ----------------------------------------------------------------------------- sampling
def sample_depths(rays_o, rays_d, gt_depth, bound, n_samples, n_surface, stage,
                  lindisp=False, t_rand=None, depth_max=None):
    """Sorted sample distances z [N, S] float64.  S = n_samples (+ n_surface if depth-guided).
    depth_max: optional 0-dim float32 tensor replacing max(gt_depth) (ray-sharded callers pass the batch's)."""
    if stage == 'coarse':
        gt_depth = None
    with torch.no_grad():
        o = rays_o.detach().unsqueeze(-1)
        d = rays_d.detach().unsqueeze(-1)
        t = (bound.unsqueeze(0) - o) / d                       # float64 [N,3,2]
        far_bb = t.max(dim=2)[0].min(dim=1)[0].unsqueeze(-1) + 0.01
    t_lin = torch.linspace(0., 1., steps=n_samples)
    if gt_depth is None:
        near = 0.01
        far = far_bb
    else:
        gd = gt_depth.reshape(-1, 1)
        near = gd.repeat(1, n_samples) * 0.01                  # float32
        dmax = torch.max(gd) if depth_max is None else depth_max.to(gd.dtype)
        far = torch.clamp(far_bb, 0, dmax * 1.2)               # batch-global max (max(gd*1.2) == max(gd)*1.2 in float32)
    if lindisp:
        z = 1. / (1. / near * (1. - t_lin) + 1. / far * t_lin)
    else:
        z = near * (1. - t_lin) + far * t_lin
    if t_rand is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    if gt_depth is not None and n_surface > 0:
        t_s = torch.linspace(0., 1., steps=n_surface).double()
        hit = (gd > 0).squeeze(-1)
        zs = torch.zeros(gd.shape[0], n_surface, dtype=torch.float64)
        dh = gd[hit].reshape(-1, 1).repeat(1, n_surface)
        zs[hit] = 0.95 * dh * (1. - t_s) + 1.05 * dh * t_s
        zs[~hit] = 0.001 * (1. - t_s) + dmax * t_s
        z = torch.sort(torch.cat([z, zs], -1), -1)[0]
    return z


def inside_bound(p, bound):
    m = torch.ones(p.shape[0], dtype=torch.bool)
    for a in range(3):
        m &= (p[:, a] > bound[a, 0]) & (p[:, a] < bound[a, 1])
    return m


# ----------------------------------------------------------------------------- trilinear gather
def normalized(p, bound):
    q = torch.empty_like(p)
    for a in range(3):
        q[:, a] = ((p[:, a] - bound[a, 0]) / (bound[a, 1] - bound[a, 0])) * 2 - 1.0
    return q


def voxel_coords(p, bound, shape):
    """(ix0,iy0,iz0 int64, fx,fy,fz float32, inner flags) following ATen:
    unnormalize ((c+1)/2)*(size-1) in float32, clip to [0,size-1], floor."""
    pn = normalized(p, bound).float()
    D, H, W = shape
    idx, frac, inner = [], [], []
    for a, size in ((0, W), (1, H), (2, D)):
        co = ((pn[:, a] + 1) / 2) * (size - 1)
        inner.append((co > 0) & (co < size - 1))
        co = torch.clamp(torch.clamp(co, min=0.0), max=float(size - 1))
        fl = torch.floor(co)
        idx.append(fl.long())
        frac.append(co - fl)
    return idx, frac, inner


def trilinear_explicit(grid, p, bound):
    """Explicit 8-corner form of F.grid_sample(bilinear, border, align_corners=True).
    Corners beyond size-1 contribute 0 (their weight is 0 as well).  [P, C] float32."""
    C, D, H, W = grid.shape[1:]
    (ix, iy, iz), (fx, fy, fz), _ = voxel_coords(p, bound, (D, H, W))
    g = grid[0].reshape(C, -1)
    out = torch.zeros(p.shape[0], C, dtype=grid.dtype)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                w = (fx if dx else 1 - fx) * (fy if dy else 1 - fy) * (fz if dz else 1 - fz)
                x, y, z = ix + dx, iy + dy, iz + dz
                ok = (x < W) & (y < H) & (z < D)
                lin = (z.clamp(max=D - 1) * H + y.clamp(max=H - 1)) * W + x.clamp(max=W - 1)
                out = out + (g[:, lin].t() * (w * ok).unsqueeze(-1))
    return out


def trilinear(grid, p, bound):
    """[P, C] features; the port of the reference call (normalise in float64, cast, grid_sample)."""
    vg = normalized(p, bound).float().reshape(1, -1, 1, 1, 3)
    c = F.grid_sample(grid, vg, padding_mode='border', align_corners=True, mode='bilinear')
    return c.reshape(grid.shape[1], -1).t()


# ----------------------------------------------------------------------------- decoders
def fourier_embed(p32, B):
    return torch.sin(p32 @ B)


def _lin(params, prefix, x):
    return F.linear(x, params[prefix + '.weight'], params[prefix + '.bias'])


def decode_xyz(params, name, p, c, n_blocks=5, skip=2):
    """middle / fine / color decoder: relu(W h + b) + (Wc c + bc), skip-concat after block `skip`."""
    emb = fourier_embed(p.float(), params[f'{name}.embedder._B'])
    h = emb
    for i in range(n_blocks):
        h = torch.relu(_lin(params, f'{name}.pts_linears.{i}', h)) + _lin(params, f'{name}.fc_c.{i}', c)
        if i == skip:
            h = torch.cat([emb, h], -1)
    return _lin(params, f'{name}.output_linear', h)


def decode_feat(params, name, c, n_blocks=5, skip=2):
    """coarse decoder: features only, no xyz."""
    h = c
    for i in range(n_blocks):
        h = torch.relu(_lin(params, f'{name}.pts_linears.{i}', h))
        if i == skip:
            h = torch.cat([c, h], -1)
    return _lin(params, f'{name}.output_linear', h)


def decode_stage(params, grids, p, stage, bound, coarse_enlarge=2):
    """raw [P,4] = [r,g,b,occ] for the stage."""
    P = p.shape[0]
    raw = torch.zeros(P, 4)
    if stage == 'coarse':
        c = trilinear(grids['grid_coarse'], p, bound * coarse_enlarge)
        occ = decode_feat(params, 'coarse_decoder', c).squeeze(-1)
        return torch.cat([raw[:, :3], occ.unsqueeze(-1)], -1)
    c_mid = trilinear(grids['grid_middle'], p, bound)
    mid = decode_xyz(params, 'middle_decoder', p, c_mid).squeeze(-1)
    if stage == 'middle':
        return torch.cat([raw[:, :3], mid.unsqueeze(-1)], -1)
    # the middle-feature concat of the fine decoder is evaluated without gradient
    # (neither to grid_middle nor to the sample position)
    c_fine = torch.cat([trilinear(grids['grid_fine'], p, bound), c_mid.detach()], -1)
    fine = decode_xyz(params, 'fine_decoder', p, c_fine).squeeze(-1)
    occ = fine + mid
    if stage == 'fine':
        return torch.cat([raw[:, :3], occ.unsqueeze(-1)], -1)
    c_col = trilinear(grids['grid_color'], p, bound)
    rgbx = decode_xyz(params, 'color_decoder', p, c_col)
    return torch.cat([rgbx[:, :3], occ.unsqueeze(-1)], -1)


def eval_points(params, grids, p, stage, bound, coarse_enlarge=2):
    raw = decode_stage(params, grids, p, stage, bound, coarse_enlarge)
    keep = inside_bound(p, bound)
    occ = torch.where(keep, raw[:, 3], torch.full_like(raw[:, 3], 100.0))
    return torch.cat([raw[:, :3], occ.unsqueeze(-1)], -1)


# ----------------------------------------------------------------------------- compositing
def composite(raw, z):
    """raw [N,S,4] float32, z [N,S] float64 -> depth f64, var f64, rgb f32, weights f32."""
    alpha = torch.sigmoid(10 * raw[..., 3])
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), (1. - alpha + 1e-10)], -1), -1)[:, :-1]
    w = alpha * trans
    rgb = (w.unsqueeze(-1) * raw[..., :3]).sum(-2)
    depth = (w * z).sum(-1)
    dz = z - depth.unsqueeze(-1)
    var = (w * dz * dz).sum(1)
    return depth, var, rgb, w


def inverse_cdf_samples(bins, weights, n, det=True):
    """src/common.py:19-63 (sample_pdf): n samples per ray from the piecewise-constant density `weights` [B, M-1] over the
    bin edges `bins` [B, M] by inverting its CDF; det: evenly spaced quantiles, else one torch.rand draw (CPU generator)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    shape = list(cdf.shape[:-1]) + [n]
    u = torch.linspace(0., 1., steps=n).expand(shape) if det else torch.rand(shape)
    u = u.contiguous()
    above = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(above - 1, min=0)
    above = torch.clamp(above, max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b_lo, b_hi = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return b_lo + ((u - c_lo) / span) * (b_hi - b_lo)


def render_batch_ray(params, grids, rays_d, rays_o, stage, bound, gt_depth=None,
                     n_samples=32, n_surface=16, coarse_enlarge=2, lindisp=False, t_rand=None,
                     return_aux=False, n_importance=0, det=True):
    """(depth f64 [N], var f64 [N], rgb f32 [N,3]).  Argument order rays_d, rays_o as the reference.
    n_importance > 0: Renderer.py:182-197 -- inverse-CDF samples of the first pass's weights (detached), merged with the
    first pass's distances and rendered again; only that second pass is returned."""
    z = sample_depths(rays_o, rays_d, gt_depth, bound, n_samples, n_surface, stage, lindisp, t_rand)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]).reshape(-1, 3)
    raw = eval_points(params, grids, pts, stage, bound, coarse_enlarge).reshape(z.shape[0], z.shape[1], 4)
    depth, var, rgb, w = composite(raw, z)
    if n_importance > 0:
        mid = .5 * (z[..., 1:] + z[..., :-1])
        extra = inverse_cdf_samples(mid, w[..., 1:-1], n_importance, det=det).detach()
        z = torch.sort(torch.cat([z, extra], -1), -1)[0]
        pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]).reshape(-1, 3)
        raw = eval_points(params, grids, pts, stage, bound, coarse_enlarge).reshape(z.shape[0], z.shape[1], 4)
        depth, var, rgb, w = composite(raw, z)
    if return_aux:
        return depth, var, rgb, dict(z_vals=z, pts=pts, raw=raw, weights=w)
    return depth, var, rgb


def mapper_loss(depth, color, gt_depth, gt_color, stage, w_color=0.2):
    """Mapper.py:553-562: L1 depth on valid pixels (+ w_color * L1 colour in the colour stage)."""
    m = gt_depth > 0
    loss = torch.abs(gt_depth[m] - depth[m]).sum()
    if stage == 'color':
        loss = loss + w_color * torch.abs(gt_color - color).sum()
    return loss
