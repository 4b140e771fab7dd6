"""Ray generation and pixel sampling (reference: src/common.py:74-107,130-169,300-340).

These are a handful of tiny, differentiable-in-c2w PyTorch-ROCm ops (<1 % of a step) and the
only RNG draw on the path: they stay in PyTorch so that `torch.randint` consumes the caller's generator
exactly as the reference does (bit-exact sample indices), and so pose gradients flow through autograd."""
import numpy as np
import torch


def get_rays_from_uv(i, j, c2w, H, W, fx, fy, cx, cy, device):
    """Rays through pixels (i = column, j = row): rays_d = R @ [(i-cx)/fx, -(j-cy)/fy, -1], rays_o = t."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w).to(device)
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1).to(device)
    rays_d = (dirs.reshape(-1, 1, 3) * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def select_uv(i, j, n, depth, color, device='cuda:0'):
    """n uniformly drawn pixels (with replacement) of the flattened window."""
    i, j = i.reshape(-1), j.reshape(-1)
    idx = torch.randint(i.shape[0], (n,), device=device)
    return i[idx], j[idx], depth.reshape(-1)[idx], color.reshape(-1, 3)[idx]


def get_sample_uv(H0, H1, W0, W1, n, depth, color, device='cuda:0'):
    # (integer-valued with unit step: the same values whichever device computes them; made on the device so that a
    # captured iteration performs no host-to-device copy)
    # the reference materialises the window's meshgrid and indexes it (common.py:137-141); indexing the two axes gives
    # the same values without the 2 x H x W temporaries.  Same single RNG draw as select_uv.
    ww = W1 - W0
    idx = torch.randint((H1 - H0) * ww, (n,), device=device)
    if idx.is_cuda:
        from . import functional as EF          # (lazy: this module has no other dependency on the HIP library)
        if EF.gather_pixels_ok(idx, depth, color):
            return EF.gather_pixels(idx, H0, W0, ww, depth, color)     # one launch instead of eight
    cols = torch.linspace(W0, W1 - 1, W1 - W0, device=device)
    rows = torch.linspace(H0, H1 - 1, H1 - H0, device=device)
    col, row = idx % ww, idx // ww
    d = depth[H0:H1, W0:W1][row, col]
    c = color[H0:H1, W0:W1][row, col]
    return cols[col], rows[row], d, c


def get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2w, depth, color, device):
    """n rays from the image window [H0,H1) x [W0,W1) with their depth / colour samples."""
    i, j, d, c = get_sample_uv(H0, H1, W0, W1, n, depth, color, device=device)
    rays_o, rays_d = get_rays_from_uv(i, j, c2w, H, W, fx, fy, cx, cy, device)
    return rays_o, rays_d, d, c


def _image_rays(H, W, new_H, new_W, fx, fy, cx, cy, c2w, device):
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    cols = torch.linspace(0, W - 1, new_W)
    rows = torch.linspace(0, H - 1, new_H)
    j, i = torch.meshgrid(rows, cols, indexing='ij')
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1).to(device)
    c2w = c2w.to(device)
    rays_d = (dirs.reshape(new_H, new_W, 1, 3) * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays(H, W, fx, fy, cx, cy, c2w, device):
    """Rays of the whole image, [H,W,3] each."""
    return _image_rays(H, W, H, W, fx, fy, cx, cy, c2w, device)


def get_rays_rescale(H, W, new_H, new_W, fx, fy, cx, cy, c2w, device):
    """Rays of the image strided down to (new_H,new_W) pixel centres (not averaged)."""
    return _image_rays(H, W, new_H, new_W, fx, fy, cx, cy, c2w, device)


def normalize_3d_coordinate(p, bound):
    """[-1,1] coordinates of p inside bound (float64 arithmetic when p is float64)."""
    p = p.reshape(-1, 3)
    out = torch.empty_like(p)
    for a in range(3):
        out[:, a] = ((p[:, a] - bound[a, 0]) / (bound[a, 1] - bound[a, 0])) * 2 - 1.0
    return out


def raw2outputs_nerf_color(raw, z_vals, rays_d, occupancy=False, device='cuda:0'):
    """Alpha compositing of per-sample (r,g,b,occ) -> depth, depth variance, colour, weights
    (reference: src/common.py:256-297).  Only the occupancy branch -- the one every shipped NICE config
    selects (configs/nice_slam.yaml:5) -- is built; it runs as one HIP kernel (and one for its backward)."""
    if not occupancy:
        raise NotImplementedError("volume-density compositing (occupancy=False) is the iMAP mode, out of scope")
    from . import functional as EF
    return EF.composite(raw, z_vals)


# ------------------------------------------------------------------------------------------------
# camera tensor <-> pose (reference: src/common.py:189-254).  The tracker and the BA mapper optimise a 7-vector
# [qr, qi, qj, qk, tx, ty, tz]; its gradient reaches the path through rays_o / rays_d.
# ------------------------------------------------------------------------------------------------
def quad2rotation(quad):
    """Rotation matrices [B,3,3] of (unnormalised) quaternions [B,4] = (real, i, j, k); differentiable."""
    qr, qi, qj, qk = quad.unbind(-1)
    two_s = 2.0 / (quad * quad).sum(-1)
    rows = (
        1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
        two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr),
        two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2),
    )
    return torch.stack(rows, -1).reshape(quad.shape[:-1] + (3, 3))


def get_camera_from_tensor(inputs):
    """[.., 7] (quaternion, translation) -> camera-to-world [.., 3, 4]."""
    single = inputs.dim() == 1
    x = inputs[None] if single else inputs
    RT = torch.cat([quad2rotation(x[:, :4]), x[:, 4:, None]], 2)
    return RT[0] if single else RT


def get_tensor_from_camera(RT, Tquad=False):
    """camera-to-world [3|4, 4] -> 7-vector (quaternion first, or translation first with Tquad), a float32 tensor on
    the CPU whatever RT's device -- exactly what the reference returns (src/common.py:231-252 moves RT to the host
    and its `gpu_id` is read after that move, so the result never goes back to the device).

    The reference calls `mathutils.Matrix(R).to_quaternion()`; mathutils (pinned 2.81.2, environment.yaml:150) is
    not in this image, so this restates Blender 2.81's `mat3_to_quat` = `normalize_m3` + `mat3_normalized_to_quat`
    branch for branch: float32 matrix, columns normalised, the trace branch whenever 0.25 * (1 + trace) > 1e-4
    (real part > 0), otherwise the largest-diagonal branches WITHOUT forcing the real part positive (near 180 degree
    rotations can come back with a small negative real part, as from mathutils), then `normalize_qt`.  Sums in
    float32 / float64 as in the C source.  Pinned on w >= 0 cases by tests/golden/ate_cases.npz and the tracker
    fixture; the negative-w branches have no reference-generated fixture (mathutils absent)."""
    if not torch.is_tensor(RT):
        RT = torch.from_numpy(np.asarray(RT))
    M = RT.detach().to('cpu')
    T = M[:3, 3].to(torch.float32)
    f32 = np.float32
    R = M[:3, :3].to(torch.float32).numpy().astype(np.float32)
    # Blender stores mat[col][row]: its row vectors are the columns of R; normalize_m3 normalises those
    mat = R.T.copy()
    for i in range(3):
        n = f32(np.sqrt(f32(mat[i, 0] * mat[i, 0] + mat[i, 1] * mat[i, 1] + mat[i, 2] * mat[i, 2])))
        if n > f32(0):
            mat[i] = mat[i] / n
    q = np.zeros(4, dtype=np.float32)
    tr = 0.25 * float(f32(f32(f32(f32(1.0) + mat[0, 0]) + mat[1, 1]) + mat[2, 2]))
    if tr > float(f32(1e-4)):
        s = np.sqrt(tr)
        q[0] = f32(s)
        s = 1.0 / (4.0 * s)
        q[1] = f32(float(f32(mat[1, 2] - mat[2, 1])) * s)
        q[2] = f32(float(f32(mat[2, 0] - mat[0, 2])) * s)
        q[3] = f32(float(f32(mat[0, 1] - mat[1, 0])) * s)
    elif mat[0, 0] > mat[1, 1] and mat[0, 0] > mat[2, 2]:
        s = float(f32(2.0) * f32(np.sqrt(f32(f32(f32(f32(1.0) + mat[0, 0]) - mat[1, 1]) - mat[2, 2]))))
        q[1] = f32(0.25 * s)
        s = 1.0 / s
        q[0] = f32(float(f32(mat[1, 2] - mat[2, 1])) * s)
        q[2] = f32(float(f32(mat[1, 0] + mat[0, 1])) * s)
        q[3] = f32(float(f32(mat[2, 0] + mat[0, 2])) * s)
    elif mat[1, 1] > mat[2, 2]:
        s = float(f32(2.0) * f32(np.sqrt(f32(f32(f32(f32(1.0) + mat[1, 1]) - mat[0, 0]) - mat[2, 2]))))
        q[2] = f32(0.25 * s)
        s = 1.0 / s
        q[0] = f32(float(f32(mat[2, 0] - mat[0, 2])) * s)
        q[1] = f32(float(f32(mat[1, 0] + mat[0, 1])) * s)
        q[3] = f32(float(f32(mat[2, 1] + mat[1, 2])) * s)
    else:
        s = float(f32(2.0) * f32(np.sqrt(f32(f32(f32(f32(1.0) + mat[2, 2]) - mat[0, 0]) - mat[1, 1]))))
        q[3] = f32(0.25 * s)
        s = 1.0 / s
        q[0] = f32(float(f32(mat[0, 1] - mat[1, 0])) * s)
        q[1] = f32(float(f32(mat[2, 0] + mat[0, 2])) * s)
        q[2] = f32(float(f32(mat[2, 1] + mat[1, 2])) * s)
    ln = f32(np.sqrt(f32(f32(f32(q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3])))      # normalize_qt
    if ln != f32(0):
        q = (q * f32(f32(1.0) / ln)).astype(np.float32)
    else:
        q = np.array([1, 0, 0, 0], dtype=np.float32)
    quad = torch.from_numpy(q)
    # the reference concatenates float64 numpy pieces (mathutils floats widen to Python floats) and casts .float()
    return torch.cat([T, quad]) if Tquad else torch.cat([quad, T])


def sample_pdf(bins, weights, N_samples, det=False, device='cuda:0'):
    """Inverse-CDF ("hierarchical", NeRF sec. 5.2) samples of the piecewise-constant density `weights` over `bins`
    (src/common.py:19-63): bins [B, M], weights [B, M-1] -> samples [B, N_samples].  det: evenly spaced quantiles,
    else one torch.rand draw.  Plain torch ops on the inputs' device: only iMAP / N_importance > 0 use it, which no
    shipped NICE configuration does; the HIP renderer itself raises for N_importance > 0."""
    w = weights + 1e-5
    cdf = torch.cumsum(w / w.sum(-1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    lead = list(cdf.shape[:-1])
    if det:
        u = torch.linspace(0., 1., steps=N_samples).expand(lead + [N_samples])
    else:
        u = torch.rand(lead + [N_samples])
    u = u.to(device).contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b0, b1 = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)
