"""Scene geometry the path consumes: rounded bound and grid shapes.

Mirrors EvenNICER_SLAM.load_bound / grid_init (src/EvenNICER_SLAM.py:162-182, 217-275 in the reference);
written independently, results pinned bit-exact by tests/golden/bounds.npz."""
import numpy as np
import torch

GRID_KEYS = ('grid_coarse', 'grid_middle', 'grid_fine', 'grid_color')


def scene_bound(cfg_bound, scale=1.0, bound_divisible=0.32):
    """float64 [3,2].  hi = lo + float32(k*divisible): the reference's int32-tensor * python-float
    product is float32, and that rounding is part of the scene definition."""
    b = torch.as_tensor(np.asarray(cfg_bound, dtype=np.float64) * scale).clone()
    k = torch.trunc((b[:, 1] - b[:, 0]) / bound_divisible).to(torch.int32) + 1
    span32 = k.to(torch.float32) * torch.tensor(bound_divisible, dtype=torch.float32)
    b[:, 1] = span32.double() + b[:, 0]
    return b


def grid_shapes(bound, grid_len, coarse_bound_enlarge=2):
    """{'grid_coarse'|...: [D,H,W]} -- (z,y,x) order, i.e. x is the contiguous axis of [1,C,D,H,W]."""
    ext = bound[:, 1] - bound[:, 0]
    out = {}
    for name in ('coarse', 'middle', 'fine', 'color'):
        e = ext * coarse_bound_enlarge if name == 'coarse' else ext
        nx, ny, nz = [int(v) for v in (e / grid_len[name]).tolist()]
        out['grid_' + name] = [nz, ny, nx]
    return out


def grid_init(bound, grid_len, c_dim=32, coarse_bound_enlarge=2, device='cpu', std=None, memory_format=None):
    """Feature grids [1,c_dim,D,H,W] ~ N(0, std): 0.01 (fine: 1e-4) as EvenNICER_SLAM.py:248-272.
    memory_format=torch.channels_last_3d: the same tensors (shape, values, indexing) stored voxel-major -- the layout the HIP
    kernels gather from, so nothing is converted per render call (functional.is_native_grid)."""
    std = std or {'grid_coarse': 0.01, 'grid_middle': 0.01, 'grid_fine': 0.0001, 'grid_color': 0.01}
    shapes = grid_shapes(bound, grid_len, coarse_bound_enlarge)
    out = {k: torch.zeros([1, c_dim, *shapes[k]]).normal_(mean=0, std=std[k]).to(device) for k in GRID_KEYS}
    if memory_format is not None:
        out = {k: v.contiguous(memory_format=memory_format) for k, v in out.items()}
    return out
