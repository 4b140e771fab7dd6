"""Helpers for the -m gpu tests: build the HIP-backed Renderer / NICE from golden fixtures."""
import types

import torch

import evennicer_slam_amd as E
from tests.util import GRID_KEYS, load

DEV = 'cuda:0'


def cfg_like(n_samples=32, n_surface=16, grid_len=None):
    return {
        'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': n_samples, 'N_surface': n_surface,
                      'N_importance': 0},
        'scale': 1, 'occupancy': True, 'coarse': True, 'data': {'dim': 3},
        'model': {'c_dim': 32, 'coarse_bound_enlarge': 2, 'pos_embedding_method': 'fourier'},
        'grid_len': grid_len or {'coarse': 2, 'middle': 0.32, 'fine': 0.16, 'color': 0.16, 'bound_divisible': 0.32},
    }


def model_from_state(sd_arrays, bound, device=DEV):
    """NICE with the fixture's weights; bounds assigned as EvenNICER_SLAM.load_bound does."""
    model = E.get_model(cfg_like())
    sd = {k[3:]: torch.from_numpy(v.copy()) for k, v in sd_arrays.items() if k.startswith('sd_')}
    model.load_state_dict(sd)
    model = model.to(device)
    model.bound = bound
    for name in ('middle_decoder', 'fine_decoder', 'color_decoder'):
        getattr(model, name).bound = bound
    model.coarse_decoder.bound = bound * 2
    return model


def renderer_for(bound, cam=(48, 64, 50.0, 50.0, 31.5, 23.5), cfg=None):
    H, W, fx, fy, cx, cy = cam
    slam = types.SimpleNamespace(nice=True, bound=bound, H=int(H), W=int(W), fx=fx, fy=fy, cx=cx, cy=cy)
    return E.Renderer(cfg or cfg_like(), None, slam)


def tiny_on_gpu():
    s = load('tiny_scene')
    bound = torch.from_numpy(s['bound'].copy())
    model = model_from_state(s, bound)
    grids = {k: torch.from_numpy(s[k].copy()).to(DEV) for k in GRID_KEYS}
    rays = {k: torch.from_numpy(s[k].copy()).to(DEV) for k in ('rays_o', 'rays_d', 'gt_depth', 'gt_color')}
    return s, bound, model, grids, rays, renderer_for(bound)


def as_layout(t, layout):
    """a detached copy of grid `t` ([1,32,D,H,W]) in the reference's contiguous layout or in torch.channels_last_3d -- the memory
    format whose storage is the kernels' own [V][32] (functional.is_native_grid): same shape, values and indexing"""
    t = t.detach().clone()
    return t.contiguous(memory_format=torch.channels_last_3d) if layout == 'channels_last_3d' else t


LAYOUTS = ['contiguous', 'channels_last_3d']
