"""Shared helpers for the tests: fixture loading and comparison metrics."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRID_KEYS = ['grid_coarse', 'grid_middle', 'grid_fine', 'grid_color']
STAGES = ['coarse', 'middle', 'fine', 'color']


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def tiny_scene():
    """(params dict of tensors keyed like the reference state_dict, grids, bound f64, scene arrays)."""
    s = load("tiny_scene")
    params = {k[3:]: torch.from_numpy(v.copy()) for k, v in s.items() if k.startswith("sd_")}
    grids = {k: torch.from_numpy(s[k].copy()) for k in GRID_KEYS}
    bound = torch.from_numpy(s["bound"].copy())
    return params, grids, bound, s


def rel_err(a, b):
    """max |a-b| / max(|b|, eps): error relative to the largest reference magnitude."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elem_rel_err(a, b, floor):
    """max elementwise |a-b| / max(|b|, floor)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def within_rel(a, b, rel=1e-4, floor=1e-2):
    """north_star's output bar as a PURE relative test with an explicit near-zero floor: |a - b| <= rel * max(|b|, floor * max|b|)
    elementwise -- entries smaller than `floor` of the tensor's largest magnitude are compared against that magnitude (a
    relative test of a value that is itself rounding noise says nothing)."""
    import numpy as np
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), floor * np.abs(b).max())
    return bool(np.all(np.abs(a - b) <= rel * scale)), float((np.abs(a - b) / scale).max())
