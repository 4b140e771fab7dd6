"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/enslam_hip.h declares,
host logic matches the reference goldens, and the product path refuses to run without a HIP device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.util import GRID_KEYS, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    import __graft_entry__ as G
    G.build()


def test_abi_library_exports_every_declared_symbol():
    _build()
    import evennicer_slam_amd as E
    header = open(os.path.join(ROOT, "include", "enslam_hip.h")).read()
    declared = set(re.findall(r"\b(enslam_[a-z_0-9]+)\s*\(", header))
    declared -= {"enslam_packed_floats(kind)"}
    assert len(declared) >= 17
    handle = ctypes.CDLL(E.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in enslam_hip.h but not exported"
    assert set(E._lib.EXPORTS) == declared, set(E._lib.EXPORTS) ^ declared
    lib = E._lib.lib()
    assert lib.enslam_abi_version() == 1 and lib.enslam_arch() == b"gfx950"
    # packed sizes are pure host arithmetic: decoder parameter counts + padding
    assert [lib.enslam_packed_grad_floats(k) for k in range(4)] == [6832, 16592, 21712, 16592]


def test_scene_helpers_match_reference():
    import evennicer_slam_amd as E
    b = load("bounds")
    for tag in ("room0", "office0", "recording4"):
        bound = E.scene.scene_bound(b[tag + "_cfg_bound"], float(b[tag + "_scale"]), 0.32)
        assert np.array_equal(bound.numpy(), b[tag + "_bound"])
        shp = E.scene.grid_shapes(bound, dict(coarse=2, middle=0.32, fine=0.16, color=0.16))
        assert [shp[k] for k in GRID_KEYS] == b[tag + "_shapes"].tolist()


def test_ray_generation_bit_exact_and_pose_gradient():
    from evennicer_slam_amd import common as C
    r = load("rays")
    H, W, fx, fy, cx, cy = r["cam"]
    H, W = int(H), int(W)
    depth, color = torch.from_numpy(r["depth_img"]), torch.from_numpy(r["color_img"])
    c2w = torch.from_numpy(r["c2w"]).requires_grad_(True)
    for tag in ("full", "edge"):
        H0, H1, W0, W1, n = [int(v) for v in r[tag + "_args"]]
        torch.manual_seed(int(r["seed"]))
        ro, rd, d, c = C.get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2w, depth, color, 'cpu')
        assert np.array_equal(ro.detach().numpy(), r[tag + "_rays_o"])
        assert np.array_equal(rd.detach().numpy(), r[tag + "_rays_d"])
        assert np.array_equal(d.numpy(), r[tag + "_depth"]) and np.array_equal(c.numpy(), r[tag + "_color"])
    torch.manual_seed(int(r["seed"]))
    ro, rd, _, _ = C.get_samples(0, H, 0, W, 100, H, W, fx, fy, cx, cy, c2w, depth, color, 'cpu')
    ((ro * torch.from_numpy(r["cot_o"])).sum() + (rd * torch.from_numpy(r["cot_d"])).sum()).backward()
    assert np.allclose(c2w.grad.numpy(), r["g_c2w"], rtol=1e-6, atol=1e-6)
    ro, rd = C.get_rays_rescale(H, W, int(H * 0.15), int(W * 0.15), fx, fy, cx, cy, c2w.detach(), 'cpu')
    assert np.array_equal(rd.numpy(), r["rescale_rays_d"]) and np.array_equal(ro.numpy(), r["rescale_rays_o"])
    ro, rd = C.get_rays(H, W, fx, fy, cx, cy, c2w.detach(), 'cpu')
    assert np.array_equal(rd.numpy()[::7, ::11], r["img_rays_d"])


def test_decoder_module_tree_matches_reference_state_dict():
    import bench
    g = load("room0_coarse200")
    sc = bench.build_scene_cpu('room0', seed=0)
    sd = sc['model'].state_dict()
    ref_keys = sorted(k[3:] for k in g if k.startswith("sd_"))
    assert sorted(sd.keys()) == ref_keys                                   # same key names / module tree
    for k in ref_keys:                                                     # same seeded initialisation
        assert np.array_equal(sd[k].numpy(), g["sd_" + k]), k
    assert sum(p.numel() for p in sc['model'].parameters()) == 58956
    import copy
    m2 = copy.deepcopy(sc['model'])
    m2.share_memory()
    m2.load_state_dict(sd)
    assert np.array_equal(sc['grids']['grid_coarse'].numpy(), g["grid_coarse"])


def test_product_path_refuses_cpu_tensors():
    import types
    import bench
    import evennicer_slam_amd as E
    sc = bench.build_scene_cpu('room0', seed=0)
    bench.attach_bounds(sc['model'], sc['bound'])
    r = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
    ro, rd, gd, gc = bench.make_rays(sc, 8, 0)
    with pytest.raises(E.EnslamError):
        r.render_batch_ray(sc['grids'], sc['model'], rd, ro, 'cpu', 'color', gt_depth=gd)
    with pytest.raises(E.EnslamError):
        r.eval_points(torch.zeros(4, 3, dtype=torch.float64), sc['model'], sc['grids'], 'color', 'cpu')
    cfg = dict(sc['cfg'])
    cfg['occupancy'] = False
    with pytest.raises(NotImplementedError):
        E.Renderer(cfg, None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
    with pytest.raises(NotImplementedError):
        r.regulation(None, None, None, None, None, 'cpu')


def test_missing_library_fails_loudly(monkeypatch):
    import evennicer_slam_amd as E
    monkeypatch.setattr(E._lib, "_lib", None)
    monkeypatch.setattr(E._lib, "LIB_PATH", "/nonexistent/libenslam_hip.so")
    with pytest.raises(E.EnslamError, match="no CPU fallback"):
        E._lib.lib()


def test_shard_ranges_cover_batch():
    from evennicer_slam_amd.parallel import shard_range
    for n in (0, 1, 7, 1000, 5000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_sample_pdf_matches_reference_fixture():
    """common.sample_pdf (src/common.py:19-63) against outputs of the reference's function (tests/golden/sample_pdf.npz)."""
    import numpy as np
    import torch
    from evennicer_slam_amd.common import sample_pdf
    from tests.util import load
    fx = load("sample_pdf")
    for case in range(3):
        bins, w, n = torch.from_numpy(fx[f'c{case}_bins']), torch.from_numpy(fx[f'c{case}_w']), int(fx[f'c{case}_n'])
        assert np.array_equal(sample_pdf(bins, w, n, det=True, device='cpu').numpy(), fx[f'c{case}_det'])
        torch.manual_seed(100 + case)
        assert np.array_equal(sample_pdf(bins, w, n, det=False, device='cpu').numpy(), fx[f'c{case}_rand'])


def test_ctypes_signatures_match_the_header():
    """Every entry of _lib._SIGS takes as many arguments, of the same broad kind (integer / floating point / pointer), as the
    declaration in include/enslam_hip.h -- an ABI drift between the header and the Python binding would otherwise only show as
    garbage arguments on the GPU."""
    import evennicer_slam_amd as E
    header = open(os.path.join(ROOT, "include", "enslam_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    decls = dict(re.findall(r"\b(enslam_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S))
    assert set(decls) >= set(E._lib._SIGS)

    def kind_c(param):
        p = " ".join(param.split())
        if p in ("", "void"):
            return None
        if "*" in p:
            return "ptr"
        return "fp" if re.search(r"\b(float|double)\b", p) else "int"

    def kind_py(t):
        if t in (ctypes.c_float, ctypes.c_double):
            return "fp"
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) is not None and not isinstance(
                getattr(t, "_type_"), str):
            return "ptr"
        return "int"

    for name, (_res, args) in E._lib._SIGS.items():
        want = [k for k in (kind_c(p) for p in decls[name].split(",")) if k is not None]
        got = [kind_py(t) for t in args]
        assert got == want, (name, got, want)
