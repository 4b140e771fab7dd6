"""time the sampler alone (with / without block marking) through the C ABI"""
import os, sys, time, types, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF
from evennicer_slam_amd import _lib as L
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
ro, rd, gd, gc = [t.to(dev) for t in bench.make_rays(sc, 1000, 1000)]
lib = L.lib()
N, n_lin, n_surf = 1000, 32, 16
z = torch.empty((N, 48), dtype=torch.float64, device=dev)
t_lin = torch.linspace(0., 1., n_lin, device=dev)
t_surf = torch.linspace(0., 1., n_surf, device=dev).double()
scratch = torch.empty(2, device=dev)
b6 = EF.bound6(sc['bound'])
dims = {k: tuple(sc['grids'][L.GRID_NAMES[k]].shape[2:]) for k in (1, 2, 3)}
nblk = {k: (dims[k][0] * dims[k][1] * dims[k][2] + 63) // 64 for k in dims}
flags = {k: torch.zeros(nblk[k], dtype=torch.uint8, device=dev) for k in dims}
msc = L.Scene(); msc.bound = b6; msc.coarse_bound = b6
fptr = (ctypes.c_void_p * 4)()
for k in dims:
    msc.grids[k].D, msc.grids[k].H, msc.grids[k].W = dims[k]
    fptr[k] = flags[k].data_ptr()
st = EF._stream()
def run(mark):
    L.check(lib.enslam_sample_rays(N, n_lin, n_surf, EF._ptr(ro), EF._ptr(rd), EF._ptr(gd), b6, EF._ptr(t_lin), EF._ptr(t_surf), 0,
                                   None, EF._ptr(scratch), 0, EF._ptr(z), 3, ctypes.byref(msc) if mark else None, fptr if mark else None, EF._stream()), "s")
for mark in (True, False):
    for _ in range(5): run(mark)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st = EF._stream()
        run(mark)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(50): run(mark)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 1000
    print(f"sample_kernel x50 in a graph, marking={mark}: {t * 1e6:.2f} us per launch")
