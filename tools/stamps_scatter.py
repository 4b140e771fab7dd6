"""(make -C evennicer-slam_amd/csrc stamps3; ENSLAM_LIB=build/exp/libexp_stamps3.so ENSLAM_LIB_ALLOW_MISSING=1 ENSLAM_DEFER_SCATTER=1
python tools/stamps_scatter.py)  Per-segment wave cycles of grid_scatter_kernel (csrc/grid_scatter.hip)."""
import ctypes, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import evennicer_slam_amd as E
NSEG, WAVES = 12, 16
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
rays = bench.make_rays(sc, 1000, 1000)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
grids = {k: v.to(dev).requires_grad_(True) for k, v in sc['grids'].items()}
ro, rd, gd, gc = [t.to(dev) for t in rays]
ro.requires_grad_(True); rd.requires_grad_(True)
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
n_wg = 63 * 3
buf = torch.zeros(n_wg * WAVES * NSEG, dtype=torch.int64, device=dev)
handle = ctypes.CDLL(E.LIB_PATH)
assert handle.enslam_debug_set_stamp_buffer3(ctypes.c_void_p(buf.data_ptr())) == 0
for i in range(5):
    buf.zero_()
    d, v, c = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
    E.losses.rgbd_loss(d, c, gd, gc, 0.2).backward()
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(n_wg, WAVES, NSEG).astype(np.float64)
names = ["table clear + key", "sort", "rays, loads of all units, geometry", "scale (barrier)", "probes", "adds", "wait at the last barrier", "flush"]
s = a.reshape(-1, NSEG)
s = s[s[:, NSEG - 1] > 0]
tot = s[:, :NSEG - 1].sum(1)
rt = s[:, NSEG - 1]
print(f"waves {s.shape[0]}, cycles per wave mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); realtime span mean {rt.mean() / 100:.1f} us (max {rt.max() / 100:.1f})")
for k, nm in enumerate(names):
    print(f"    {nm:38s} {s[:, k].mean():9.0f} cycles {100 * s[:, k].mean() / tot.mean():5.1f} %   max {s[:, k].max():9.0f}")
wg = a[:, :, NSEG - 1].max(1)
print("workgroup span us: mean %.1f  p10 %.1f  p90 %.1f  max %.1f" % (wg.mean() / 100, np.percentile(wg, 10) / 100, np.percentile(wg, 90) / 100, wg.max() / 100))
