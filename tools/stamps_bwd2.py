"""(make -C evennicer-slam_amd/csrc stamps2; ENSLAM_LIB=build/exp/libexp_stamps2.so python tools/stamps_bwd2.py)
Per-segment wave cycles of the two-kernel backward (render_bwd2.hip): the chain kernel and the weight-gradient kernel."""
import ctypes, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import evennicer_slam_amd as E
NSEG = 12
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
rays = bench.make_rays(sc, 1000, 1000)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
grids = {k: v.to(dev).requires_grad_(True) for k, v in sc['grids'].items()}
ro, rd, gd, gc = [t.to(dev) for t in rays]
ro.requires_grad_(True); rd.requires_grad_(True)
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
buf = torch.zeros(2 * 256 * 16 * NSEG, dtype=torch.int64, device=dev)       # [kernel: chain, dw][workgroup][wave][segment]
handle = ctypes.CDLL(E.LIB_PATH)
assert handle.enslam_debug_set_stamp_buffer2(ctypes.c_void_p(buf.data_ptr())) == 0
for i in range(5):
    buf.zero_()
    d, v, c = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
    E.losses.rgbd_loss(d, c, gd, gc, 0.2).backward()
torch.cuda.synchronize()
allb = buf.cpu().numpy().reshape(2, 256, 16, NSEG).astype(np.float64)
names_dw = ["barrier", "fill issue", "fragment reads + products", "wait for fill n+1", "deposit tiles", "scatter", "drain"]
names_ch = ["prefetched loads arrive", "issue next loads + output layer", "scatter pieces", "backward layers", "cos, d_arg, dp", "dB^T",
            "hand-off + staging", "last scatter"]
for which, a, names in (("chain", allb[0], names_ch), ("dw", allb[1], names_dw)):
    used = a[:, :, NSEG - 1] > 0
    s = a[used]
    tot = s[:, :NSEG - 1].sum(1)
    rt = s[:, NSEG - 1]
    print(f"{which}: waves {s.shape[0]}, cycles per wave mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); realtime span mean {rt.mean() / 100:.1f} us "
          f"(max {rt.max() / 100:.1f}) -> clock {tot.mean() / max(rt.mean(), 1) * 100:.0f} MHz")
    for k, nm in enumerate(names):
        print(f"    {nm:34s} {s[:, k].mean():10.0f} cycles  {100 * s[:, k].mean() / tot.mean():5.1f} %")
