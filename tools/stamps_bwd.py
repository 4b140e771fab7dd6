"""(make -C evennicer-slam_amd/csrc stamps; ENSLAM_LIB=build/exp/libexp_stamps.so python tools/stamps_bwd.py)
Run the stamps build of the backward kernel and print the share of wave cycles per segment, per role."""
import ctypes, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF
NSEG = 12
names = ["d_raw+vote", "deferred scatter", "previous dX arrives", "fill issue + h4/mask loads", "wait for dW reads", "ring waits", "layer deposits",
         "owned dW MFMAs", "tail dX", "emb tail", "coord+scatter+rays", "-"]
dev = torch.device('cuda', 0)
lib = E._lib.lib()
sc = bench.build_scene_cpu('room0', 0)
rays = bench.make_rays(sc, 1000, 1000)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
grids = {k: v.to(dev).requires_grad_(True) for k, v in sc['grids'].items()}
ro, rd, gd, gc = [t.to(dev) for t in rays]
ro.requires_grad_(True); rd.requires_grad_(True)
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
TLB = 2 * 256 * 4 * NSEG
buf = torch.zeros(TLB + 256 * 8 * 16, dtype=torch.int64, device=dev)
handle = ctypes.CDLL(E.LIB_PATH)
assert handle.enslam_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
for i in range(5):
    buf.zero_()
    d, v, c = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
    E.losses.rgbd_loss(d, c, gd, gc, 0.2).backward()
torch.cuda.synchronize()
allb = buf.cpu().numpy()
both = allb[:TLB].reshape(2, 256, 4, NSEG).astype(np.float64)
tl = allb[TLB:].reshape(256, 8, 16).astype(np.float64)
st, sd = both[0], both[1]
# role ranges as in ens_launch_decoder_bwd: 0.30 / 0.40 / 0.30 of 256 workgroups
r0 = 75; r1 = r0 + 94          # split chosen by the launcher for 3000 tiles (see ens_launch_decoder_bwd)
for name, sl in (("middle", slice(0, r0)), ("fine", slice(r0, r1)), ("color", slice(r1, 256))):
    s = st[sl].reshape(-1, NSEG)
    tot = s[:, :NSEG - 1].sum(1)
    print(f"role {name}: waves {s.shape[0]}, cycles per wave mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}) = {tot.mean()/2.4e3:.0f} us @2.4GHz")
    rt = s[:, NSEG - 1].mean(); cyc = s[:, :NSEG - 1].sum(1).mean()
    print(f"    s_memrealtime (100 MHz) {rt:.0f} ticks = {rt / 100:.1f} us; s_memtime {cyc:.0f} ticks -> {cyc / max(rt, 1) * 100:.0f} MHz")
    for k in range(NSEG - 1):
        print(f"    {names[k]:28s} {s[:, k].mean():10.0f} cycles  {100 * s[:, k].mean() / tot.mean():5.1f} %")

dnames = ["vote barrier", "fill issue", "wait for deposits", "slot fill landed", "owned products", "scatter piece", "wait for d_arg", "dB^T", "-", "-", "-", "-"]
for name, sl in (("middle", slice(0, r0)), ("fine", slice(r0, r1)), ("color", slice(r1, 256))):
    s = sd[sl].reshape(-1, NSEG)
    tot = s.sum(1)
    print(f"dW waves, role {name}: cycles per wave mean {tot.mean():.0f} (before the flush)")
    for k in range(8):
        print(f"    {dnames[k]:28s} {s[:, k].mean():10.0f} cycles  {100 * s[:, k].mean() / tot.mean():5.1f} %")

# ---- end-to-end budget of one launch from the absolute s_memrealtime marks (100 MHz): slot 0 kernel entry, 1 prologue done,
# 2..9 end of executed round r, 10 loop exit, 11 last scatter done (dW waves), 12 flush done
t0 = tl[:, :, 0][tl[:, :, 0] > 0].min()
def us(x): return (x - t0) / 100.0
print("\nkernel budget (us after the first wave's entry; mean [min .. max] over workgroups)")
for name, sl in (("middle", slice(0, r0)), ("fine", slice(r0, r1)), ("color", slice(r1, 256))):
    for half, w in (("chain", 0), ("dW", 4)):
        T = tl[sl, w, :]
        def f(k):
            v = us(T[:, k][T[:, k] > 0])
            return f"{v.mean():6.1f} [{v.min():6.1f} .. {v.max():6.1f}]" if v.size else "   -"
        nr = int((T[:, 2:10] > 0).sum(1).max())
        print(f"  {name:6s} {half:5s}: entry {f(0)}  prologue {f(1)}  " + "  ".join(f"r{r} {f(2 + r)}" for r in range(nr)))
        print(f"               loop exit {f(10)}  last scatter {f(11)}  flush done {f(12)}")
end = us(tl[:, :, 12].max())
print(f"  last flush done {end:.1f} us after the first entry")
