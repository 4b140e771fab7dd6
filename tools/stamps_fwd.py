"""(make -C evennicer-slam_amd/csrc stamps; ENSLAM_LIB=build/exp/libexp_stamps.so python tools/stamps_fwd.py)
Run the stamps build of the forward ring kernel and print the share of wave cycles per segment."""
import ctypes, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import evennicer_slam_amd as E
NSEG = 12
names = ["geometry + first gather", "embedding (B^T, MFMA, sin)", "ws stores emb/c/coords", "layer compute", "ws stores h",
         "barrier + chunk wait", "output layer", "later gathers", "raw store", "-", "-", "-"]
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
N = int(os.environ.get('RAYS', 1000))
rays = bench.make_rays(sc, N, 1000)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
grids = {k: v.to(dev).requires_grad_(True) for k, v in sc['grids'].items()}
ro, rd, gd, gc = [t.to(dev) for t in rays]
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
nwg = (3 * N + 3) // 4
buf = torch.zeros(nwg * 4 * NSEG, dtype=torch.int64, device=dev)
handle = ctypes.CDLL(E.LIB_PATH)
assert handle.enslam_debug_set_stamp_buffer_fwd(ctypes.c_void_p(buf.data_ptr())) == 0
for i in range(5):
    buf.zero_()
    d, v, c = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(nwg * 4, NSEG).astype(np.float64)
tot = st.sum(1)
print(f"rays {N}: waves {st.shape[0]}, cycles per wave mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f})")
for k in range(9):
    print(f"    {names[k]:30s} {st[:, k].mean():10.0f} cycles  {100 * st[:, k].mean() / tot.mean():5.1f} %")
