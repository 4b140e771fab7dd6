"""cProfile of the eager step loop (host-side cost per step)."""
import cProfile, pstats, sys, os, types, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
rays = bench.make_rays(sc, 1000, 1000)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
mf = torch.channels_last_3d if os.environ.get('LAYOUT', 'contiguous') == 'channels_last_3d' else torch.contiguous_format
grids = {k: v.to(dev).contiguous(memory_format=mf).requires_grad_(True) for k, v in sc['grids'].items()}
ro, rd, gd, gc = [t.to(dev) for t in rays]
ro.requires_grad_(True); rd.requires_grad_(True)
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
leaves = [grids[k] for k in ('grid_middle','grid_fine','grid_color')] + [p for n in ('middle_decoder','fine_decoder','color_decoder') for p in getattr(model, n).parameters()]
def step():
    torch._C._increment_version(leaves)
    for t in leaves: t.grad = None
    ro.grad = None; rd.grad = None
    d, v, c = renderer.render_batch_ray(grids, model, rd, ro, dev, 'color', gt_depth=gd)
    (bench.mapper_loss(d, c, gd, gc, 'color') if os.environ.get('TORCH_LOSS', '1') == '1' else E.losses.rgbd_loss(d, c, gd, gc, 0.2)).backward()
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host-only per step {1e3*(t1-t0)/200:.3f} ms ; with final sync {1e3*(t2-t0)/200:.3f} ms")
torch.autograd.set_multithreading_enabled(False)      # the engine then calls the Python backward on this thread: cProfile sees it
if os.environ.get('NO_CPROFILE') == '1': sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30); print(s.getvalue()[:7000])
