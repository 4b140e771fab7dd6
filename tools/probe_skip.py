"""Backward time when far tiles carry no gradient (opaque surface right in front of the camera: the transmittance
underflows after a few samples, d_raw of the later tiles is exactly zero): the work list of the saved-activation backward
leaves those tiles out.  Compares the bench scene as is with the same scene after raising the fine decoder's output bias;
ENSLAM_WORK_LIST=0 walks every tile."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF
dev = torch.device('cuda', 0)
sc = bench.build_scene_cpu('room0', 0)
model = sc['model'].to(dev); bench.attach_bounds(model, sc['bound'])
grids = {k: v.to(dev).requires_grad_(True) for k, v in sc['grids'].items()}
ro, rd, gd, gc = [t.to(dev) for t in bench.make_rays(sc, 1000, 1000)]
renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **bench.CAM))
def run(tag):
    renderer.state.profile['decoder_bwd'] = []
    for _ in range(30):
        EF.clear_caches()
        for g in grids.values(): g.grad = None
        loss, depth, var, color = renderer.render_batch_ray_rgbd_loss(grids, model, rd, ro, dev, 'color', gd, gc, 0.2)
        loss.backward()
    torch.cuda.synchronize()
    ev = renderer.state.profile.pop('decoder_bwd')
    t = np.array([a.elapsed_time(b) for a, b in ev][5:]) * 1e3
    print(f"work list {'on' if renderer.state.use_work_list else 'off'}; {tag}: decoder backward {t.mean():.1f} us (min {t.min():.1f}), loss {loss.item():.1f}, "
          f"|grad fine| {float(grids['grid_fine'].grad.abs().sum()):.6e}")
run("bench scene (every tile carries gradient)")
with torch.no_grad():
    model.fine_decoder.output_linear.bias += 5.0
run("opaque from the first samples on (tiles 2 and 3 of every ray have d_raw == 0)")
