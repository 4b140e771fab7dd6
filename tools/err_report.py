"""Maximum errors of the HIP path against the reference-generated fixtures of the three scene sizes (room0 1000 x 48, office0
5000 x 48, recording4 1000 x 48): rendered outputs, ray gradients, decoder-parameter gradients, sampled grid-gradient entries.
Errors are |got - ref|.max() / |ref|.max() per tensor (the scene tests' measure).  Run with the library under test:
    python tools/err_report.py                          # the shipped library
    ENSLAM_LIB=build/exp/libexp_<x>.so python tools/err_report.py"""
import gc, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import evennicer_slam_amd as E
import evennicer_slam_amd.functional as EF
from tests.util import load, rel_err

FIX = {'room0': 'room0_color1000', 'office0': 'office0_color5000', 'recording4': 'recording4_color1000'}
print("library:", E.LIB_PATH)
x = ((torch.rand(400000, generator=torch.Generator().manual_seed(3)) - 0.5) * 8000.0).cuda()
s, c = EF.fourier_sincos(x)
print(f"embedding sin / cos vs float64, |x| <= 4000: {float((s.double().cpu() - torch.sin(x.double().cpu())).abs().max()):.2e} / "
      f"{float((c.double().cpu() - torch.cos(x.double().cpu())).abs().max()):.2e} absolute")
for tag, fx in FIX.items():
    sc = bench.build_scene_cpu(tag, seed=0)
    g = load(fx)
    model = sc['model'].cuda(); bench.attach_bounds(model, sc['bound'])
    renderer = E.Renderer(sc['cfg'], None, types.SimpleNamespace(nice=True, bound=sc['bound'], **sc['cam']))
    t = lambda k: torch.from_numpy(g[k]).cuda()
    cg = {k: v.cuda().requires_grad_(True) for k, v in sc['grids'].items()}
    for p in model.parameters(): p.grad = None
    ro, rd, gd, gc_ = t('rays_o').requires_grad_(True), t('rays_d').requires_grad_(True), t('gt_depth'), t('gt_color')
    d, v, c = renderer.render_batch_ray(cg, model, rd, ro, 'cuda:0', 'color', gt_depth=gd)
    bench.mapper_loss(d, c, gd, gc_, 'color').backward()
    out = {n: rel_err(a.detach().cpu().numpy(), g[n]) for n, a in (('depth', d), ('var', v), ('color', c))}
    out['g_rays_o'] = rel_err(ro.grad.cpu().numpy(), g['g_rays_o']); out['g_rays_d'] = rel_err(rd.grad.cpu().numpy(), g['g_rays_d'])
    out['g_params(max of %d)' % sum(1 for n, _ in model.named_parameters() if 'gp_' + n in g)] = max(
        rel_err(p.grad.cpu().numpy(), g['gp_' + n]) for n, p in model.named_parameters() if 'gp_' + n in g and np.abs(g['gp_' + n]).max() > 0)
    for key in ('grid_middle', 'grid_fine', 'grid_color'):
        gg = cg[key].grad.reshape(-1)
        out['g_' + key] = rel_err(gg[torch.from_numpy(g['gidx_' + key]).cuda()].cpu().numpy(), g['gval_' + key])
    print(f"{tag:11s}", "  ".join(f"{k} {e:.2e}" for k, e in out.items()))
    # where the largest ray-gradient error sits, and how close that ray's colour is to a kink of the L1 loss
    e = np.abs(ro.grad.cpu().numpy() - g['g_rays_o']).max(1)
    r = int(e.argmax())
    dc_ref = np.abs(g['color'] - g['gt_color'])
    dc_got = (c.detach().cpu().numpy() - g['gt_color'])
    flips = np.argwhere(np.sign(dc_got) != np.sign(g['color'] - g['gt_color']))
    print(f"            worst ray {r}: |g_rays_o error| {e[r]:.3e} (second worst ray {np.sort(e)[-2]:.3e}); smallest |colour - gt| of the batch "
          f"{dc_ref.min():.3e} at ray {int(dc_ref.min(1).argmin())}; L1 sign flips vs the reference at (ray, channel) {flips.tolist()}")
    del model, cg, renderer
    EF.clear_caches(); gc.collect(); torch.cuda.empty_cache()
