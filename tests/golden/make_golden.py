#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, CPU torch).  Nothing
under tests/, bench.py or smoke() imports this file; they read the .npz data.

Recipe:
  * cwd = /root/reference, sys.path[0] = /root/reference
  * empty stub modules for `torchvision` (only Renderer.render_img_rescale uses it)
  * Tensor.to shim: the reference builds the literal device string
    f'cuda:{p.get_device()}' (decoder.py:316) which is 'cuda:-1' on CPU.
  * decoders: reference `config.get_model(cfg)` with torch.manual_seed, biases
    and fc_c perturbed so that no term is identically zero (pretrained weights
    are not in the tree).
  * grids: shapes from the reference expressions (EvenNICER_SLAM.py:236-273).

Every array written here is either an input chosen by this script or an output
of a reference function called on those inputs.  The only derived quantities
are the voxel base index / fractions, which are computed from the reference's
own normalize_3d_coordinate output with the ATen grid_sampler formulas
(torch/include/ATen/native/GridSampler.h) in float32 torch ops.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.path.insert(0, REF)

_tv = types.ModuleType('torchvision')
_tvt = types.ModuleType('torchvision.transforms')
_tv.transforms = _tvt
sys.modules['torchvision'] = _tv
sys.modules['torchvision.transforms'] = _tvt

import numpy as np  # noqa: E402
import torch  # noqa: E402

_orig_to = torch.Tensor.to


def _to(self, *a, **k):
    a = tuple('cpu' if (isinstance(x, str) and x == 'cuda:-1') else x for x in a)
    return _orig_to(self, *a, **k)


torch.Tensor.to = _to
os.chdir(REF)

from src import config  # noqa: E402
import src.utils.Renderer as ref_renderer_mod  # noqa: E402
from src.utils.Renderer import Renderer  # noqa: E402
from src import common as ref_common  # noqa: E402

torch.set_num_threads(8)
STAGES = ['coarse', 'middle', 'fine', 'color']
GRID_KEYS = ['grid_coarse', 'grid_middle', 'grid_fine', 'grid_color']


# ----------------------------------------------------------------------------- scene set-up
def ref_bound(cfg):
    """EvenNICER_SLAM.load_bound (EvenNICER_SLAM.py:170-175), same expressions."""
    scale = cfg['scale']
    bound = torch.from_numpy(np.array(cfg['mapping']['bound']) * scale)
    bound_divisible = cfg['grid_len']['bound_divisible']
    bound[:, 1] = (((bound[:, 1] - bound[:, 0]) / bound_divisible).int() + 1) * bound_divisible + bound[:, 0]
    return bound


def ref_grid_shapes(cfg, bound):
    """EvenNICER_SLAM.grid_init shapes (EvenNICER_SLAM.py:236-273)."""
    xyz_len = bound[:, 1] - bound[:, 0]
    enlarge = cfg['model']['coarse_bound_enlarge']
    out = {}
    for key in ['coarse', 'middle', 'fine', 'color']:
        gl = cfg['grid_len'][key]
        if key == 'coarse':
            shp = list(map(int, (xyz_len * enlarge / gl).tolist()))
        else:
            shp = list(map(int, (xyz_len / gl).tolist()))
        shp[0], shp[2] = shp[2], shp[0]
        out['grid_' + key] = shp
    return out


def build_scene(cfg, seed, grid_std):
    torch.manual_seed(seed)
    model = config.get_model(cfg)
    # make every bias non-zero (DenseLayer zero-inits them, decoder.py:78-79)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('bias'):
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    bound = ref_bound(cfg)
    enlarge = cfg['model']['coarse_bound_enlarge']
    model.bound = bound
    model.middle_decoder.bound = bound
    model.fine_decoder.bound = bound
    model.color_decoder.bound = bound
    model.coarse_decoder.bound = bound * enlarge
    shapes = ref_grid_shapes(cfg, bound)
    c = {}
    g = torch.Generator().manual_seed(seed + 2)
    for key in GRID_KEYS:
        c[key] = torch.zeros([1, cfg['model']['c_dim'], *shapes[key]]).normal_(
            mean=0, std=grid_std[key], generator=g)
    return model, bound, c


def make_renderer(cfg, bound, cam):
    slam = types.SimpleNamespace(nice=True, bound=bound, **cam)
    return Renderer(cfg, None, slam)


# ----------------------------------------------------------------------------- capture helpers
class Capture:
    """Wrap the reference's eval_points / raw2outputs to record intermediates."""

    def __init__(self, renderer):
        self.r = renderer
        self.rec = {}
        self._orig_eval = renderer.eval_points
        self._orig_raw2 = ref_renderer_mod.raw2outputs_nerf_color

    def __enter__(self):
        cap = self

        def eval_points(p, decoders, c=None, stage='color', device='cuda:0'):
            cap.rec['pts'] = p.detach().clone()
            ret = cap._orig_eval(p, decoders, c, stage, device)
            cap.rec['raw'] = ret.detach().clone()
            return ret

        def raw2(raw, z_vals, rays_d, occupancy=False, device='cuda:0'):
            cap.rec['z_vals'] = z_vals.detach().clone()
            out = cap._orig_raw2(raw, z_vals, rays_d, occupancy=occupancy, device=device)
            cap.rec['weights'] = out[3].detach().clone()
            return out

        self.r.eval_points = eval_points
        ref_renderer_mod.raw2outputs_nerf_color = raw2
        return self

    def __exit__(self, *a):
        self.r.eval_points = self._orig_eval
        ref_renderer_mod.raw2outputs_nerf_color = self._orig_raw2


def voxel_index(pts, bound, shape):
    """Base voxel index and fractions: normalise with the reference function,
    then ATen's unnormalize/clip/floor (align_corners=True, border) in float32."""
    p_nor = ref_common.normalize_3d_coordinate(pts.clone(), bound).float()
    D, H, W = shape
    res = {}
    for ax, size, name in ((0, W, 'x'), (1, H, 'y'), (2, D, 'z')):
        co = ((p_nor[:, ax] + 1) / 2) * (size - 1)
        co = torch.clamp(co, min=0.0)
        co = torch.clamp(co, max=float(size - 1))
        fl = torch.floor(co)
        res['i' + name] = fl.to(torch.int32).numpy()
        res['f' + name] = (co - fl).numpy()
    return res


def in_bound_mask(pts, bound):
    m = torch.ones(pts.shape[0], dtype=torch.bool)
    for a in range(3):
        m &= (pts[:, a] < bound[a][1]) & (pts[:, a] > bound[a][0])
    return m


def run_stage(renderer, model, c, bound, rays_o, rays_d, gt_depth, stage, cot, mapper_loss_gt=None):
    """One reference render_batch_ray forward + backward.  Returns dict of arrays."""
    for p in model.parameters():
        p.grad = None
    cg = {k: v.clone().requires_grad_(True) for k, v in c.items()}
    ro = rays_o.clone().requires_grad_(True)
    rd = rays_d.clone().requires_grad_(True)
    with Capture(renderer) as cap:
        depth, var, color = renderer.render_batch_ray(cg, model, rd, ro, 'cpu', stage, gt_depth=gt_depth)
    out = {
        'depth': depth.detach().numpy(), 'var': var.detach().numpy(), 'color': color.detach().numpy(),
        'z_vals': cap.rec['z_vals'].numpy(), 'pts': cap.rec['pts'].numpy(),
        'raw': cap.rec['raw'].numpy(), 'weights': cap.rec['weights'].numpy(),
        'mask': in_bound_mask(cap.rec['pts'], bound).numpy(),
    }
    if mapper_loss_gt is None:
        gd, gv, gc = cot
        loss = (depth * gd).sum() + (var * gv).sum() + (color * gc).sum()
    else:
        gt_d, gt_c = mapper_loss_gt      # Mapper.py:553-562 loss
        m = gt_d > 0
        loss = torch.abs(gt_d[m] - depth[m]).sum()
        if stage == 'color':
            loss = loss + 0.2 * torch.abs(gt_c - color).sum()
    loss.backward()
    out['loss'] = np.array(loss.item())
    out['g_rays_o'] = ro.grad.numpy() if ro.grad is not None else np.zeros_like(rays_o.numpy())
    out['g_rays_d'] = rd.grad.numpy() if rd.grad is not None else np.zeros_like(rays_d.numpy())
    for k, v in cg.items():
        if v.grad is not None:
            out['g_' + k] = v.grad.numpy()
    for name, p in model.named_parameters():
        if p.grad is not None:
            out['gp_' + name] = p.grad.detach().numpy().copy()
    return out, cap.rec['pts']


def state_arrays(model):
    return {'sd_' + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


# ----------------------------------------------------------------------------- fixtures
def tiny_cfg():
    cfg = config.load_config('configs/Replica/room0.yaml', 'configs/nice_slam.yaml')
    cfg['mapping']['bound'] = [[-1.0, 1.1], [-0.9, 0.8], [-0.7, 0.6]]
    cfg['grid_len'].update({'coarse': 0.8, 'middle': 0.4, 'fine': 0.2, 'color': 0.2})
    return cfg


def make_tiny():
    cfg = tiny_cfg()
    std = {k: 0.01 for k in GRID_KEYS}
    std['grid_coarse'] = 0.3
    std['grid_middle'] = 0.3
    std['grid_fine'] = 0.3
    std['grid_color'] = 0.5
    model, bound, c = build_scene(cfg, seed=1234, grid_std=std)
    cam = dict(H=48, W=64, fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    renderer = make_renderer(cfg, bound, cam)

    g = torch.Generator().manual_seed(99)
    depth_img = torch.rand(cam['H'], cam['W'], generator=g) * 1.4 + 0.2
    depth_img[20:24, :] = 0.0                          # band of zero depth
    color_img = torch.rand(cam['H'], cam['W'], 3, generator=g)
    # camera: small rotation about y, inside the bound
    th = 0.3
    c2w = torch.tensor([[np.cos(th), 0, np.sin(th), 0.1],
                        [0, 1, 0, -0.05],
                        [-np.sin(th), 0, np.cos(th), 0.2]], dtype=torch.float32)
    torch.manual_seed(7)
    ro, rd, gd, gc = ref_common.get_samples(0, cam['H'], 0, cam['W'], 56, cam['H'], cam['W'],
                                            cam['fx'], cam['fy'], cam['cx'], cam['cy'],
                                            c2w, depth_img, color_img, 'cpu')
    # hand-made rays: axis-parallel directions, an origin near a wall (early exit),
    # depth beyond the scene, zero depth.
    extra_o = torch.tensor([[0.1, -0.05, 0.2], [0.1, -0.05, 0.2], [0.9, 0.6, 0.4], [0.9, 0.6, 0.4],
                            [0.1, -0.05, 0.2], [-0.8, -0.7, -0.5], [0.1, -0.05, 0.2], [0.0, 0.0, 0.0]])
    extra_d = torch.tensor([[0.0, 0.0, -1.0], [1.0, 0.0, 0.0], [0.3, 0.2, 0.1], [1.0, 1.0, 1.0],
                            [0.0, -1.0, 0.0], [-0.2, -0.3, -1.0], [0.5, 0.5, -1.0], [-0.0, 0.0, -1.0]])
    extra_gd = torch.tensor([0.5, 0.7, 0.6, 0.0, 3.0, 1.0, 0.0, 0.4])
    extra_gc = torch.rand(8, 3, generator=g)
    rays_o = torch.cat([ro.float(), extra_o]).contiguous()
    rays_d = torch.cat([rd.float(), extra_d]).contiguous()
    gt_depth = torch.cat([gd.float(), extra_gd]).contiguous()
    gt_color = torch.cat([gc.float(), extra_gc]).contiguous()
    assert (gt_depth == 0).sum() >= 4
    N = rays_o.shape[0]

    scene = dict(bound=bound.numpy(), rays_o=rays_o.numpy(), rays_d=rays_d.numpy(),
                 gt_depth=gt_depth.numpy(), gt_color=gt_color.numpy(),
                 cam=np.array([cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy']]),
                 coarse_bound_enlarge=np.array(cfg['model']['coarse_bound_enlarge']),
                 N_samples=np.array(cfg['rendering']['N_samples']),
                 N_surface=np.array(cfg['rendering']['N_surface']),
                 map_bound_cfg=np.array(cfg['mapping']['bound']),
                 grid_len=np.array([cfg['grid_len'][k] for k in ['coarse', 'middle', 'fine', 'color']]),
                 bound_divisible=np.array(cfg['grid_len']['bound_divisible']))
    scene.update({k: v.numpy() for k, v in c.items()})
    scene.update(state_arrays(model))
    np.savez(os.path.join(HERE, 'tiny_scene.npz'), **scene)

    g = torch.Generator().manual_seed(5)
    cot = (torch.randn(N, generator=g).double(), torch.randn(N, generator=g).double(),
           torch.randn(N, 3, generator=g))
    for stage in STAGES:
        out, pts = run_stage(renderer, model, c, bound, rays_o, rays_d, gt_depth, stage, cot)
        out['cot_depth'], out['cot_var'], out['cot_color'] = [t.numpy() for t in cot]
        # voxel indices for the grids the stage reads
        used = {'coarse': ['grid_coarse'], 'middle': ['grid_middle'], 'fine': ['grid_fine', 'grid_middle'],
                'color': ['grid_fine', 'grid_middle', 'grid_color']}[stage]
        for key in used:
            b = bound * cfg['model']['coarse_bound_enlarge'] if key == 'grid_coarse' else bound
            vi = voxel_index(pts, b, c[key].shape[2:])
            for kk, vv in vi.items():
                out[f'vox_{key}_{kk}'] = vv
        np.savez(os.path.join(HERE, f'tiny_{stage}.npz'), **out)
        print('tiny', stage, 'depth[:3]', out['depth'][:3], 'loss', out['loss'])
    # mapper-style loss on the colour stage (Mapper.py:553-562)
    out, _ = run_stage(renderer, model, c, bound, rays_o, rays_d, gt_depth, 'color', None,
                       mapper_loss_gt=(gt_depth, gt_color))
    keep = {k: v for k, v in out.items() if k.startswith('g') or k in ('loss', 'depth', 'color')}
    np.savez(os.path.join(HERE, 'tiny_color_mapperloss.npz'), **keep)

    # eval_points fixture (a3): arbitrary points incl. out-of-bound ones
    g = torch.Generator().manual_seed(11)
    p = (torch.rand(200, 3, generator=g).double() - 0.5) * torch.tensor([2.6, 2.2, 1.8]).double()
    p[0] = torch.tensor([bound[0, 0], 0.0, 0.0])       # exactly on the lower face -> masked
    p[1] = torch.tensor([bound[0, 1], 0.0, 0.0])       # exactly on the upper face -> masked
    ev = {'p': p.numpy()}
    with torch.no_grad():
        for stage in STAGES:
            ev['raw_' + stage] = renderer.eval_points(p, model, c, stage, 'cpu').numpy()
    np.savez(os.path.join(HERE, 'tiny_eval_points.npz'), **ev)
    return cfg, model, bound, c, renderer


def make_room0():
    """BASELINE configs[0] (coarse, 200x32) and configs[1] (colour, 1000x48) on room0 shapes."""
    cfg = config.load_config('configs/Replica/room0.yaml', 'configs/nice_slam.yaml')
    std = {'grid_coarse': 0.01, 'grid_middle': 0.01, 'grid_fine': 0.0001, 'grid_color': 0.01}
    torch.manual_seed(0)
    model = config.get_model(cfg)
    bound = ref_bound(cfg)
    enlarge = cfg['model']['coarse_bound_enlarge']
    model.bound = bound
    for d in (model.middle_decoder, model.fine_decoder, model.color_decoder):
        d.bound = bound
    model.coarse_decoder.bound = bound * enlarge
    shapes = ref_grid_shapes(cfg, bound)
    c = {}
    for key in GRID_KEYS:       # same RNG stream order as bench.py / tests regenerate it
        c[key] = torch.zeros([1, 32, *shapes[key]]).normal_(mean=0, std=std[key])
    cam = dict(H=680, W=1200, fx=600.0, fy=600.0, cx=599.5, cy=339.5)
    renderer = make_renderer(cfg, bound, cam)
    depth_img = torch.rand(cam['H'], cam['W']) * 3.0 + 0.5
    depth_img[300:340, :] = 0.0          # 5.9 % zeros
    color_img = torch.rand(cam['H'], cam['W'], 3)
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 3] = torch.tensor([3.0, 1.0, 0.0])
    meta = dict(bound=bound.numpy(), shapes=np.array([shapes[k] for k in GRID_KEYS]),
                grid_checksum=np.array([c[k].double().sum().item() for k in GRID_KEYS]),
                grid_abs_checksum=np.array([c[k].double().abs().sum().item() for k in GRID_KEYS]))

    # ---- config 0: coarse, 200 rays x 32 samples
    ro, rd, gd, gc = ref_common.get_samples(0, cam['H'], 0, cam['W'], 200, cam['H'], cam['W'], cam['fx'],
                                            cam['fy'], cam['cx'], cam['cy'], c2w, depth_img, color_img, 'cpu')
    ro, rd, gd, gc = ro.float().contiguous(), rd.float().contiguous(), gd.float(), gc.float()
    out, _ = run_stage(renderer, model, c, bound, ro, rd, None, 'coarse', None, mapper_loss_gt=(gd, gc))
    keep = {k: v for k, v in out.items() if k not in ('pts', 'raw', 'weights', 'mask')}
    keep.update(rays_o=ro.numpy(), rays_d=rd.numpy(), gt_depth=gd.numpy(), gt_color=gc.numpy(),
                grid_coarse=c['grid_coarse'].numpy(), **meta)
    keep.update(state_arrays(model))
    np.savez(os.path.join(HERE, 'room0_coarse200.npz'), **keep)
    print('room0 coarse200 loss', out['loss'])

    # ---- config 1: colour, 1000 rays x 48 samples (grids regenerated from the seed by consumers)
    ro, rd, gd, gc = ref_common.get_samples(0, cam['H'], 0, cam['W'], 1000, cam['H'], cam['W'], cam['fx'],
                                            cam['fy'], cam['cx'], cam['cy'], c2w, depth_img, color_img, 'cpu')
    ro, rd, gd, gc = ro.float().contiguous(), rd.float().contiguous(), gd.float(), gc.float()
    out, pts = run_stage(renderer, model, c, bound, ro, rd, gd, 'color', None, mapper_loss_gt=(gd, gc))
    keep = {k: out[k] for k in ('depth', 'var', 'color', 'z_vals', 'loss', 'g_rays_o', 'g_rays_d')}
    for k, v in out.items():
        if k.startswith('gp_'):
            keep[k] = v
    # (round 4) point-level parity at the headline scene's size as well, as make_golden_scenes.py keeps it for office0 / recording4:
    # decoder outputs and bound mask of the first 256 rays, their samples' voxel indices / fractions in the three grids
    keep['raw_256'] = out['raw'].reshape(1000, -1, 4)[:256].copy()
    keep['mask_256'] = out['mask'].reshape(1000, -1)[:256].copy()
    p256 = pts[:256 * out['z_vals'].shape[1]]
    for key in ('grid_middle', 'grid_fine', 'grid_color'):
        for kk, vv in voxel_index(p256, bound, c[key].shape[2:]).items():
            keep[f'vox_{key}_{kk}'] = vv
    rng = np.random.RandomState(0)
    for key in ('grid_middle', 'grid_fine', 'grid_color'):
        gg = out['g_' + key].reshape(-1)
        nz = np.flatnonzero(gg)
        keep[f'gstat_{key}'] = np.array([gg.astype(np.float64).sum(), np.abs(gg).astype(np.float64).sum(),
                                         float(nz.size), float(gg.size)])
        pick = rng.choice(nz, size=min(2000, nz.size), replace=False)
        zero_pick = rng.choice(np.setdiff1d(np.arange(gg.size)[:200000], nz), size=200, replace=False)
        pick = np.concatenate([pick, zero_pick])
        keep[f'gidx_{key}'] = pick.astype(np.int64)
        keep[f'gval_{key}'] = gg[pick]
    keep.update(rays_o=ro.numpy(), rays_d=rd.numpy(), gt_depth=gd.numpy(), gt_color=gc.numpy(), **meta)
    np.savez_compressed(os.path.join(HERE, 'room0_color1000.npz'), **keep)
    print('room0 color1000 loss', out['loss'],
          'nonzero frac', [keep[f'gstat_{k}'][2] / keep[f'gstat_{k}'][3] for k in ('grid_middle', 'grid_fine', 'grid_color')])


def make_ray_fixtures():
    """a1: get_samples (RNG + ray generation); a10: get_rays_rescale; pose gradient through get_samples."""
    H, W, fx, fy, cx, cy = 68, 120, 60.0, 60.0, 59.5, 33.5
    g = torch.Generator().manual_seed(3)
    depth = torch.rand(H, W, generator=g) * 3
    color = torch.rand(H, W, 3, generator=g).double()
    c2w = torch.tensor([[0.8, -0.6, 0.0, 1.0], [0.6, 0.8, 0.0, 2.0], [0.0, 0.0, 1.0, 3.0], [0, 0, 0, 1.0]])
    c2w.requires_grad_(True)
    out = {}
    for tag, (H0, H1, W0, W1, n) in {'full': (0, H, 0, W, 100), 'edge': (10, H - 10, 20, W - 20, 37)}.items():
        torch.manual_seed(42)
        ro, rd, sd, sc = ref_common.get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2w, depth, color, 'cpu')
        # recover the indices the reference drew (same generator state)
        torch.manual_seed(42)
        idx = torch.randint((H1 - H0) * (W1 - W0), (n,))
        out[f'{tag}_args'] = np.array([H0, H1, W0, W1, n])
        out[f'{tag}_idx'] = idx.numpy()
        out[f'{tag}_rays_o'] = ro.detach().numpy()
        out[f'{tag}_rays_d'] = rd.detach().numpy()
        out[f'{tag}_depth'] = sd.numpy()
        out[f'{tag}_color'] = sc.numpy()
    g = torch.Generator().manual_seed(8)
    cot_o = torch.randn(100, 3, generator=g)
    cot_d = torch.randn(100, 3, generator=g)
    torch.manual_seed(42)
    ro, rd, _, _ = ref_common.get_samples(0, H, 0, W, 100, H, W, fx, fy, cx, cy, c2w, depth, color, 'cpu')
    ((ro * cot_o).sum() + (rd * cot_d).sum()).backward()
    out.update(cam=np.array([H, W, fx, fy, cx, cy]), depth_img=depth.numpy(), color_img=color.numpy(),
               c2w=c2w.detach().numpy(), seed=np.array(42), cot_o=cot_o.numpy(), cot_d=cot_d.numpy(),
               g_c2w=c2w.grad.numpy())
    with torch.no_grad():
        ro, rd = ref_common.get_rays_rescale(H, W, int(H * 0.15), int(W * 0.15), fx, fy, cx, cy, c2w, 'cpu')
        out['rescale_rays_o'] = ro.numpy()
        out['rescale_rays_d'] = rd.numpy()
        ro, rd = ref_common.get_rays(H, W, fx, fy, cx, cy, c2w, 'cpu')
        out['img_rays_o'] = ro.numpy()[::7, ::11]
        out['img_rays_d'] = rd.numpy()[::7, ::11]
    np.savez(os.path.join(HERE, 'rays.npz'), **out)


def make_bounds():
    """Rounded bounds + grid shapes for the three scenes BASELINE.json names."""
    out = {}
    for scene in ['configs/Replica/room0.yaml', 'configs/Replica/office0.yaml', 'configs/rpg/recording4.yaml']:
        cfg = config.load_config(scene, 'configs/nice_slam.yaml')
        b = ref_bound(cfg)
        shp = ref_grid_shapes(cfg, b)
        tag = os.path.basename(scene)[:-5]
        out[tag + '_cfg_bound'] = np.array(cfg['mapping']['bound'])
        out[tag + '_scale'] = np.array(cfg['scale'])
        out[tag + '_bound'] = b.numpy()
        out[tag + '_shapes'] = np.array([shp[k] for k in GRID_KEYS])
    np.savez(os.path.join(HERE, 'bounds.npz'), **out)
    print({k: v.tolist() for k, v in out.items() if k.endswith('shapes')})


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'room0':       # only the two room0 fixtures (self-seeded: torch.manual_seed(0) inside)
        make_room0()
        sys.exit(0)
    make_bounds()
    make_ray_fixtures()
    make_tiny()
    make_room0()
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith('.npz'))
    print('total fixture bytes', tot)
