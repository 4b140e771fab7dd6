#!/usr/bin/env python3
"""Golden fixture for hierarchical sampling: tests/golden/tiny_importance.npz.

Runs the reference's `Renderer.render_batch_ray` with `rendering.N_importance = 8` (the second pass of
Renderer.py:182-197: sample_pdf on the first pass's weights, sorted union of 32 + 16 + 8 = 56 distances per ray) on the
tiny scene of make_golden.py, stages colour and middle with gt_depth and stage coarse without (32 + 8 = 40 distances),
forward and backward (mapper loss / a cotangent for the coarse stage).  Only build-container infrastructure."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    cfg = MG.tiny_cfg()
    cfg['rendering']['N_importance'] = 8
    std = {'grid_coarse': 0.3, 'grid_middle': 0.3, 'grid_fine': 0.3, 'grid_color': 0.5}
    model, bound, c = MG.build_scene(cfg, seed=1234, grid_std=std)          # the scene of tiny_scene.npz
    s = dict(np.load(os.path.join(HERE, 'tiny_scene.npz')))
    for k in MG.GRID_KEYS:
        assert np.array_equal(c[k].numpy(), s[k]), k
    cam = dict(H=48, W=64, fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    renderer = MG.make_renderer(cfg, bound, cam)
    assert renderer.N_importance == 8 and renderer.perturb == 0
    ro, rd = torch.from_numpy(s['rays_o']), torch.from_numpy(s['rays_d'])
    gd, gc = torch.from_numpy(s['gt_depth']), torch.from_numpy(s['gt_color'])
    N = ro.shape[0]
    g = torch.Generator().manual_seed(21)
    cot = (torch.randn(N, generator=g).double(), torch.randn(N, generator=g).double(), torch.randn(N, 3, generator=g))
    out = {'N_importance': np.array(8)}
    for stage in ('color', 'middle', 'coarse'):
        res, _ = MG.run_stage(renderer, model, c, bound, ro, rd, None if stage == 'coarse' else gd, stage,
                              cot if stage == 'coarse' else None,
                              mapper_loss_gt=None if stage == 'coarse' else (gd, gc))
        for k, v in res.items():
            if k in ('pts', 'mask'):
                continue
            out[f'{stage}_{k}'] = v
        print(stage, 'z_vals', res['z_vals'].shape, 'loss', res['loss'])
    out['cot_depth'], out['cot_var'], out['cot_color'] = [t.numpy() for t in cot]
    np.savez_compressed(os.path.join(HERE, 'tiny_importance.npz'), **out)
    print('bytes', os.path.getsize(os.path.join(HERE, 'tiny_importance.npz')))


if __name__ == '__main__':
    main()
