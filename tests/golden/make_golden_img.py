#!/usr/bin/env python3
"""Golden fixture for the full-image API: tests/golden/tiny_render_img.npz.

Runs the reference's own `Renderer.render_img` (src/utils/Renderer.py:201-256) on the tiny scene of
make_golden.py with its 48 x 64 camera (3072 rays, chunked by ray_batch_size = 1000 so that the chunk loop and the
per-chunk depth maxima are exercised), stages colour and middle, both with gt_depth (the reference's render_img
reshapes gt_depth unconditionally, Renderer.py:231: None is not accepted).  Only build-container infrastructure: needs
/root/reference."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    cfg = MG.tiny_cfg()
    std = {'grid_coarse': 0.3, 'grid_middle': 0.3, 'grid_fine': 0.3, 'grid_color': 0.5}
    model, bound, c = MG.build_scene(cfg, seed=1234, grid_std=std)          # the scene of tiny_scene.npz
    s = dict(np.load(os.path.join(HERE, 'tiny_scene.npz')))
    for k in MG.GRID_KEYS:
        assert np.array_equal(c[k].numpy(), s[k]), k
    cam = dict(H=48, W=64, fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    slam_renderer = MG.make_renderer(cfg, bound, cam)
    slam_renderer.ray_batch_size = 1000
    g = torch.Generator().manual_seed(99)
    depth_img = torch.rand(cam['H'], cam['W'], generator=g) * 1.4 + 0.2
    depth_img[20:24, :] = 0.0
    th = 0.3
    c2w = torch.tensor([[np.cos(th), 0, np.sin(th), 0.1], [0, 1, 0, -0.05], [-np.sin(th), 0, np.cos(th), 0.2]],
                       dtype=torch.float32)
    out = dict(c2w=c2w.numpy(), depth_img=depth_img.numpy(), ray_batch_size=np.array(1000))
    d, u, col = slam_renderer.render_img(c, model, c2w, 'cpu', 'color', gt_depth=depth_img)
    out.update(color_depth=d.numpy(), color_unc=u.numpy(), color_color=col.numpy())
    d, u, col = slam_renderer.render_img(c, model, c2w, 'cpu', 'middle', gt_depth=depth_img)
    out.update(middle_depth=d.numpy(), middle_unc=u.numpy(), middle_color=col.numpy())
    np.savez_compressed(os.path.join(HERE, 'tiny_render_img.npz'), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == '__main__':
    main()
