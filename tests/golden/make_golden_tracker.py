#!/usr/bin/env python3
"""Golden fixture for the tracker's camera iteration (RGB-D part): tests/golden/tiny_tracker_iter.npz.

Runs only in the build container (needs /root/reference).  src/Tracker.py cannot be imported here (cv2, colorama,
wandb, torchvision are absent), so this script executes the statements of Tracker.optimize_cam_in_batch that touch
the path -- :141 get_camera_from_tensor, :161-162 get_samples, :164-174 the in-bound prefilter, :175-179
render_batch_ray with the uncertainty detached, :184 the depth mask (handle_dynamic off), :187-195 the
uncertainty-weighted depth loss and the colour loss, :197 backward -- with the REFERENCE's functions
(src/common.py get_camera_from_tensor / get_samples, src/utils/Renderer.py, the decoders) on the tiny scene.

One extra shim beyond make_golden.py's: quad2rotation (common.py:201) moves a fresh tensor `.to(quad.get_device())`,
which is -1 on CPU; the integer -1 is mapped to 'cpu' the same way the 'cuda:-1' string is."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

_prev_to = torch.Tensor.to


def _to(self, *a, **k):
    a = tuple('cpu' if (isinstance(x, int) and not isinstance(x, bool) and x == -1) else x for x in a)
    return _prev_to(self, *a, **k)


torch.Tensor.to = _to
from src.common import get_camera_from_tensor, get_samples  # noqa: E402


def main():
    cfg = MG.tiny_cfg()
    std = {'grid_coarse': 0.3, 'grid_middle': 0.3, 'grid_fine': 0.3, 'grid_color': 0.5}
    model, bound, c = MG.build_scene(cfg, seed=1234, grid_std=std)          # identical to tiny_scene.npz
    s = np.load(os.path.join(HERE, 'tiny_scene.npz'))
    for k in MG.GRID_KEYS:
        assert np.array_equal(s[k], c[k].numpy()), k
    cam = dict(H=48, W=64, fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    H, W, fx, fy, cx, cy = (cam[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
    renderer = MG.make_renderer(cfg, bound, cam)
    g = torch.Generator().manual_seed(99)
    gt_depth = torch.rand(H, W, generator=g) * 1.4 + 0.2
    gt_depth[20:24, :] = 0.0
    gt_color = torch.rand(H, W, 3, generator=g)
    device = 'cpu'
    w_color_loss = cfg['tracking']['w_color_loss']
    Hedge, Wedge = 4, 6                                                     # ignore_edge_H / _W
    batch_size = 80
    # an unnormalised quaternion near a small rotation about y, translation inside the bound
    camera_tensor = torch.tensor([0.98, 0.02, 0.17, -0.03, 0.1, -0.05, 0.2], requires_grad=True)

    out = {'camera_tensor': camera_tensor.detach().numpy().copy(), 'gt_depth': gt_depth.numpy(),
           'gt_color': gt_color.numpy(), 'cam': np.array([H, W, fx, fy, cx, cy]), 'edge': np.array([Hedge, Wedge]),
           'batch_size': np.array(batch_size), 'w_color_loss': np.array(w_color_loss), 'seed': np.array(31)}
    # ---- Tracker.py:141
    c2w = get_camera_from_tensor(camera_tensor)
    out['c2w'] = c2w.detach().numpy().copy()
    # the pixel draw get_samples is about to make (select_uv, common.py:99: the only RNG call), recorded so that
    # other devices can be fed the identical indices
    torch.manual_seed(31)
    idx = torch.randint((H - 2 * Hedge) * (W - 2 * Wedge), (batch_size,))
    out['idx'] = idx.numpy()
    # ---- :161-162
    torch.manual_seed(31)
    batch_rays_o, batch_rays_d, batch_gt_depth, batch_gt_color = get_samples(
        Hedge, H - Hedge, Wedge, W - Wedge, batch_size, H, W, fx, fy, cx, cy, c2w, gt_depth, gt_color, device)
    wi = (Wedge + idx % (W - 2 * Wedge)).float()
    wj = (Hedge + idx // (W - 2 * Wedge)).float()
    dirs = torch.stack([(wi - cx) / fx, -(wj - cy) / fy, -torch.ones_like(wi)], -1)
    assert torch.equal((dirs[:, None, :] * c2w[:3, :3]).sum(-1), batch_rays_d)          # the recorded idx is the draw
    assert torch.equal(gt_depth.reshape(-1)[(wj.long() * W + wi.long())], batch_gt_depth)
    out['rays_o_all'] = batch_rays_o.detach().numpy().copy()
    out['rays_d_all'] = batch_rays_d.detach().numpy().copy()
    # ---- :164-174
    with torch.no_grad():
        det_rays_o = batch_rays_o.clone().detach().unsqueeze(-1)
        det_rays_d = batch_rays_d.clone().detach().unsqueeze(-1)
        t = (bound.unsqueeze(0).to(device) - det_rays_o) / det_rays_d
        t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
        inside_mask = t >= batch_gt_depth
    out['inside_mask'] = inside_mask.numpy()
    batch_rays_d = batch_rays_d[inside_mask]
    batch_rays_o = batch_rays_o[inside_mask]
    batch_gt_depth = batch_gt_depth[inside_mask]
    batch_gt_color = batch_gt_color[inside_mask]
    # ---- :175-179
    depth, uncertainty, color = renderer.render_batch_ray(c, model, batch_rays_d, batch_rays_o, device, stage='color',
                                                          gt_depth=batch_gt_depth)
    uncertainty = uncertainty.detach()
    mask = batch_gt_depth > 0                                               # :184 (handle_dynamic off)
    # ---- :187-195
    loss = (torch.abs(batch_gt_depth - depth) / torch.sqrt(uncertainty + 1e-10))[mask].sum()
    color_loss = torch.abs(batch_gt_color - color)[mask].sum()
    loss = loss + w_color_loss * color_loss
    loss.backward()                                                         # :197
    out.update(depth=depth.detach().numpy(), uncertainty=uncertainty.numpy(), color=color.detach().numpy(),
               loss=np.array(loss.item()), g_camera_tensor=camera_tensor.grad.numpy().copy(),
               batch_gt_depth=batch_gt_depth.numpy(), batch_gt_color=batch_gt_color.numpy())
    np.savez_compressed(os.path.join(HERE, 'tiny_tracker_iter.npz'), **out)
    print('loss', loss.item(), 'grad', camera_tensor.grad.numpy(), 'rays', int(inside_mask.sum()), 'of', batch_size)


if __name__ == '__main__':
    main()
