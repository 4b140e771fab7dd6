#!/usr/bin/env python3
"""Golden fixture for the mapper inner-iteration glue: tests/golden/tiny_mapper_iters.npz.

Runs only in the build container (needs /root/reference).  The reference's Mapper cannot be imported here (cv2,
colorama, wandb, torchvision are absent), so this script executes the statements of Mapper.optimize_map that form
the glue -- src/Mapper.py:343-361 (val_grad = val[mask]), :396-419 (torch.optim.Adam, one group per grid),
:448-473 (re-materialisation, stage and learning-rate schedule), :548-575 (render, loss, backward, step) and
:594-602 (zero_grad, write-back) -- in that order, on the tiny scene of make_golden.py, with the REFERENCE's
Renderer, decoders and torch.optim.Adam.  The rays are a fixed batch (the per-iteration get_samples draw is the
caller's business); the frustum mask is a fixed box-shaped selection standing in for get_mask_from_c2w (:118-178,
numpy/cv2 code outside the path).

Every array written is an input chosen here or an output of those reference statements."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets up sys.path, the torchvision stub, the 'cuda:-1' shim, cwd)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.autograd import Variable  # noqa: E402

KEYS = ['grid_middle', 'grid_fine', 'grid_color']
NUM_JOINT_ITERS = 10


def make_masks(c):
    """bool [D,H,W] per grid: a slab of the volume plus a sprinkling of single voxels (like a frustum's ragged
    boundary); then the channel-repeated 5-D form the reference builds (Mapper.py:345-346)."""
    g = torch.Generator().manual_seed(4321)
    masks = {}
    for key in KEYS:
        D, H, W = c[key].shape[2:]
        m = torch.zeros(D, H, W, dtype=torch.bool)
        m[:, :, W // 3:] = True
        m |= torch.rand(D, H, W, generator=g) < 0.1
        m &= torch.rand(D, H, W, generator=g) < 0.95
        masks[key] = m
    return masks


def main():
    cfg = MG.tiny_cfg()
    std = {'grid_coarse': 0.3, 'grid_middle': 0.3, 'grid_fine': 0.3, 'grid_color': 0.5}
    model, bound, c = MG.build_scene(cfg, seed=1234, grid_std=std)          # identical to tiny_scene.npz
    s = np.load(os.path.join(HERE, 'tiny_scene.npz'))
    for k in MG.GRID_KEYS:
        assert np.array_equal(s[k], c[k].numpy()), k
    cam = dict(H=48, W=64, fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    renderer = MG.make_renderer(cfg, bound, cam)
    rays_o, rays_d = torch.from_numpy(s['rays_o']), torch.from_numpy(s['rays_d'])
    gt_depth, gt_color = torch.from_numpy(s['gt_depth']), torch.from_numpy(s['gt_color'])

    masks3 = make_masks(c)
    device = 'cpu'
    w_color_loss = cfg['mapping']['w_color_loss']
    lr_factor = cfg['mapping']['lr_factor']
    middle_iter_ratio, fine_iter_ratio = cfg['mapping']['middle_iter_ratio'], cfg['mapping']['fine_iter_ratio']

    # ---- Mapper.py:326-361
    decoders_para_list = []
    coarse_grid_para, middle_grid_para, fine_grid_para, color_grid_para = [], [], [], []
    masked_c_grad = {}
    for key, val in c.items():
        if 'coarse' in key:
            continue
        mask = masks3[key].unsqueeze(0).unsqueeze(0).repeat(1, val.shape[1], 1, 1, 1)
        val = val.to(device)
        val_grad = val[mask].clone()
        val_grad = Variable(val_grad.to(device), requires_grad=True)
        masked_c_grad[key] = val_grad
        masked_c_grad[key + 'mask'] = mask
        if key == 'grid_middle':
            middle_grid_para.append(val_grad)
        elif key == 'grid_fine':
            fine_grid_para.append(val_grad)
        elif key == 'grid_color':
            color_grid_para.append(val_grad)
    decoders_para_list += list(model.color_decoder.parameters())           # fix_fine: True, fix_color: False
    # ---- :409-413
    optimizer = torch.optim.Adam([{'params': decoders_para_list, 'lr': 0},
                                  {'params': coarse_grid_para, 'lr': 0},
                                  {'params': middle_grid_para, 'lr': 0},
                                  {'params': fine_grid_para, 'lr': 0},
                                  {'params': color_grid_para, 'lr': 0}])
    out = {'num_joint_iters': np.array(NUM_JOINT_ITERS), 'w_color_loss': np.array(w_color_loss),
           'lr_factor': np.array(lr_factor)}
    for key in KEYS:
        out['mask_' + key] = masks3[key].numpy()
    losses, stages, lrs = [], [], []
    for joint_iter in range(NUM_JOINT_ITERS):
        # ---- :448-458
        for key, val in c.items():
            if 'coarse' not in key:
                val_grad = masked_c_grad[key]
                mask = masked_c_grad[key + 'mask']
                val = val.to(device)
                val[mask] = val_grad
                c[key] = val
        # ---- :460-473
        if joint_iter <= int(NUM_JOINT_ITERS * middle_iter_ratio):
            stage = 'middle'
        elif joint_iter <= int(NUM_JOINT_ITERS * fine_iter_ratio):
            stage = 'fine'
        else:
            stage = 'color'
        st = cfg['mapping']['stage'][stage]
        optimizer.param_groups[0]['lr'] = st['decoders_lr'] * lr_factor
        optimizer.param_groups[1]['lr'] = st['coarse_lr'] * lr_factor
        optimizer.param_groups[2]['lr'] = st['middle_lr'] * lr_factor
        optimizer.param_groups[3]['lr'] = st['fine_lr'] * lr_factor
        optimizer.param_groups[4]['lr'] = st['color_lr'] * lr_factor
        optimizer.zero_grad()
        # ---- :548-575
        depth, uncertainty, color = renderer.render_batch_ray(c, model, rays_d, rays_o, device, stage, gt_depth=gt_depth)
        depth_mask = (gt_depth > 0)
        loss = torch.abs(gt_depth[depth_mask] - depth[depth_mask]).sum()
        if stage == 'color':
            color_loss = torch.abs(gt_color - color).sum()
            loss += w_color_loss * color_loss
        loss.backward(retain_graph=False)
        optimizer.step()
        losses.append(loss.item())
        stages.append(stage)
        lrs.append([st['decoders_lr'], st['middle_lr'], st['fine_lr'], st['color_lr']])
        # ---- :594-602
        optimizer.zero_grad()
        for key, val in c.items():
            if 'coarse' not in key:
                val_grad = masked_c_grad[key]
                mask = masked_c_grad[key + 'mask']
                val = val.detach()
                val[mask] = val_grad.clone().detach()
                c[key] = val
        if joint_iter in (0, 4, 6):                                     # intermediate snapshots of one grid
            out[f'grid_middle_after_{joint_iter}'] = c['grid_middle'].numpy().copy()
        print(joint_iter, stage, 'loss', losses[-1])
    out['losses'] = np.array(losses)
    out['stages'] = np.array(stages)
    out['lrs'] = np.array(lrs) * lr_factor
    for key in KEYS:
        out['final_' + key] = c[key].numpy().copy()
    for name, p in model.color_decoder.named_parameters():
        out['final_cd_' + name] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'tiny_mapper_iters.npz'), **out)
    print('bytes', os.path.getsize(os.path.join(HERE, 'tiny_mapper_iters.npz')))


if __name__ == '__main__':
    main()
