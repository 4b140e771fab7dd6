#!/usr/bin/env python3
"""Golden fixture for local bundle adjustment in the mapper glue: tests/golden/tiny_mapper_ba.npz.

Runs only in the build container (needs /root/reference).  src/Mapper.py cannot be imported here (cv2, colorama, wandb,
torchvision are absent), so -- like make_golden_mapper.py -- this script executes the statements of Mapper.optimize_map
that form the glue, now with `BA: True`: :343-361 (val_grad = val[mask]), :374-390 (one camera tensor per optimised
frame, the oldest keyframe fixed), :396-407 (torch.optim.Adam with the camera group), :448-490 (re-materialisation,
stage and learning-rate schedule incl. BA_cam_lr in the colour stage), :502-535 (per frame: get_camera_from_tensor,
get_samples; concatenation), :537-547 (the in-bound prefilter), :548-575 (render, loss, backward, step) and :594-602 --
with the REFERENCE's get_camera_from_tensor / get_samples / Renderer / decoders / torch.optim.Adam on the tiny scene:
two keyframes (the first one fixed) and the current frame, 24 pixels each, 8 joint iterations.  The camera tensors are
inputs chosen here (get_tensor_from_camera needs mathutils, absent); the frustum mask is the fixed selection of
make_golden_mapper.py.  The pixel draws of every get_samples call are recorded so that another device can be fed the
identical indices."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_tracker as MT  # noqa: E402  (make_golden's set-up + the integer -1 device shim; defines get_camera_from_tensor, get_samples)
import make_golden_mapper as MM  # noqa: E402
MG = MT.MG

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.autograd import Variable  # noqa: E402

get_camera_from_tensor, get_samples = MT.get_camera_from_tensor, MT.get_samples
KEYS = MM.KEYS
NUM_JOINT_ITERS = 8
PIXELS = 72                                                                  # mapping.pixels: 24 per frame


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
    device = 'cpu'
    w_color_loss = cfg['mapping']['w_color_loss']
    lr_factor = cfg['mapping']['lr_factor']
    BA_cam_lr = cfg['mapping']['BA_cam_lr']
    middle_iter_ratio, fine_iter_ratio = cfg['mapping']['middle_iter_ratio'], cfg['mapping']['fine_iter_ratio']

    # three frames: keyframes 0 (oldest, fixed) and 1, and the current frame (-1)
    g = torch.Generator().manual_seed(77)
    frames = []
    tensors = [torch.tensor([0.98, 0.02, 0.17, -0.03, 0.1, -0.05, 0.2]),
               torch.tensor([0.97, -0.04, 0.21, 0.02, 0.05, 0.02, 0.15]),
               torch.tensor([0.99, 0.05, 0.10, 0.01, 0.15, -0.02, 0.25])]
    for f in range(3):
        depth = torch.rand(H, W, generator=g) * 1.4 + 0.2
        depth[10 + 5 * f:13 + 5 * f, :] = 0.0
        color = torch.rand(H, W, 3, generator=g)
        frames.append(dict(depth=depth, color=color, est_c2w=get_camera_from_tensor(tensors[f]).detach()))
    keyframe_dict = frames[:2]
    cur_gt_depth, cur_gt_color, cur_c2w = frames[2]['depth'], frames[2]['color'], frames[2]['est_c2w']
    optimize_frame = [0, 1, -1]
    oldest_frame = min(optimize_frame[:-1])
    pixs_per_image = PIXELS // len(optimize_frame)

    masks3 = MM.make_masks(c)
    # ---- Mapper.py:326-369
    decoders_para_list = []
    coarse_grid_para, middle_grid_para, fine_grid_para, color_grid_para = [], [], [], []
    masked_c_grad = {}
    for key, val in c.items():
        if 'coarse' in key:
            continue
        mask = masks3[key].unsqueeze(0).unsqueeze(0).repeat(1, val.shape[1], 1, 1, 1)
        val_grad = Variable(val[mask].clone().to(device), requires_grad=True)
        masked_c_grad[key] = val_grad
        masked_c_grad[key + 'mask'] = mask
        {'grid_middle': middle_grid_para, 'grid_fine': fine_grid_para, 'grid_color': color_grid_para}[key].append(val_grad)
    decoders_para_list += list(model.color_decoder.parameters())           # fix_fine: True, fix_color: False
    # ---- :374-390 (camera tensors given instead of get_tensor_from_camera(c2w): mathutils is absent)
    camera_tensor_list = []
    for fi, frame in enumerate(optimize_frame):
        if frame != oldest_frame:
            camera_tensor = Variable(tensors[fi].clone().to(device), requires_grad=True)
            camera_tensor_list.append(camera_tensor)
    # ---- :396-407
    optimizer = torch.optim.Adam([{'params': decoders_para_list, 'lr': 0},
                                  {'params': coarse_grid_para, 'lr': 0},
                                  {'params': middle_grid_para, 'lr': 0},
                                  {'params': fine_grid_para, 'lr': 0},
                                  {'params': color_grid_para, 'lr': 0},
                                  {'params': camera_tensor_list, 'lr': 0}])
    out = {'num_joint_iters': np.array(NUM_JOINT_ITERS), 'w_color_loss': np.array(w_color_loss), 'lr_factor': np.array(lr_factor),
           'BA_cam_lr': np.array(BA_cam_lr), 'pixs_per_image': np.array(pixs_per_image), 'cam': np.array([H, W, fx, fy, cx, cy]),
           'camera_tensors': np.stack([t.numpy() for t in tensors])}
    for f in range(3):
        out[f'depth_{f}'], out[f'color_{f}'] = frames[f]['depth'].numpy(), frames[f]['color'].numpy()
        out[f'est_c2w_{f}'] = frames[f]['est_c2w'].numpy()
    for key in KEYS:
        out['mask_' + key] = masks3[key].numpy()
    losses, stages, n_inside, idx_all, cam_after, cam_grads = [], [], [], [], [], []
    for joint_iter in range(NUM_JOINT_ITERS):
        # ---- :448-458
        for key, val in c.items():
            if 'coarse' not in key:
                val = val.to(device)
                val[masked_c_grad[key + 'mask']] = masked_c_grad[key]
                c[key] = val
        # ---- :460-490
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
        if stage == 'color':
            optimizer.param_groups[5]['lr'] = BA_cam_lr
        optimizer.zero_grad()
        # ---- :502-535
        batch_rays_d_list, batch_rays_o_list, batch_gt_depth_list, batch_gt_color_list = [], [], [], []
        camera_tensor_id = 0
        idx_iter = []
        for frame in optimize_frame:
            if frame != -1:
                gt_depth = keyframe_dict[frame]['depth'].to(device)
                gt_color = keyframe_dict[frame]['color'].to(device)
                if frame != oldest_frame:
                    camera_tensor = camera_tensor_list[camera_tensor_id]
                    camera_tensor_id += 1
                    c2w = get_camera_from_tensor(camera_tensor)
                else:
                    c2w = keyframe_dict[frame]['est_c2w']
            else:
                gt_depth = cur_gt_depth.to(device)
                gt_color = cur_gt_color.to(device)
                camera_tensor = camera_tensor_list[camera_tensor_id]
                c2w = get_camera_from_tensor(camera_tensor)
            seed = 1000 + 10 * joint_iter + len(idx_iter)
            torch.manual_seed(seed)
            idx_iter.append(torch.randint(H * W, (pixs_per_image,)).numpy())         # the draw get_samples is about to make
            torch.manual_seed(seed)
            batch_rays_o, batch_rays_d, batch_gt_depth, batch_gt_color = get_samples(
                0, H, 0, W, pixs_per_image, H, W, fx, fy, cx, cy, c2w, gt_depth, gt_color, device)
            assert torch.equal(batch_gt_depth, gt_depth.reshape(-1)[torch.from_numpy(idx_iter[-1])])
            batch_rays_o_list.append(batch_rays_o.float())
            batch_rays_d_list.append(batch_rays_d.float())
            batch_gt_depth_list.append(batch_gt_depth.float())
            batch_gt_color_list.append(batch_gt_color.float())
        batch_rays_d = torch.cat(batch_rays_d_list)
        batch_rays_o = torch.cat(batch_rays_o_list)
        batch_gt_depth = torch.cat(batch_gt_depth_list)
        batch_gt_color = torch.cat(batch_gt_color_list)
        # ---- :537-547
        with torch.no_grad():
            det_rays_o = batch_rays_o.clone().detach().unsqueeze(-1)
            det_rays_d = batch_rays_d.clone().detach().unsqueeze(-1)
            t = (bound.unsqueeze(0).to(device) - det_rays_o) / det_rays_d
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            inside_mask = t >= batch_gt_depth
        batch_rays_d = batch_rays_d[inside_mask]
        batch_rays_o = batch_rays_o[inside_mask]
        batch_gt_depth = batch_gt_depth[inside_mask]
        batch_gt_color = batch_gt_color[inside_mask]
        # ---- :548-575
        depth, uncertainty, color = renderer.render_batch_ray(c, model, batch_rays_d, batch_rays_o, device, stage,
                                                              gt_depth=batch_gt_depth)
        depth_mask = (batch_gt_depth > 0)
        loss = torch.abs(batch_gt_depth[depth_mask] - depth[depth_mask]).sum()
        if stage == 'color':
            loss += w_color_loss * torch.abs(batch_gt_color - color).sum()
        loss.backward(retain_graph=False)
        cam_grads.append(np.stack([t.grad.numpy().copy() for t in camera_tensor_list]))
        optimizer.step()
        losses.append(loss.item())
        stages.append(stage)
        n_inside.append(int(inside_mask.sum()))
        idx_all.append(np.stack(idx_iter))
        cam_after.append(np.stack([t.detach().numpy().copy() for t in camera_tensor_list]))
        # ---- :594-602
        optimizer.zero_grad()
        for key, val in c.items():
            if 'coarse' not in key:
                val = val.detach()
                val[masked_c_grad[key + 'mask']] = masked_c_grad[key].clone().detach()
                c[key] = val
        print(joint_iter, stage, 'loss', losses[-1], 'inside', n_inside[-1], 'of', PIXELS)
    out.update(losses=np.array(losses), stages=np.array(stages), n_inside=np.array(n_inside), idx=np.stack(idx_all),
               cam_after=np.stack(cam_after), cam_grads=np.stack(cam_grads))
    for key in KEYS:
        out['final_' + key] = c[key].numpy().copy()
    for name, p in model.color_decoder.named_parameters():
        out['final_cd_' + name] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'tiny_mapper_ba.npz'), **out)
    print('bytes', os.path.getsize(os.path.join(HERE, 'tiny_mapper_ba.npz')), 'camera drift',
          np.abs(cam_after[-1] - np.stack([t.numpy() for t in tensors[1:]])).max())


if __name__ == '__main__':
    main()
