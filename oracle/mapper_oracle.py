"""CPU restatement of the mapper's inner-iteration glue.  TEST INFRASTRUCTURE ONLY: imported by
tests/ (and nothing in the product path).  Pinned by tests/golden/tiny_mapper_iters.npz, which was produced by the
reference's own statements (tests/golden/make_golden_mapper.py).

Follows src/Mapper.py: :343-361 val_grad = val[mask] as the optimised leaf; :409-413 torch.optim.Adam with one
parameter group per grid; :448-458 val[mask] = val_grad before every render; :460-473 stage / learning-rate
schedule; :548-575 render, RGB-D loss, backward, step; :594-602 zero_grad and write-back.  The renderer is
oracle/render_oracle.py; the optimiser is torch.optim.Adam itself (the third-party arithmetic at this boundary)."""
import torch

from . import render_oracle as R

KEYS = ('grid_middle', 'grid_fine', 'grid_color')
STAGE_LR = {   # configs/nice_slam.yaml:71-95 (decoders, middle, fine, color)
    'middle': (0.0, 0.1, 0.0, 0.0),
    'fine': (0.0, 0.005, 0.005, 0.0),
    'color': (0.005, 0.005, 0.005, 0.005),
}


def stage_of(joint_iter, num_joint_iters, middle_iter_ratio=0.4, fine_iter_ratio=0.6):
    """Mapper.py:460-467."""
    if joint_iter <= int(num_joint_iters * middle_iter_ratio):
        return 'middle'
    if joint_iter <= int(num_joint_iters * fine_iter_ratio):
        return 'fine'
    return 'color'


def optimize_map_iters(params, grids, masks, rays_o, rays_d, gt_depth, gt_color, bound, num_joint_iters,
                       w_color=0.2, lr_factor=1.0):
    """params: decoder state (tensors; the colour decoder's entries are optimised in place), grids: dict of
    [1,32,D,H,W] tensors (updated in place like the reference's shared `c`), masks: dict key -> bool [D,H,W].
    Returns the list of per-iteration losses."""
    c = {k: v for k, v in grids.items()}
    masked, mask5 = {}, {}
    for key in KEYS:
        mask5[key] = masks[key][None, None].repeat(1, c[key].shape[1], 1, 1, 1)
        masked[key] = c[key][mask5[key]].clone().requires_grad_(True)                 # :349-351
    dec = [v for k, v in params.items() if k.startswith('color_decoder.')]
    for p in dec:
        p.requires_grad_(True)
    opt = torch.optim.Adam([{'params': dec, 'lr': 0}, {'params': [], 'lr': 0},
                            {'params': [masked['grid_middle']], 'lr': 0},
                            {'params': [masked['grid_fine']], 'lr': 0},
                            {'params': [masked['grid_color']], 'lr': 0}])
    losses = []
    for it in range(num_joint_iters):
        for key in KEYS:                                                               # :448-458
            val = c[key]
            val[mask5[key]] = masked[key]
            c[key] = val
        stage = stage_of(it, num_joint_iters)
        lr = STAGE_LR[stage]
        opt.param_groups[0]['lr'] = lr[0] * lr_factor
        opt.param_groups[2]['lr'] = lr[1] * lr_factor
        opt.param_groups[3]['lr'] = lr[2] * lr_factor
        opt.param_groups[4]['lr'] = lr[3] * lr_factor
        opt.zero_grad()
        depth, var, color = R.render_batch_ray(params, c, rays_d, rays_o, stage, bound, gt_depth=gt_depth)
        loss = R.mapper_loss(depth, color, gt_depth, gt_color, stage, w_color)
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
        opt.zero_grad()
        for key in KEYS:                                                               # :596-602
            val = c[key].detach()
            val[mask5[key]] = masked[key].clone().detach()
            c[key] = val
    for key in KEYS:
        grids[key] = c[key]
    return losses
