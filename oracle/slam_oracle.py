"""CPU restatement of a whole multi-frame run: tracker and mapper alternating on the reference's schedule.  TEST
INFRASTRUCTURE ONLY (imported by tests/test_hip_twin.py; nothing in the product path imports it).

What it follows (reference file:line):
  frame loop        src/Tracker.py:262-466 and src/Mapper.py:736-879 -- frame 0 at the ground-truth pose with
                    `mapping.iters_first` joint iterations (Tracker.py:283-285, Mapper.py:794-798), every later frame tracked
                    from the constant-speed initialisation (Tracker.py:295-303), mapped every `mapping.every_frame` frames and
                    on the last one, keyframes every `mapping.keyframe_every` frames (Mapper.py:687-692)
  camera iterations Tracker.py:141-197, 303-330: Adam on the 7-vector, the candidate with the smallest loss is kept
                    (oracle/tracker_oracle.camera_iteration)
  mapping round     Mapper.py:280-303 ('global' keyframe selection), :326-413 (frustum-masked grid leaves, colour decoder,
                    camera tensors of the non-fixed frames under BA, ONE torch.optim.Adam with a parameter group each),
                    :448-602 (per iteration: val[mask] = val_grad, stage / learning rates, one get_samples draw per frame, the
                    in-bound prefilter, render, RGB-D loss, backward, step, write-back), :644-660 (poses back after BA)
  ATE               src/tools/eval_ate.py:44-78 through the caller (tests use evennicer-slam_amd/eval_ate.py, pinned against the
                    reference's evaluate_ate on the CPU)

The renderer is oracle/render_oracle.py (pinned by the reference-generated fixtures), the optimiser torch.optim.Adam itself.
Randomness enters through `rand(high, n)` (pixel draws, in call order) and numpy's global generator (keyframe selection):
the HIP harness is fed the same streams, so both runs see identical pixels.  Two pieces of host-side torch code are shared
with the product rather than restated, because they are the product's own restatements of third-party functions that are
pinned elsewhere: `frustum_mask` (cv2.remap form of Mapper.get_mask_from_c2w: tests/test_harness_cpu.py) and
`get_tensor_from_camera` (mathutils' mat3_to_quat); both are passed in by the caller."""
import numpy as np
import torch

from . import render_oracle as R
from . import tracker_oracle as TO

MAP_KEYS = ('grid_middle', 'grid_fine', 'grid_color')


def _stage_of(it, n, middle_ratio, fine_ratio):                      # Mapper.py:460-467
    if it <= int(n * middle_ratio):
        return 'middle'
    if it <= int(n * fine_ratio):
        return 'fine'
    return 'color'


def mapping_round(params, c, bound, frames, cam, masks, m, iters, lr_factor, ba, rand, tensor_from_camera):
    """One call of Mapper.optimize_map on `frames` (dicts: depth, color, c2w [4,4], fixed); updates c (in place) and the colour
    decoder entries of params; returns (camera tensors or None per frame, last loss)."""
    H, W, fx, fy, cx, cy = (cam[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
    masked, mask5 = {}, {}
    for key in MAP_KEYS:                                             # :330-361
        mk = masks[key] if masks is not None else torch.ones(c[key].shape[2:], dtype=torch.bool)
        mask5[key] = mk[None, None].repeat(1, c[key].shape[1], 1, 1, 1)
        masked[key] = c[key][mask5[key]].clone().requires_grad_(True)
    dec = [v for k, v in params.items() if k.startswith('color_decoder.')]          # :363-369 (fix_fine, not fix_color)
    for p in dec:
        p.requires_grad_(True)
    cams = [None] * len(frames)
    if ba:                                                           # :374-390
        for i, f in enumerate(frames):
            if not f['fixed']:
                cams[i] = tensor_from_camera(f['c2w']).float().clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': dec, 'lr': 0}, {'params': [masked['grid_middle']], 'lr': 0},
                            {'params': [masked['grid_fine']], 'lr': 0}, {'params': [masked['grid_color']], 'lr': 0},
                            {'params': [t for t in cams if t is not None], 'lr': 0}])       # :396-413
    n = m['pixels'] // len(frames)
    loss = None
    for it in range(iters):
        for key in MAP_KEYS:                                         # :448-458
            val = c[key]
            val[mask5[key]] = masked[key]
            c[key] = val
        stage = _stage_of(it, iters, m['middle_iter_ratio'], m['fine_iter_ratio'])
        st = m['stage'][stage]
        opt.param_groups[0]['lr'] = st['decoders_lr'] * lr_factor                       # :469-490
        opt.param_groups[1]['lr'] = st['middle_lr'] * lr_factor
        opt.param_groups[2]['lr'] = st['fine_lr'] * lr_factor
        opt.param_groups[3]['lr'] = st['color_lr'] * lr_factor
        if ba and stage == 'color':
            opt.param_groups[4]['lr'] = m['BA_cam_lr']
        opt.zero_grad()
        ro, rd, gd, gc = [], [], [], []
        for f, ct in zip(frames, cams):                              # :502-535
            c2w = TO.camera_from_tensor(ct) if ct is not None else f['c2w'][:3]
            o, d, dep, col = R.sample_pixels(0, H, 0, W, n, c2w, f['depth'], f['color'], fx, fy, cx, cy, idx=rand(H * W, n))
            ro.append(o.float()); rd.append(d.float()); gd.append(dep.float()); gc.append(col.float())
        ro, rd, gd, gc = torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)
        inside = TO.inside_prefilter(ro, rd, gd, bound.to(ro.dtype))                    # :537-547
        ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
        depth, _var, color = R.render_batch_ray(params, c, rd, ro, stage, bound, gt_depth=gd)
        loss = R.mapper_loss(depth, color, gd, gc, stage, m['w_color_loss'])            # :553-562
        loss.backward()
        opt.step()                                                   # :573-575
        opt.zero_grad()
        for key in MAP_KEYS:                                         # :596-602
            val = c[key].detach()
            val[mask5[key]] = masked[key].clone().detach()
            c[key] = val
    for p in dec:
        p.requires_grad_(False)
    return [None if t is None else t.detach() for t in cams], (float(loss.item()) if loss is not None else None)


def coarse_round(params, c, bound, frames, cam, m, iters, lr_factor, rand):
    """The coarse mapper's round (the reference's third process: Mapper(..., coarse_mapper=True), EvenNICER_SLAM.py:303-311):
    the same frames at fixed poses, stage `coarse` only (Mapper.py:460-461), every voxel of grid_coarse a leaf (:326-328), no
    depth guidance in the render (:550), depth term of the loss only."""
    H, W, fx, fy, cx, cy = (cam[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
    leaf = c['grid_coarse'].clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [leaf], 'lr': m['stage']['coarse']['coarse_lr'] * lr_factor}])
    n = m['pixels'] // len(frames)
    for _ in range(iters):
        opt.zero_grad()
        ro, rd, gd, gc = [], [], [], []
        for f in frames:
            o, d, dep, col = R.sample_pixels(0, H, 0, W, n, f['c2w'][:3], f['depth'], f['color'], fx, fy, cx, cy, idx=rand(H * W, n))
            ro.append(o.float()); rd.append(d.float()); gd.append(dep.float()); gc.append(col.float())
        ro, rd, gd, gc = torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)
        inside = TO.inside_prefilter(ro, rd, gd, bound.to(ro.dtype))
        ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
        depth, _var, color = R.render_batch_ray(params, dict(c, grid_coarse=leaf), rd, ro, 'coarse', bound, gt_depth=None)
        R.mapper_loss(depth, color, gd, gc, 'coarse', m['w_color_loss']).backward()
        opt.step()
    c['grid_coarse'] = leaf.detach()


def track_frame(params, c, bound, est, depth, color, cam, t, rand, tensor_from_camera):
    """Tracker.py:303-330 for one frame: `tracking.iters` camera iterations from the initial pose `est`; returns (c2w [4,4] of
    the least-loss candidate, losses)."""
    H, W, fx, fy, cx, cy = (cam[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
    He, We = t['ignore_edge_H'], t['ignore_edge_W']
    ct = tensor_from_camera(est.detach()).float().clone().requires_grad_(True)
    opt = torch.optim.Adam([ct], lr=t['lr'])
    best, best_loss, losses = ct.detach().clone(), None, []
    for _ in range(t['iters']):
        opt.zero_grad()
        idx = rand((H - 2 * He) * (W - 2 * We), t['pixels'])
        loss, _d, _v, _c, _inside = TO.camera_iteration(params, c, bound, ct, depth, color, (H, W, fx, fy, cx, cy), (He, We), t['pixels'],
                                                        t['w_color_loss'], idx=idx)
        loss.backward()
        opt.step()
        lv = float(loss.item())
        losses.append(lv)
        if best_loss is None or lv < best_loss:                      # the candidate kept is the pose AFTER the step (:321-330)
            best_loss, best = lv, ct.detach().clone()
    c2w = torch.eye(4)
    c2w[:3] = TO.camera_from_tensor(best)
    return c2w, losses


def run(params, c, bound, cam, frames, cfg, rand, frustum_mask, tensor_from_camera):
    """frames: list of (color [H,W,3], depth [H,W], gt_c2w [4,4]) CPU tensors.  Returns dict(est=[n,4,4], keyframes=[...],
    track_losses, map_losses)."""
    m, t = cfg['mapping'], cfg['tracking']
    n = len(frames)
    est_list = torch.zeros((n, 4, 4))
    keyframes = []
    out = dict(track_losses=[], map_losses=[], ba_rounds=0)
    for idx, (color, depth, gt_c2w) in enumerate(frames):
        color = color.float()
        if idx == 0:
            c2w = gt_c2w.clone().float()
        else:
            pre = est_list[idx - 1].float()
            if t.get('const_speed_assumption', True) and idx - 2 >= 0:       # Tracker.py:295-301
                est = (pre @ est_list[idx - 2].float().inverse()) @ pre
            else:
                est = pre
            c2w, losses = track_frame(params, c, bound, est, depth, color, cam, t, rand, tensor_from_camera)
            out['track_losses'].append(losses)
        if idx == 0 or idx % m['every_frame'] == 0 or idx == n - 1:
            iters = m['iters_first'] if idx == 0 else m['iters']
            sel = []
            if keyframes:                                            # Mapper.py:280-303 ('global')
                n_old = len(keyframes) - 1
                pick = list(np.random.permutation(n_old)[:max(m.get('mapping_window_size', 5) - 2, 0)]) if n_old > 0 else []
                pick = sorted(set(int(p) for p in pick) | {len(keyframes) - 1})
                oldest = min(pick)
                for k in pick:
                    kf = keyframes[k]
                    sel.append(dict(depth=kf['depth'], color=kf['color'], c2w=kf['est_c2w'], fixed=(k == oldest), key=k))
            sel.append(dict(depth=depth, color=color, c2w=c2w, fixed=False, key=-1))
            masks = None
            if m.get('frustum_feature_selection', True):
                masks = {k: frustum_mask(c2w, depth, tuple(c[k].shape[2:]), bound, cam) for k in MAP_KEYS}
            ba = bool(m.get('BA', False)) and len(keyframes) > 4 and idx > 0
            lr_factor = m.get('lr_first_factor', m['lr_factor']) if idx == 0 else m['lr_factor']
            cams, loss = mapping_round(params, c, bound, sel, cam, masks, m, iters, lr_factor, ba, rand, tensor_from_camera)
            out['map_losses'].append(loss)
            if cfg.get('coarse', False):
                coarse_round(params, c, bound, [dict(f, fixed=True) for f in sel], cam, m, iters, lr_factor, rand)
            if ba:                                                   # Mapper.py:644-660
                out['ba_rounds'] += 1
                for f, ct in zip(sel, cams):
                    if ct is None:
                        continue
                    p = torch.eye(4)
                    p[:3] = TO.camera_from_tensor(ct)
                    if f['key'] == -1:
                        c2w = p
                    else:
                        keyframes[f['key']]['est_c2w'] = p
            if idx % m['keyframe_every'] == 0 or idx == n - 2:       # Mapper.py:687-692
                keyframes.append(dict(idx=idx, color=color, depth=depth, est_c2w=c2w.clone(), gt_c2w=gt_c2w.clone()))
        est_list[idx] = c2w.detach()
    out['est'] = est_list
    out['keyframes'] = keyframes
    return out
