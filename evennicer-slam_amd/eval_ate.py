"""Absolute trajectory error of an estimated camera trajectory and the checkpoint container it is read from
(`src/tools/eval_ate.py:11-77,111-256`, `src/utils/Logger.py:21-32`).

Host-side numpy: Horn's closed-form rigid alignment (rotation from the SVD of the cross-covariance, reflection fixed
through the determinant) of the estimated onto the ground-truth positions, then the per-pose translational error and
its statistics, under the reference's key names.  `convert_poses` / `evaluate` keep the reference's signatures so
`python -m evennicer_slam_amd.eval_ate <ckpt.tar>` and notebook use read like the reference's tool."""
import os

import numpy as np
import torch

from .common import get_tensor_from_camera


def associate(first_list, second_list, offset=0.0, max_difference=0.02):
    """Greedy closest-stamp matching of two {stamp: data} dicts (eval_ate.py:11-41): candidate pairs closer than
    max_difference, taken in order of increasing distance, each stamp used once.  Returns sorted [(a, b)]."""
    a = np.array(sorted(first_list.keys()), dtype=np.float64)
    b = np.array(sorted(second_list.keys()), dtype=np.float64)
    ka, kb = sorted(first_list.keys()), sorted(second_list.keys())
    cand = []
    lo = np.searchsorted(b + offset, a - max_difference, side='left')
    hi = np.searchsorted(b + offset, a + max_difference, side='right')
    for i in range(len(a)):
        for j in range(lo[i], hi[i]):
            d = abs(a[i] - (b[j] + offset))
            if d < max_difference:
                cand.append((d, ka[i], kb[j]))
    cand.sort()
    used_a, used_b, matches = set(), set(), []
    for _, x, y in cand:
        if x not in used_a and y not in used_b:
            used_a.add(x)
            used_b.add(y)
            matches.append((x, y))
    matches.sort()
    return matches


def align(model, data):
    """Horn alignment of `model` (3xn) onto `data` (3xn): rot (3x3), trans (3x1), per-point error (n) after
    alignment (eval_ate.py:44-77)."""
    model = np.asarray(model, dtype=np.float64)
    data = np.asarray(data, dtype=np.float64)
    mm, dm = model.mean(1, keepdims=True), data.mean(1, keepdims=True)
    W = (model - mm) @ (data - dm).T                       # sum of outer products
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    rot = U @ S @ Vh
    trans = dm - rot @ mm
    err = rot @ model + trans - data
    return rot, trans, np.sqrt((err * err).sum(0))


def evaluate_ate(first_list, second_list, plot="", _args="", offset=0.0, scale=1.0, max_difference=0.02):
    """ATE statistics of second_list (estimate) against first_list (ground truth), both {stamp: [tx,ty,tz,...]}
    (eval_ate.py:111-223).  `plot`: path of a png of the two xy-trajectories, or "" (needs matplotlib)."""
    matches = associate(first_list, second_list, float(offset), float(max_difference))
    if len(matches) < 2:
        raise ValueError("Couldn't find matching timestamp pairs between groundtruth and estimated trajectory! "
                         "Did you choose the correct sequence?")
    first_xyz = np.array([[float(v) for v in first_list[a][0:3]] for a, _ in matches]).T
    second_xyz = np.array([[float(v) * float(scale) for v in second_list[b][0:3]] for _, b in matches]).T
    rot, trans, trans_error = align(second_xyz, first_xyz)
    if plot:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        fs, ss = sorted(first_list.keys()), sorted(second_list.keys())
        full1 = np.array([[float(v) for v in first_list[k][0:3]] for k in fs]).T
        full2 = rot @ np.array([[float(v) * float(scale) for v in second_list[k][0:3]] for k in ss]).T + trans
        fig = plt.figure()
        ax = fig.add_subplot(111)
        rmse = np.sqrt(np.dot(trans_error, trans_error) / len(trans_error))
        ax.set_title(f'len:{len(trans_error)} ATE RMSE:{rmse} {plot[:-3]}')
        ax.plot(full1[0], full1[1], '-', color='black', label='ground truth')
        ax.plot(full2[0], full2[1], '-', color='blue', label='estimated')
        ax.legend()
        ax.set_xlabel('x [m]')
        ax.set_ylabel('y [m]')
        plt.savefig(plot, dpi=90)
        plt.close(fig)
    return {
        "compared_pose_pairs": len(trans_error),
        "absolute_translational_error.rmse": np.sqrt(np.dot(trans_error, trans_error) / len(trans_error)),
        "absolute_translational_error.mean": np.mean(trans_error),
        "absolute_translational_error.median": np.median(trans_error),
        "absolute_translational_error.std": np.std(trans_error),
        "absolute_translational_error.min": np.min(trans_error),
        "absolute_translational_error.max": np.max(trans_error),
    }


def evaluate(poses_gt, poses_est, plot):
    """poses_*: [N, >=3] tensors (translation first, as `convert_poses` returns them); prints and returns the
    statistics (eval_ate.py:226-236)."""
    gt, est = poses_gt.cpu().numpy(), poses_est.cpu().numpy()
    n = gt.shape[0]
    results = evaluate_ate({i: gt[i] for i in range(n)}, {i: est[i] for i in range(n)}, plot)
    print(results)
    return results


def convert_poses(c2w_list, N, scale, gt=True):
    """c2w_list[0..N] -> ([n,7] translation + quaternion, bool mask [N+1] of the usable ground-truth poses); the
    translations are divided by `scale` IN PLACE like the reference does (eval_ate.py:239-256)."""
    poses = []
    mask = torch.ones(N + 1).bool()
    for idx in range(0, N + 1):
        if gt and (torch.isinf(c2w_list[idx]).any() or torch.isnan(c2w_list[idx]).any()):
            mask[idx] = 0
            continue
        c2w_list[idx][:3, 3] /= scale
        poses.append(get_tensor_from_camera(c2w_list[idx], Tquad=True))
    return torch.stack(poses), mask


class Logger(object):
    """Checkpoint writer with the reference's file layout and keys (`src/utils/Logger.py:6-32`):
    `<ckptsdir>/<idx:05d>.tar` = torch.save of {'c', 'decoder_state_dict', 'gt_c2w_list', 'estimate_c2w_list',
    'keyframe_list', 'selected_keyframes', 'idx'} in the legacy (non-zip) serialisation."""

    def __init__(self, cfg, args, slam):
        self.verbose = slam.verbose
        self.ckptsdir = slam.ckptsdir
        self.shared_c = slam.shared_c
        self.gt_c2w_list = slam.gt_c2w_list
        self.shared_decoders = slam.shared_decoders
        self.estimate_c2w_list = slam.estimate_c2w_list

    def log(self, idx, keyframe_dict, keyframe_list, selected_keyframes=None):
        path = os.path.join(self.ckptsdir, '{:05d}.tar'.format(idx))
        torch.save({'c': self.shared_c, 'decoder_state_dict': self.shared_decoders.state_dict(),
                    'gt_c2w_list': self.gt_c2w_list, 'estimate_c2w_list': self.estimate_c2w_list,
                    'keyframe_list': keyframe_list, 'selected_keyframes': selected_keyframes, 'idx': idx},
                   path, _use_new_zipfile_serialization=False)
        if self.verbose:
            print('Saved checkpoints at', path)
        return path


def latest_checkpoint(ckptsdir):
    names = [f for f in sorted(os.listdir(ckptsdir)) if 'tar' in f] if os.path.isdir(ckptsdir) else []
    return os.path.join(ckptsdir, names[-1]) if names else None


def evaluate_checkpoint(ckpt_path, scale=1.0, plot=""):
    """ATE of the trajectory stored in one checkpoint (the body of eval_ate.py's __main__, :281-303)."""
    ckpt = torch.load(ckpt_path, map_location=torch.device('cpu'), weights_only=False)
    N = ckpt['idx']
    poses_gt, mask = convert_poses(ckpt['gt_c2w_list'], N, scale)
    poses_est, _ = convert_poses(ckpt['estimate_c2w_list'], N, scale)
    return evaluate(poses_gt, poses_est[mask], plot)


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description='ATE of the trajectory in a checkpoint (or the newest of a ckpts folder).')
    ap.add_argument('path')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--plot', default='')
    a = ap.parse_args()
    p = a.path if os.path.isfile(a.path) else latest_checkpoint(a.path)
    if p is None:
        raise SystemExit(f'no checkpoint under {a.path}')
    print('Get ckpt :', p)
    evaluate_checkpoint(p, a.scale, a.plot)
