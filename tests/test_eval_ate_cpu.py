"""CPU: ATE evaluation and the checkpoint container (evennicer-slam_amd/eval_ate.py) against the fixture produced
by the reference's own src/tools/eval_ate.py (tests/golden/make_golden_ate.py)."""
import types

import numpy as np
import pytest
import torch

from tests.util import load


def test_ate_matches_reference_fixture():
    from evennicer_slam_amd import eval_ate as EA
    fx = load("ate_cases")
    for case in range(4):
        gt, est = fx[f'c{case}_gt'], fx[f'c{case}_est']
        n = gt.shape[0]
        res = EA.evaluate_ate({i: gt[i] for i in range(n)}, {i: est[i] for i in range(n)}, "")
        assert '\n'.join(sorted(res.keys())) == str(fx[f'c{case}_stat_keys'])
        got = np.array([res[k] for k in sorted(res.keys())], dtype=np.float64)
        assert np.allclose(got, fx[f'c{case}_stats'], rtol=1e-9, atol=1e-12)
        rot, trans, err = EA.align(est[:, :3].T, gt[:, :3].T)
        assert np.allclose(rot, fx[f'c{case}_rot'], atol=1e-10) and np.allclose(trans, fx[f'c{case}_trans'], atol=1e-10)
        assert np.allclose(err, fx[f'c{case}_err'], atol=1e-10)
        assert abs(np.linalg.det(rot) - 1) < 1e-9                       # a proper rotation, also in the reflected case
    m = EA.associate({float(s): None for s in fx['assoc_a']}, {float(s): None for s in fx['assoc_b']}, offset=-0.05,
                     max_difference=0.02)
    assert np.array_equal(np.array(m, dtype=np.float64), fx['assoc_matches'])
    with pytest.raises(ValueError):
        EA.evaluate_ate({0: [0, 0, 0]}, {5: [0, 0, 0]}, "")


def test_checkpoint_roundtrip_and_evaluation(tmp_path):
    from evennicer_slam_amd import eval_ate as EA
    g = torch.Generator().manual_seed(0)
    n = 12
    gt = torch.eye(4).repeat(n, 1, 1)
    gt[:, :3, 3] = torch.cumsum(0.1 * torch.rand(n, 3, generator=g), 0)
    est = gt.clone()
    est[:, :3, 3] += 0.01 * torch.randn(n, 3, generator=g)
    gt[5] = float('nan')                                                 # an unusable ground-truth pose is masked out
    dec = torch.nn.Linear(3, 2)
    slam = types.SimpleNamespace(verbose=False, ckptsdir=str(tmp_path), shared_c={'grid_coarse': torch.zeros(1, 2, 2, 2, 2)},
                                 gt_c2w_list=gt, shared_decoders=dec, estimate_c2w_list=est)
    path = EA.Logger(None, None, slam).log(n - 1, None, [0, 5], selected_keyframes={0: [0]})
    assert path.endswith('00011.tar') and EA.latest_checkpoint(str(tmp_path)) == path
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert sorted(ck.keys()) == ['c', 'decoder_state_dict', 'estimate_c2w_list', 'gt_c2w_list', 'idx', 'keyframe_list',
                                 'selected_keyframes']
    res = EA.evaluate_checkpoint(path, scale=1.0)
    assert res['compared_pose_pairs'] == n - 1
    assert 0 < res['absolute_translational_error.rmse'] < 0.05
    poses, mask = EA.convert_poses(ck['estimate_c2w_list'], n - 1, 2.0, gt=False)
    assert poses.shape == (n, 7) and bool(mask.all())
    assert torch.allclose(poses[:, 3:], torch.tensor([1.0, 0, 0, 0]).repeat(n, 1))      # identity rotations: (w,x,y,z)
