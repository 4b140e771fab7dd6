"""-m gpu: a multi-frame oracle twin of the run harness (north_star: "per-scene ATE matches the reference
within run-to-run noise").  No dataset or pretrained weights are in the image, so the reachable form is: ONE tiny analytic,
view-consistent scene (synthetic.BoxRoom), seven frames, the reference's schedule --

    frame 0 mapped at the ground-truth pose, every later frame tracked (constant-speed initialisation, `tracking.iters`
    camera iterations, least-loss candidate) and mapped (frustum-masked grids + colour decoder, 'global' keyframe window),
    the last two mapping rounds with local bundle adjustment over the window's camera tensors --

run twice on IDENTICAL pixel draws: by the HIP harness (slam.SLAM: TrackerIteration, MapperIteration, FusedAdam,
MaskedGridOptimizer on the kernels) and by oracle/slam_oracle.py on the CPU (oracle/render_oracle + torch.optim.Adam), both
from the same seeded grids and the same decoders.  Compared: the pose estimated for every frame, the keyframe poses after
bundle adjustment, every tracking / mapping loss, and the ATE of the two trajectories."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class _Frames:
    """dataset stand-in: (idx, color float64 [H,W,3], depth float32 [H,W], c2w [4,4]) on `device`"""

    def __init__(self, frames, device):
        self.items = [(i, c.to(device), d.to(device), p.to(device)) for i, (c, d, p) in enumerate(frames)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _oracle_run(SO, params, c_cpu, bound, cam, frames, cfg, rand, frustum_mask, tensor_from_camera):
    return SO.run(params, c_cpu, bound, cam, frames, cfg, rand,
                  lambda c2w, depth, shape, b, cm: frustum_mask(c2w, depth, shape, b, cm), tensor_from_camera)


def test_multi_frame_run_twin_against_the_oracle(tmp_path, monkeypatch):
    from evennicer_slam_amd import synthetic as SY
    from evennicer_slam_amd.common import get_tensor_from_camera
    from evennicer_slam_amd.eval_ate import align
    from evennicer_slam_amd.slam import SLAM, frustum_mask
    from oracle import slam_oracle as SO
    from tests.test_hip_harness import _cfg
    H, W, n_frames = 48, 64, 7
    cfg = _cfg(None, None, H, W)
    cfg['coarse'] = True                                                     # (the coarse mapper's round runs on both sides)
    cfg['grid_memory_format'] = 'channels_last_3d'
    cfg['mapping'].update(pixels=300, iters_first=250, iters=15, every_frame=1, keyframe_every=1, mapping_window_size=4, BA=True,
                          BA_cam_lr=0.001, lr_first_factor=1)
    cfg['tracking'].update(handle_dynamic=False, pixels=300, iters=20, lr=0.002, graphed=False)
    cam = dict(H=H, W=W, fx=50.0, fy=50.0, cx=31.5, cy=23.5)

    # ---- the scene: an analytic room inside the configuration's bound, a short arc of poses
    from evennicer_slam_amd.scene import scene_bound
    bound = scene_bound(cfg['mapping']['bound'], 1.0, cfg['grid_len']['bound_divisible'])
    room = SY.BoxRoom.for_bound(bound, margin=0.12, seed=1)
    poses = SY.trajectory(room, n_frames, step=0.012, yaw_deg=0.5)
    frames = []
    for c2w in poses:
        col, dep = room.render(c2w.double(), cam)
        frames.append((col, dep, c2w.float()))

    # ---- HIP harness; decoders fitted through the HIP path stand in for the pretrained ones, then ALL runs start from them
    torch.manual_seed(0)
    slam0 = SLAM(cfg, _Frames(frames, DEV), str(tmp_path / 'fit'), device=DEV)
    slam0.prefit_decoders([0, 2, 4, 6], iters=400, pixels=400)
    dec_state = {k: v.detach().clone() for k, v in slam0.shared_decoders.state_dict().items()}
    grid_state = {k: v.detach().clone() for k, v in slam0.shared_c.items()}
    params = {k: v.cpu().clone() for k, v in dec_state.items()}
    c_cpu = {k: v.cpu().contiguous().clone() for k, v in grid_state.items()}
    del slam0

    calls = {'hip': [], 'hip2': [], 'cpu': []}
    real_randint = torch.randint

    def stream(tag, seed):
        g = torch.Generator().manual_seed(seed)
        def rand(high, n):
            calls[tag].append((int(high), int(n)))
            return real_randint(int(high), (int(n),), generator=g)
        return rand

    def hip_run(tag):
        """the harness from the common start state; every torch.randint answered from a CPU generator in call order"""
        slam = SLAM(cfg, _Frames(frames, DEV), str(tmp_path / tag), device=DEV)
        slam.shared_decoders.load_state_dict(dec_state)
        with torch.no_grad():
            for k, v in grid_state.items():
                slam.shared_c[k].copy_(v)
        rand_hip = stream(tag, 123)
        monkeypatch.setattr(torch, 'randint', lambda high, size, *a, device=None, **kw:
                            rand_hip(high, tuple(size)[0]).to(device if device is not None else 'cpu'))
        np.random.seed(7)
        try:
            res = slam.run()
        finally:
            monkeypatch.setattr(torch, 'randint', real_randint)
        return res, slam.estimate_c2w_list.clone(), [kf['est_c2w'].detach().cpu() for kf in slam.keyframe_dict]

    res, est_hip, kf_hip = hip_run('hip')
    _res2, est_hip2, _kf2 = hip_run('hip2')            # the same run again: the HIP path's own run-to-run noise (float atomics)

    # ---- the oracle's run on the same draws
    np.random.seed(7)
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(n_thr, 8))       # (many small CPU ops: all cores of the GPU box's host run them 2-3x slower than 8 threads)
    try:
        o = _oracle_run(SO, params, c_cpu, bound, cam, frames, cfg, stream('cpu', 123), frustum_mask, get_tensor_from_camera)
    finally:
        torch.set_num_threads(n_thr)
    assert calls['hip'] == calls['cpu'] == calls['hip2'], "the runs drew pixels in a different order"
    assert o['ba_rounds'] >= 2                                               # bundle adjustment took part

    # ---- poses: every frame, and the keyframes after bundle adjustment
    gt_t = torch.stack([f[2] for f in frames])[:, :3, 3]
    dt = (est_hip[:, :3, 3] - o['est'][:, :3, 3]).norm(dim=1)
    dr = (est_hip[:, :3, :3] - o['est'][:, :3, :3]).abs().amax(dim=(1, 2))
    noise = (est_hip[:, :3, 3] - est_hip2[:, :3, 3]).norm(dim=1)
    err_o = (o['est'][:, :3, 3] - gt_t).norm(dim=1)
    err_h = (est_hip[:, :3, 3] - gt_t).norm(dim=1)
    fmt = lambda v: [f"{float(x):.2e}" for x in v]
    print("translation difference HIP vs oracle per frame [m]:  ", fmt(dt))
    print("translation difference HIP vs HIP (same run twice):  ", fmt(noise))
    print("rotation-matrix difference HIP vs oracle per frame:  ", fmt(dr))
    print("distance from the ground truth, oracle / HIP [m]:    ", fmt(err_o), fmt(err_h))
    kd = [float((a[:3, 3] - b['est_c2w'][:3, 3]).norm()) for a, b in zip(kf_hip, o['keyframes'])]
    print("keyframe poses after bundle adjustment, HIP vs oracle:", fmt(kd))
    assert len(kf_hip) == len(o['keyframes'])
    # What can be asked of two runs of this loop.  Adam normalises every gradient component to a step of ~lr whatever its size,
    # the first frame's map comes out of 250 such iterations and every later map is trained from tracked poses: differences of
    # 1e-6 in a gradient (CPU vs GPU arithmetic; on the GPU alone, the order of float atomics) grow to millimetres of pose within
    # a frame or two -- between the oracle and a HIP run exactly as between two HIP runs of identical code on identical draws
    # (measured over five runs of this test: HIP vs HIP 0.06-5 cm, HIP vs oracle 0.7-3 cm per frame, each run 0.2-4.5 cm from the
    # ground truth: a 48 x 64 pixel camera in a 2 m room is a noisy tracker).  That is north_star's "within run-to-run noise".
    # The bars are absolute, from those measurements -- tying them to the noise of the two HIP runs at hand made the test depend
    # on how deterministic the HIP path happens to be.  Two realisations decorrelate frame by frame (every map is trained from the
    # poses tracked so far), so the bar grows with the frame index: 1 cm + 1.5 cm per frame (observed over eight runs: at most
    # 1.4 / 2.5 / 4.1 / 2.9 / 4.8 / 6.3 cm at frames 1..6), and every run tracks (centimetres from the ground truth, not drift).
    bar = torch.tensor([0.01 + 0.015 * f for f in range(n_frames)])
    assert bool((dt <= bar).all()), (fmt(dt), fmt(bar))
    assert all(k <= float(b) for k, b in zip(kd, bar)), (fmt(kd), fmt(bar))
    assert float(err_o[1:].max()) > 1e-4                                     # (tracking really moved the poses)
    assert float(err_o.max()) <= 0.08 and float(err_h.max()) <= 0.08        # ... and every run tracks: centimetres, not drift

    # ---- ATE of the trajectories against the ground truth (eval_ate.py: Horn alignment, RMSE of the residuals)
    gt = gt_t.numpy().T
    rmse = lambda e: float(np.sqrt(np.mean(align(e[:, :3, 3].numpy().T, gt)[2] ** 2)))
    rm_h, rm_h2, rm_c = rmse(est_hip), rmse(est_hip2), rmse(o['est'])
    print(f"ATE-RMSE: HIP {rm_h * 100:.4f} cm, HIP again {rm_h2 * 100:.4f} cm, oracle {rm_c * 100:.4f} cm")
    assert abs(rm_h - rm_c) <= 0.02                                           # (measured: 0.01-0.9 cm, HIP vs HIP 0.03-0.84 cm)
    assert max(rm_h, rm_h2, rm_c) <= 0.05
    assert res['frames'] == n_frames
