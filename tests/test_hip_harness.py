"""-m gpu: the single-process run harness (slam.SLAM) end to end on a synthetic Replica_event-layout
sequence: frames rendered from a random map along a short trajectory, written as jpg / png / traj.txt, read back by the
cv2-free reader, tracked and mapped on the HIP path with the reference's schedule, logged as a reference-format
checkpoint and scored with eval_ate.  No dataset or pretrained weights exist in the image, so the assertions are about
the pipeline (every stage runs, poses stay close to the ground truth on this easy sequence, BA and the static-shape
mapper take part), not about reconstruction quality."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cfg(inp, evf, H, W):
    from tests.hip_util import cfg_like
    cfg = cfg_like(grid_len={'coarse': 0.8, 'middle': 0.4, 'fine': 0.2, 'color': 0.2, 'bound_divisible': 0.32})
    cfg.update({'dataset': 'replica_event', 'scale': 1,
                'cam': dict(H=H, W=W, fx=50.0, fy=50.0, cx=31.5, cy=23.5, png_depth_scale=6553.5, crop_edge=0),
                'data': {'dim': 3, 'input_folder': inp, 'event_folder': evf}})
    cfg['mapping'] = {'bound': [[-1.0, 1.1], [-0.9, 0.8], [-0.7, 0.6]], 'w_color_loss': 0.2, 'lr_factor': 1, 'BA': True,
                      'BA_cam_lr': 0.001, 'middle_iter_ratio': 0.4, 'fine_iter_ratio': 0.6, 'fix_fine': True, 'fix_color': False,
                      'pixels': 120, 'iters_first': 30, 'iters': 8, 'every_frame': 2, 'keyframe_every': 1, 'mapping_window_size': 4,
                      'frustum_feature_selection': True,
                      'stage': {'coarse': dict(decoders_lr=0.0, coarse_lr=0.001, middle_lr=0.0, fine_lr=0.0, color_lr=0.0),
                                'middle': dict(decoders_lr=0.0, coarse_lr=0.0, middle_lr=0.1, fine_lr=0.0, color_lr=0.0),
                                'fine': dict(decoders_lr=0.0, coarse_lr=0.0, middle_lr=0.005, fine_lr=0.005, color_lr=0.0),
                                'color': dict(decoders_lr=0.005, coarse_lr=0.0, middle_lr=0.005, fine_lr=0.005, color_lr=0.005)}}
    cfg['tracking'] = {'device': DEV, 'w_color_loss': 0.5, 'ignore_edge_W': 4, 'ignore_edge_H': 4, 'handle_dynamic': True,
                       'use_color_in_tracking': True, 'lr': 0.001, 'pixels': 100, 'iters': 6, 'const_speed_assumption': True,
                       'gt_camera': False}
    cfg['event'] = {'activate_events': False, 'blur': True, 'kernel_sizes': [9], 'kernel_weights': [1], 'unblurred_weight': 0,
                    'balancer': 0.025, 'scale_factor': 0.5}
    return cfg


def test_run_harness_on_a_synthetic_sequence(tmp_path):
    import evennicer_slam_amd as E
    from evennicer_slam_amd import datasets as D
    from evennicer_slam_amd.slam import SLAM
    from tests.hip_util import tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    H, W, n = 48, 64, 9
    frames, poses, events = [], [], []
    th = 0.3
    base = torch.tensor([[np.cos(th), 0, np.sin(th), 0.1], [0, 1, 0, -0.05], [-np.sin(th), 0, np.cos(th), 0.2], [0, 0, 0, 1.0]],
                        dtype=torch.float32)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for i in range(n):
            c2w = base.clone()
            c2w[:3, 3] += torch.tensor([0.01 * i, 0.004 * i, -0.006 * i])
            gt = torch.rand(H, W, generator=g) * 0.8 + 0.4                      # guide depths for the sampler
            d, u, c = renderer.render_img(grids, model, c2w[:3].to(DEV), DEV, 'color', gt_depth=gt.to(DEV))
            frames.append((c.clamp(0, 1).cpu().numpy(), d.clamp(0.05, 5.0).float().cpu().numpy()))
            poses.append(c2w.numpy())
            if i > 0:
                events.append(np.zeros((H, W, 2), dtype=np.uint8))
    inp, evf = D.write_replica_event_sequence(str(tmp_path / 'data'), frames, poses, 6553.5, events)
    cfg = _cfg(inp, evf, H, W)
    ds = D.get_dataset(cfg, types.SimpleNamespace(input_folder=None, event_folder=None), cfg['scale'], device=DEV)
    assert len(ds) == n and len(ds[3]) == 6
    torch.manual_seed(0)
    np.random.seed(0)
    slam = SLAM(cfg, ds, str(tmp_path / 'out'), device=DEV, static_shapes=True)
    res = slam.run()
    assert res['frames'] == n and res['fps'] > 0
    ckpt = torch.load(res['ckpt'], map_location='cpu', weights_only=False)
    assert ckpt['idx'] == n - 1 and len(ckpt['keyframe_list']) >= 4
    assert set(ckpt['c']) == {'grid_coarse', 'grid_middle', 'grid_fine', 'grid_color'}
    assert set(ckpt['decoder_state_dict']) == set(model.state_dict())
    est, gtl = ckpt['estimate_c2w_list'], ckpt['gt_c2w_list']
    assert torch.equal(est[0], gtl[0])                                          # frame 0 takes the ground-truth pose
    assert bool(torch.isfinite(est).all())
    # (frames rendered from a random-weight map with random guide depths are not a view-consistent scene: the tracker's
    # numbers are pinned against reference fixtures in test_hip_tracker / test_hip_event; here the poses only have to stay
    # finite and bounded while every stage runs)
    assert float((est[1:, :3, 3] - gtl[1:, :3, 3]).norm(dim=1).max()) < 0.5
    ate = slam.evaluate(res['ckpt'])
    assert ate['compared_pose_pairs'] == n and np.isfinite(ate['absolute_translational_error.rmse'])
    # the map really was optimised, the tracker works on its own copy of it
    assert float((slam.shared_c['grid_middle'] - 0).abs().max()) > 0.02
    assert slam.tracker.c['grid_middle'] is not slam.shared_c['grid_middle']
    assert torch.equal(slam.tracker.c['grid_middle'], slam.shared_c['grid_middle'])
    # with tracking.gt_camera the schedule runs on the ground-truth poses: the checkpoint's trajectory is exact, ATE = 0
    cfg['tracking']['gt_camera'] = True
    cfg['mapping']['BA'] = False
    slam2 = SLAM(cfg, ds, str(tmp_path / 'out_gt'), device=DEV, static_shapes=False)
    res2 = slam2.run(max_frames=5)
    ckpt2 = torch.load(res2['ckpt'], map_location='cpu', weights_only=False)
    assert ckpt2['idx'] == 4 and torch.equal(ckpt2['estimate_c2w_list'][:5], ckpt2['gt_c2w_list'][:5])
    assert slam2.evaluate(res2['ckpt'])['absolute_translational_error.rmse'] < 1e-6


@pytest.mark.parametrize("graphed", [False, True])
def test_tracking_converges_on_a_view_consistent_sequence(tmp_path, graphed):
    """SURVEY f3 / VERDICT r2 item 5: a sequence on which tracking MUST converge.  30 frames of an analytic room
    (synthetic.BoxRoom: every frame is a rendering of the same geometry and colours) along a known trajectory, read back
    through the Replica_event reader; decoders pre-fitted on five ground-truth-posed frames in place of the pretrained
    checkpoints the image lacks (SLAM.prefit_decoders), then the reference's schedule -- frame 0 mapped at the ground-truth
    pose, every later pose tracked from its constant-speed initialisation, mapping and COARSE mapping every second frame.
    Measured over 30 runs (`tools/run_synthetic_slam.py`): ATE-RMSE 0.6-2.6 cm, median 1.0 cm, against 10.5 cm with the camera
    iterations switched off (poses left at their constant-speed initialisation) -- about a tenth.  The run is not bit-reproducible
    (float-atomic ordering in the gradients, amplified by 40 Adam steps per frame), so the assertion leaves room: below HALF of
    the no-tracking ATE and below 5 cm (the trajectory is 36 cm long)."""
    from evennicer_slam_amd import datasets as D
    from evennicer_slam_amd.slam import SLAM
    from evennicer_slam_amd.synthetic import demo_config, write_demo_sequence
    n = 30
    cam = dict(H=60, W=80, fx=70.0, fy=70.0, cx=39.5, cy=29.5)
    (inp, evf), poses = write_demo_sequence(str(tmp_path / 'data'), n, cam)
    # graphed: the camera iterations of every frame as replays of one hipGraph (SLAM._track_graphed: GraphedCameraIteration with
    # refresh_map / set_frame per frame, least-loss candidate by one argmin on the device) instead of Python-driven iterations
    cfg = demo_config(inp, evf, cam, device=DEV, env={'TRACK_GRAPHED': int(graphed)})
    ds = D.get_dataset(cfg, types.SimpleNamespace(input_folder=None, event_folder=None), 1, device=DEV)
    assert len(ds) == n
    ate = {}
    for tag, iters in (('tracked', None), ('const_speed_init', 0)):
        torch.manual_seed(0)
        np.random.seed(0)
        slam = SLAM(cfg, ds, str(tmp_path / ('out_' + tag)), device=DEV, static_shapes=True)
        coarse0 = slam.shared_c['grid_coarse'].clone()
        fit = slam.prefit_decoders(list(range(0, n, 5)), iters=400)
        assert np.isfinite(fit)
        res = slam.run(tracking_iters=iters)
        ate[tag] = slam.evaluate(res['ckpt'])['absolute_translational_error.rmse']
        # the coarse mapper's stage ran: grid_coarse was optimised (and nothing else touches it)
        assert not torch.equal(slam.shared_c['grid_coarse'], coarse0) and bool(torch.isfinite(slam.last_coarse_loss))
    assert ate['const_speed_init'] > 0.05                       # without camera iterations the estimate falls behind by > 5 cm
    assert ate['tracked'] < 0.5 * ate['const_speed_init'], ate
    assert ate['tracked'] < 0.05, ate


def test_harness_on_an_rpg_event_layout_sequence(tmp_path):
    """BASELINE config 5's data format through the harness: a grey-scale RPG_event-layout sequence (png frames, 16-bit depth at
    png_depth_scale 1000, event pngs with channels (+, -, 0), a lens model in cam.distortion) read by datasets.RPG_event -- the
    colour and event images undistorted, the depth not -- tracked and mapped for a few frames.  The frames are renderings of the
    analytic room through an ideal pinhole, so with the (weak) lens model removed from them they are only approximately
    consistent: the assertions are about the pipeline (reader -> tracker -> mapper -> checkpoint), not about accuracy."""
    from evennicer_slam_amd import datasets as D
    from evennicer_slam_amd.scene import scene_bound
    from evennicer_slam_amd.slam import SLAM
    from evennicer_slam_amd.synthetic import BoxRoom, demo_config, trajectory
    n = 5
    cam = dict(H=52, W=70, fx=39.3, fy=39.3, cx=34.5, cy=25.5)            # configs/rpg/rpg.yaml:62-68 at a fifth of the resolution
    bound = scene_bound([[-1.0, 1.1], [-0.9, 0.8], [-0.7, 0.6]], 1.0, 0.32)
    room = BoxRoom.for_bound(bound, margin=0.12, seed=1)
    poses = trajectory(room, n, step=0.012, yaw_deg=0.5)
    frames, events = [], []
    rng = np.random.default_rng(0)
    for i, c2w in enumerate(poses):
        col, dep = room.render(c2w.double(), cam)
        grey = np.clip(np.rint(col.mean(-1).numpy() * 255), 0, 255).astype(np.uint8)
        frames.append((grey, dep.numpy()))
        if i > 0:
            events.append(rng.integers(0, 2, (cam['H'], cam['W'], 2)).astype(np.uint8))
    inp, evf = D.write_rpg_event_sequence(str(tmp_path / 'data'), frames, [p.numpy() for p in poses], 1000.0, events)
    cfg = demo_config(inp, evf, cam, device=DEV, env={'ITERS_FIRST': 60, 'MAP_ITERS': 10, 'TRACK_ITERS': 5, 'EVERY': 2})
    cfg['dataset'] = 'rpg_event'
    cfg['cam'] = dict(cam, png_depth_scale=1000.0, crop_edge=0, distortion=[-0.08409333, 0.05335822, -0.00065521, -0.0001679, 0, 0, 0, 0])
    ds = D.get_dataset(cfg, types.SimpleNamespace(input_folder=None, event_folder=None), 1, device=DEV)
    assert isinstance(ds, D.RPG_event) and len(ds) == n
    idx, color, depth, event, mask, pose = ds[2]
    assert tuple(color.shape) == (52, 70, 3) and bool((color[..., 0] == color[..., 1]).all())       # grey replicated
    assert tuple(event.shape) == (52, 70, 2) and tuple(mask.shape) == (52, 70) and depth.dtype == torch.float32
    assert float((depth.cpu() - torch.from_numpy(frames[2][1])).abs().max()) <= 0.5 / 1000.0 + 1e-6  # the depth is NOT undistorted
    torch.manual_seed(0)
    np.random.seed(0)
    slam = SLAM(cfg, ds, str(tmp_path / 'out'), device=DEV, static_shapes=True)
    res = slam.run()
    ckpt = torch.load(res['ckpt'], map_location='cpu', weights_only=False)
    assert ckpt['idx'] == n - 1 and bool(torch.isfinite(ckpt['estimate_c2w_list']).all())
    assert torch.equal(ckpt['estimate_c2w_list'][0], ckpt['gt_c2w_list'][0])
    ate = slam.evaluate(res['ckpt'])
    assert ate['compared_pose_pairs'] == n and np.isfinite(ate['absolute_translational_error.rmse'])
