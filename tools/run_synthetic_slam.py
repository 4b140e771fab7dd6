"""The run harness on an analytic, view-consistent sequence (synthetic.BoxRoom): ATE of the full schedule against the ATE of
poses left at their constant-speed initialisation (tracking_iters = 0).  usage: python tools/run_synthetic_slam.py [n_frames]"""
import os, sys, tempfile, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import evennicer_slam_amd as E
from evennicer_slam_amd import datasets as D
from evennicer_slam_amd.slam import SLAM
from evennicer_slam_amd.synthetic import BoxRoom, trajectory
from evennicer_slam_amd.scene import scene_bound

DEV = 'cuda:0'


from evennicer_slam_amd.synthetic import demo_config, write_demo_sequence


def run(n=30, verbose=True):
    cam = dict(H=60, W=80, fx=70.0, fy=70.0, cx=39.5, cy=29.5)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        (inp, evf), poses = write_demo_sequence(os.path.join(tmp, 'data'), n, cam, step=float(os.environ.get('STEP', 0.012)), yaw_deg=float(os.environ.get('YAW', 0.5)))
        cfg = demo_config(inp, evf, cam, device=DEV, env=os.environ)
        ds = D.get_dataset(cfg, types.SimpleNamespace(input_folder=None, event_folder=None), 1, device=DEV)
        for tag, iters in ((('tracked', None),) if os.environ.get('SKIP_BASE') == '1' else (('tracked', None), ('const_speed_init', 0))):
            torch.manual_seed(0); np.random.seed(0)
            slam = SLAM(cfg, ds, os.path.join(tmp, 'out_' + tag), device=DEV, static_shapes=True, verbose=os.environ.get('VERBOSE') == '1')
            fit = slam.prefit_decoders(list(range(0, n, max(n // 6, 1))), iters=int(os.environ.get('PREFIT', 400)))
            res = slam.run(tracking_iters=iters)
            ate = slam.evaluate(res['ckpt'])
            ck = torch.load(res['ckpt'], map_location='cpu', weights_only=False)
            err = (ck['estimate_c2w_list'][:, :3, 3] - ck['gt_c2w_list'][:, :3, 3]).norm(dim=1)
            out[tag] = dict(ate=ate['absolute_translational_error.rmse'], raw_rmse=float((err ** 2).mean().sqrt()), raw_max=float(err.max()),
                            fps=res['fps'], prefit_loss=fit)
            if verbose:
                print(tag, out[tag], flush=True)
    return out


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
