"""The tracker-iteration oracle against the fixture produced by the reference's own statements; the host-side
camera-tensor helpers of the product (pure torch, device independent) against the same fixture.  CPU only."""
import numpy as np
import torch

from tests.util import load, rel_err, tiny_scene


def _inputs():
    g = load("tiny_tracker_iter")
    params, grids, bound, s = tiny_scene()
    H, W, fx, fy, cx, cy = g['cam']
    cam = (int(H), int(W), float(fx), float(fy), float(cx), float(cy))
    edge = (int(g['edge'][0]), int(g['edge'][1]))
    return g, params, grids, bound, cam, edge


def test_tracker_iteration_matches_reference_statements():
    from oracle import tracker_oracle as T
    g, params, grids, bound, cam, edge = _inputs()
    ct = torch.from_numpy(g['camera_tensor']).requires_grad_(True)
    torch.manual_seed(int(g['seed']))                                    # the oracle makes the same single RNG draw
    loss, depth, var, color, inside = T.camera_iteration(params, grids, bound, ct, torch.from_numpy(g['gt_depth']),
                                                         torch.from_numpy(g['gt_color']), cam, edge, int(g['batch_size']),
                                                         float(g['w_color_loss']))
    assert np.array_equal(inside.numpy(), g['inside_mask'])
    assert rel_err(depth.detach().numpy(), g['depth']) <= 1e-6
    assert rel_err(color.detach().numpy(), g['color']) <= 1e-6
    assert abs(loss.item() - float(g['loss'])) <= 1e-6 * abs(float(g['loss']))
    loss.backward()
    assert rel_err(ct.grad.numpy(), g['g_camera_tensor']) <= 1e-5


def test_product_camera_helpers_match_reference_outputs():
    import evennicer_slam_amd as E
    C = E.common
    g = load("tiny_tracker_iter")
    ct = torch.from_numpy(g['camera_tensor'])
    c2w = C.get_camera_from_tensor(ct)
    assert np.array_equal(c2w.numpy(), g['c2w'])                          # same float32 expressions
    back = C.get_tensor_from_camera(c2w)
    n = ct[:4] / ct[:4].norm()                                            # a pose determines the quaternion up to scale
    assert torch.allclose(back[:4], n, atol=1e-6) and torch.allclose(back[4:], ct[4:], atol=1e-7)
    batch = torch.stack([ct, ct * torch.tensor([2.0, 2.0, 2.0, 2.0, 1.0, 1.0, 1.0])])
    RT = C.get_camera_from_tensor(batch)
    assert torch.allclose(RT[0], RT[1], atol=1e-6)                        # rotation is scale invariant in the quaternion
