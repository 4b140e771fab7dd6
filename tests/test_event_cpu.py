"""CPU: the event term's image operators, loss and network (evennicer-slam_amd/event.py) against
oracle/event_oracle.py, scipy.ndimage and the fixture tests/golden/tiny_event_iter.npz (reference network + the
golden script's own formulation of the three torchvision operators)."""
import numpy as np
import pytest
import torch

from oracle import event_oracle as O
from tests.util import load, rel_err


@pytest.fixture(scope="module")
def fx():
    return load("tiny_event_iter")


def test_oracle_image_ops_match_fixture(fx):
    size = (24, 32)
    assert np.array_equal(O.resize_nearest(fx['gt_event'].transpose(2, 0, 1), size).transpose(1, 2, 0), fx['gt_event_s'])
    assert np.array_equal(O.resize_nearest(fx['gt_mask'][None], size).transpose(1, 2, 0), fx['gt_mask_s'])
    assert np.array_equal(O.resize_nearest(fx['pre_gt_color'].transpose(2, 0, 1), size).transpose(1, 2, 0),
                          fx['full_color_previous'])
    assert np.abs(O.resize_bilinear(fx['gt_depth'][None], size).reshape(-1) - fx['gt_depth_s']).max() <= 1e-6
    b = O.gaussian_blur(fx['gt_event_s'].transpose(2, 0, 1), 9).transpose(1, 2, 0)
    assert rel_err(b, fx['gt_event_blur']) <= 1e-6


def test_oracle_blur_matches_scipy():
    from scipy import ndimage
    rng = np.random.default_rng(0)
    for (h, w, k) in ((11, 17, 9), (24, 32, 5), (9, 9, 3), (30, 7, 7)):
        img = rng.standard_normal((2, h, w))
        k1 = O.gaussian_kernel1d(k)
        ref = ndimage.correlate1d(ndimage.correlate1d(img, k1, axis=1, mode='mirror'), k1, axis=2, mode='mirror')
        assert np.abs(O.gaussian_blur(img, k) - ref).max() <= 1e-12
        # adjoint: <blur(x), y> == <x, blur^T(y)>
        y = rng.standard_normal((2, h, w))
        assert abs((O.gaussian_blur(img, k) * y).sum() - (img * O.gaussian_blur_adjoint(y, k)).sum()) <= 1e-9
    assert abs(O.gaussian_kernel1d(9).sum() - 1) < 1e-15
    # sigma of torchvision's default: 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    x = np.linspace(-4, 4, 9)
    e = np.exp(-0.5 * (x / 1.7) ** 2)
    assert np.allclose(O.gaussian_kernel1d(9), e / e.sum(), rtol=0, atol=1e-15)


def test_oracle_resize_definitions():
    # nearest: floor(dst * in / out); 680 -> 102 and 1200 -> 180 are the Replica sizes at scale 0.15
    a = np.arange(680 * 3).reshape(1, 680, 3)
    got = O.resize_nearest(a, (102, 3))[0, :, 0] // 3
    # (float32 arithmetic like ATen's nearest kernel: 3 * 6.6666665f rounds to 20.0f although 3 * 680 / 102 = 19.99...)
    assert np.array_equal(got, np.floor(np.arange(102, dtype=np.float32) * np.float32(680 / 102)).astype(np.int64))
    assert got[3] == 20 and got[101] == 673
    # bilinear, align_corners False: constant and linear ramps are preserved away from the clamped border
    ramp = np.tile(np.arange(40, dtype=np.float32), (1, 6, 1))
    out = O.resize_bilinear(ramp, (6, 20))
    assert np.allclose(out[0, 0, 1:-1], (np.arange(20)[1:-1] + 0.5) * 2 - 0.5, atol=1e-5)
    assert np.allclose(O.resize_bilinear(np.full((1, 9, 13), 2.5, np.float32), (4, 5)), 2.5)


def test_product_image_ops_match_oracle():
    from evennicer_slam_amd import event as EV
    g = torch.Generator().manual_seed(1)
    for (H, W, h, w) in ((48, 64, 24, 32), (680, 1200, 102, 180), (260, 346, 39, 51), (17, 23, 5, 7)):
        img = torch.rand(3, H, W, generator=g)
        assert np.array_equal(EV.resize_nearest(img, (h, w)).numpy(), O.resize_nearest(img.numpy(), (h, w)))
        lab = torch.randint(0, 2, (1, H, W), generator=g)
        r = EV.resize_nearest(lab, (h, w))
        assert r.dtype == lab.dtype and np.array_equal(r.numpy(), O.resize_nearest(lab.numpy(), (h, w)))
        assert np.abs(EV.resize_bilinear(img, (h, w)).numpy() - O.resize_bilinear(img.numpy(), (h, w))).max() <= 2e-6
    for k in (3, 5, 9):
        img = torch.randn(2, 24, 32, generator=g)
        assert np.abs(EV.gaussian_blur(img, k).numpy() - O.gaussian_blur(img.numpy(), k)).max() <= 2e-6
    with pytest.raises(ValueError):
        EV.gaussian_blur(torch.zeros(2, 8, 8), 4)


def test_product_event_loss_and_gradient_match_oracle(fx):
    from evennicer_slam_amd import event as EV
    gt = torch.from_numpy(fx['gt_event_s'])
    fe = torch.from_numpy(fx['full_event']).double().requires_grad_(True)
    loss, gts, preds, terms = EV.event_loss(gt.double(), fe, True, [9], 0.0, [1.0])
    (loss * float(fx['balancer'])).backward()
    want, wgrad = O.event_loss(fx['gt_event_s'], fx['full_event'], True, [9], [1.0], float(fx['balancer']))
    assert abs(loss.item() * float(fx['balancer']) - want) <= 1e-10 * abs(want)
    assert rel_err(fe.grad.numpy(), wgrad) <= 1e-10
    # the fixture's numbers (reference network, golden script's blur)
    assert abs(want - float(fx['loss_event'])) <= 1e-5 * abs(float(fx['loss_event']))
    assert rel_err(preds[0].detach().numpy(), fx['pred_event_blur']) <= 1e-5
    assert abs(float(terms[1]) - float(fx['loss_terms'][1])) <= 1e-5 * float(fx['loss_terms'][1])
    # blur off: the plain L2 distance
    l2, _, _, t2 = EV.event_loss(gt.double(), fe.detach(), False)
    assert abs(l2.item() - ((fx['gt_event_s'].astype(np.float64) - fx['full_event']) ** 2).sum()) <= 1e-9 * l2.item()
    assert len(t2) == 1


def test_unet_parameter_names_init_and_forward_match_reference(fx):
    from evennicer_slam_amd import event as EV
    torch.manual_seed(int(fx['unet_seed']))
    net = EV.UNet_2heads(6, 2, 2)
    net.eval()
    sd = net.state_dict()
    keys = '\n'.join(f'{k} {tuple(v.shape)}' for k, v in sd.items())
    assert keys == str(fx['unet_keys'])                       # names, shapes and ORDER of the reference checkpoints
    # same construction order -> same RNG consumption -> identical seeded weights
    assert np.array_equal(sd['inc.double_conv.0.weight'].numpy(), fx['unet_w_first'])
    assert np.array_equal(sd['outc_2.conv.weight'].numpy(), fx['unet_w_last'])
    assert np.array_equal(sd['outc_2.conv.bias'].numpy(), fx['unet_b_last'])
    with torch.no_grad():
        e, m = net(torch.from_numpy(fx['unet_in']))
    assert rel_err(e.numpy(), fx['unet_events']) <= 1e-5
    assert rel_err(m.numpy(), fx['unet_probs']) <= 1e-5
    # odd sizes: the up path pads to the skip connection (102 x 180 -> 51 x 90 -> 25 x 45 -> 12 x 22 -> 6 x 11)
    with torch.no_grad():
        e, m = net(torch.zeros(1, 6, 22, 38))
    assert e.shape == (1, 2, 22, 38) and m.shape == (1, 2, 22, 38)


def test_inference_event_matches_fixture(fx):
    from evennicer_slam_amd import event as EV
    torch.manual_seed(int(fx['unet_seed']))
    net = EV.UNet_2heads(6, 2, 2)
    a = torch.from_numpy(fx['full_color_previous'])
    b = torch.from_numpy(fx['full_color_current'])
    with torch.no_grad():
        fe, probs = EV.inference_event(net, a, b, 'cpu', scale_factor=1.0)
    assert not net.training
    assert rel_err(fe.numpy(), fx['full_event']) <= 1e-5
    assert rel_err(probs.numpy(), fx['event_prob']) <= 1e-5
    with pytest.raises(ValueError):
        EV.inference_event(net, a, b[:-1], 'cpu')
