#!/usr/bin/env python3
"""Golden fixture for the tracker's camera iteration WITH the event term (measurement config 3):
tests/golden/tiny_event_iter.npz.  Runs only in the build container (needs /root/reference).

src/Tracker.py cannot be imported here (cv2, colorama, wandb, torchvision are absent), so this script executes the
statements of Tracker.optimize_cam_in_batch (:129-232) with the REFERENCE's functions wherever they import:
  src/common.py get_camera_from_tensor / get_samples / get_rays_rescale, src/utils/Renderer.py render_batch_ray (the
  body of render_img_rescale :277-318 is a chunk loop over it), event_net.UNet_2heads, src/event_net.py
  inference_event, and the decoders -- on the tiny scene of make_golden.py.
The three torchvision calls on the way (Resize NEAREST :133-137,146; Resize BILINEAR Renderer.py:288-291;
functional.gaussian_blur :214-215) cannot run (torchvision is not installed and not vendored); they are written out
below from torchvision's published tensor algorithms, in a formulation of their own (gather / shifted slices) so that
the fixture does not simply replay the product's code.  See oracle/event_oracle.py for how those three are checked.

The U-Net weights are not stored (17 M parameters): both sides build the network under torch.manual_seed(UNET_SEED)
on the CPU; the fixture holds the parameter names/shapes, two weight checksums and the outputs for one input."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import make_golden_tracker as MGT  # noqa: E402,F401  (installs the quad2rotation device shim)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from src.common import get_camera_from_tensor, get_samples, get_rays_rescale  # noqa: E402
from src.event_net import inference_event  # noqa: E402
from event_net import UNet_2heads  # noqa: E402

UNET_SEED = 2024


def nearest_resize(img, size):                      # [C,H,W]; torchvision Resize(NEAREST) on a tensor
    H, W = img.shape[-2:]
    iy = torch.clamp((torch.arange(size[0], dtype=torch.float32) * (H / size[0])).floor().long(), max=H - 1)
    ix = torch.clamp((torch.arange(size[1], dtype=torch.float32) * (W / size[1])).floor().long(), max=W - 1)
    return img[..., iy[:, None], ix[None, :]]


def bilinear_resize(img, size):                     # [C,H,W]; torchvision Resize(BILINEAR) on a tensor, no antialias
    H, W = img.shape[-2:]

    def axis(n_in, n_out):
        x = torch.clamp((torch.arange(n_out, dtype=torch.float32) + 0.5) * (n_in / n_out) - 0.5, min=0)
        i0 = torch.clamp(x.floor().long(), max=n_in - 1)
        return i0, torch.clamp(i0 + 1, max=n_in - 1), x - i0.float()
    y0, y1, fy = axis(H, size[0])
    x0, x1, fx = axis(W, size[1])
    fy, fx = fy[:, None], fx[None, :]
    g = lambda yy, xx: img[..., yy[:, None], xx[None, :]]
    return (1 - fy) * ((1 - fx) * g(y0, x0) + fx * g(y0, x1)) + fy * ((1 - fx) * g(y1, x0) + fx * g(y1, x1))


def blur(img, k):                                   # [C,H,W]; torchvision functional.gaussian_blur(kernel_size=k)
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    x = torch.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k)
    k1 = torch.exp(-0.5 * (x / sigma) ** 2)
    k1 = k1 / k1.sum()
    r = k // 2
    C, H, W = img.shape

    def refl(i, n):
        i = i.abs()
        return torch.where(i >= n, 2 * (n - 1) - i, i)
    tmp = torch.zeros_like(img)
    for d in range(-r, r + 1):                      # rows, then columns (separable)
        tmp = tmp + k1[d + r] * img[:, refl(torch.arange(H) + d, H), :]
    out = torch.zeros_like(img)
    for d in range(-r, r + 1):
        out = out + k1[d + r] * tmp[:, :, refl(torch.arange(W) + d, W)]
    return out


def main():
    cfg = MG.tiny_cfg()
    std = {'grid_coarse': 0.3, 'grid_middle': 0.3, 'grid_fine': 0.3, 'grid_color': 0.5}
    model, bound, c = MG.build_scene(cfg, seed=1234, grid_std=std)          # identical to tiny_scene.npz
    cam = dict(H=48, W=64, fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    H, W, fx, fy, cx, cy = (cam[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
    renderer = MG.make_renderer(cfg, bound, cam)
    g = torch.Generator().manual_seed(99)
    gt_depth = torch.rand(H, W, generator=g) * 1.4 + 0.2
    gt_depth[20:24, :] = 0.0
    gt_color = torch.rand(H, W, 3, generator=g)
    pre_gt_color = (gt_color + 0.15 * torch.rand(H, W, 3, generator=g)).clamp(0, 1)
    gt_event = torch.randint(0, 4, (H, W, 2), generator=g).float()
    gt_mask = (gt_event.sum(-1) > 2).long()
    device = 'cpu'
    ev = cfg['event']
    scale_factor = 0.5
    balancer, kernel_sizes, kernel_weights, unblurred_weight = ev['balancer'], ev['kernel_sizes'], ev['kernel_weights'], ev['unblurred_weight']
    assert ev['blur'] and cfg['tracking']['handle_dynamic'] and cfg['tracking']['use_color_in_tracking']
    w_color_loss = cfg['tracking']['w_color_loss']
    Hedge, Wedge = 4, 6
    batch_size = 80
    camera_tensor = torch.tensor([0.98, 0.02, 0.17, -0.03, 0.1, -0.05, 0.2], requires_grad=True)

    torch.manual_seed(UNET_SEED)
    net = UNet_2heads(6, 2, 2)
    net.eval()
    for p in net.parameters():
        p.requires_grad_(False)
    sd = net.state_dict()
    out = {'unet_seed': np.array(UNET_SEED),
           'unet_keys': np.array('\n'.join(f'{k} {tuple(v.shape)}' for k, v in sd.items())),
           'unet_w_first': sd['inc.double_conv.0.weight'].numpy().copy(),
           'unet_w_last': sd['outc_2.conv.weight'].numpy().copy(),
           'unet_b_last': sd['outc_2.conv.bias'].numpy().copy()}
    xin = torch.rand(1, 6, 24, 32, generator=g)
    with torch.no_grad():
        e, m = net(xin)
    out.update(unet_in=xin.numpy(), unet_events=e.numpy(), unet_probs=m.numpy())

    out.update(camera_tensor=camera_tensor.detach().numpy().copy(), gt_depth=gt_depth.numpy(), gt_color=gt_color.numpy(),
               pre_gt_color=pre_gt_color.numpy(), gt_event=gt_event.numpy(), gt_mask=gt_mask.numpy(),
               cam=np.array([H, W, fx, fy, cx, cy]), edge=np.array([Hedge, Wedge]), batch_size=np.array(batch_size),
               w_color_loss=np.array(w_color_loss), seed=np.array(31), scale_factor=np.array(scale_factor),
               balancer=np.array(balancer), kernel_sizes=np.array(kernel_sizes), kernel_weights=np.array(kernel_weights),
               unblurred_weight=np.array(unblurred_weight))

    # ---- Tracker.py:129-137: ground truth at the event resolution
    ge = gt_event.permute(2, 0, 1)
    _, h, w = ge.shape
    h_new, w_new = int(scale_factor * h), int(scale_factor * w)
    gt_event_s = nearest_resize(ge, (h_new, w_new)).permute(1, 2, 0)
    gt_mask_s = nearest_resize(gt_mask[None, :, :], (h_new, w_new)).permute(1, 2, 0)
    # ---- :139
    c2w = get_camera_from_tensor(camera_tensor)
    # ---- :146
    full_color_previous = nearest_resize(pre_gt_color.permute(2, 0, 1), (h_new, w_new)).permute(1, 2, 0)
    # ---- :150 = Renderer.render_img_rescale (Renderer.py:275-318), one chunk
    new_H, new_W = int(H * scale_factor), int(W * scale_factor)
    rays_o, rays_d = get_rays_rescale(H, W, new_H, new_W, fx, fy, cx, cy, c2w, device)
    rays_o = rays_o.reshape(-1, 3)
    rays_d = rays_d.reshape(-1, 3)
    gd_s = bilinear_resize(gt_depth.unsqueeze(0), (new_H, new_W)).reshape(-1)
    depth_i, unc_i, color_i = renderer.render_batch_ray(c, model, rays_d, rays_o, device, 'color', gt_depth=gd_s)
    full_color_current = color_i.reshape(new_H, new_W, 3)
    # ---- :153
    full_event, event_mask = inference_event(net=net, img1=full_color_previous, img2=full_color_current, device=device,
                                             scale_factor=1.0, out_threshold=0.5)
    # ---- :160-185 RGB-D part (handle_dynamic on)
    torch.manual_seed(31)
    idx = torch.randint((H - 2 * Hedge) * (W - 2 * Wedge), (batch_size,))
    out['idx'] = idx.numpy()
    torch.manual_seed(31)
    b_ro, b_rd, b_gd, b_gc = get_samples(Hedge, H - Hedge, Wedge, W - Wedge, batch_size, H, W, fx, fy, cx, cy, c2w,
                                         gt_depth, gt_color, device)
    with torch.no_grad():
        t = (bound.unsqueeze(0).to(device) - b_ro.clone().detach().unsqueeze(-1)) / b_rd.clone().detach().unsqueeze(-1)
        t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
        inside_mask = t >= b_gd
    b_rd, b_ro, b_gd, b_gc = b_rd[inside_mask], b_ro[inside_mask], b_gd[inside_mask], b_gc[inside_mask]
    depth, uncertainty, color = renderer.render_batch_ray(c, model, b_rd, b_ro, device, stage='color', gt_depth=b_gd)
    uncertainty = uncertainty.detach()
    tmp = torch.abs(b_gd - depth) / torch.sqrt(uncertainty + 1e-10)
    mask = (tmp < 10 * tmp.median()) & (b_gd > 0)
    # ---- :187-199
    loss_rgbd = (torch.abs(b_gd - depth) / torch.sqrt(uncertainty + 1e-10))[mask].sum()
    loss_rgbd = loss_rgbd + w_color_loss * torch.abs(b_gc - color)[mask].sum()
    loss_rgbd.backward(retain_graph=True)
    g_rgbd = camera_tensor.grad.numpy().copy()
    camera_tensor.grad.zero_()              # (the reference accumulates; recorded separately to keep g_event exact)
    # ---- :206-229
    loss_event = ((gt_event_s - full_event) ** 2).sum()
    terms = [float(unblurred_weight * loss_event)]
    gts_b, preds_b = [], []
    for k, wk in zip(kernel_sizes, kernel_weights):
        gb = blur(gt_event_s.permute(2, 0, 1), k).permute(1, 2, 0)
        pb = blur(full_event.permute(2, 0, 1), k).permute(1, 2, 0)
        tk = ((gb - pb) ** 2).sum()
        loss_event = loss_event + wk * tk
        gts_b.append(gb)
        preds_b.append(pb)
        terms.append(tk.item())
    loss_mask = torch.nn.CrossEntropyLoss()(event_mask, gt_mask_s.permute(2, 0, 1))
    loss_event = loss_event * balancer
    loss_event.backward()                                                   # :231-232 (activate_events)
    g_event = camera_tensor.grad.numpy().copy()
    g_total = g_rgbd + g_event

    out.update(gt_event_s=gt_event_s.numpy(), gt_mask_s=gt_mask_s.numpy(), full_color_previous=full_color_previous.numpy(),
               gt_depth_s=gd_s.numpy(), full_color_current=full_color_current.detach().numpy(),
               full_event=full_event.detach().numpy(), event_prob=event_mask.detach().numpy(),
               inside_mask=inside_mask.numpy(), dyn_mask=mask.numpy(), depth=depth.detach().numpy(),
               uncertainty=uncertainty.numpy(), color=color.detach().numpy(),
               loss_rgbd=np.array(loss_rgbd.item()), loss_event=np.array(loss_event.item()),
               loss_mask=np.array(loss_mask.item()), loss_terms=np.array(terms),
               gt_event_blur=gts_b[0].numpy(), pred_event_blur=preds_b[0].detach().numpy(),
               g_rgbd=g_rgbd, g_event=g_event, g_total=g_total)
    np.savez_compressed(os.path.join(HERE, 'tiny_event_iter.npz'), **out)
    print('loss rgbd', loss_rgbd.item(), 'event', loss_event.item(), 'mask', loss_mask.item())
    print('g_rgbd', g_rgbd)
    print('g_event', g_event)
    print('rays kept', int(inside_mask.sum()), 'dyn', int(mask.sum()))


if __name__ == '__main__':
    main()
