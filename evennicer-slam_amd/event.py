"""Event term of the tracker's camera iteration (SURVEY.md 8 f2 / measurement config 3).

Reference: src/Tracker.py:129-150 (ground-truth event / mask / previous colour image resized with torchvision's
`Resize(NEAREST)`), :150 (`render_img_rescale` of the current pose, with gradient), :153 + src/event_net.py:67-99
(`inference_event`: the two colour images -> `UNet_2heads(6, 2, 2)` -> events * P(event)), :206-228 (L2 event loss,
optionally on Gaussian-blurred images, scaled by `event.balancer`).

The U-Net is a caller-side network and stays a PyTorch-ROCm module (north_star); it is written here from its
published architecture (5-level U-Net with bilinear up-sampling and two decoder heads) with the parameter names of
the reference checkpoints (event_net/unet_model.py:72-122, event_net/unet_parts.py), so
`pretrained/eventnet_2head_*.pth` loads with `load_state_dict`.  torchvision is not part of this image: the two
torchvision ops on the path are restated on torch primitives --
  Resize(NEAREST) on a tensor   == F.interpolate(mode='nearest')   (source index floor(dst * in/out))
  functional.gaussian_blur(k)   == reflect-pad + depthwise conv with the normalised kernel exp(-x^2 / 2 sigma^2),
                                   sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
and checked against oracle/event_oracle.py (numpy) and scipy.ndimage in tests/test_event_cpu.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# torchvision stand-ins (tensor inputs, layout [..., H, W])
# ------------------------------------------------------------------------------------------------
def resize_nearest(img, size):
    """`transforms.Resize(size, InterpolationMode.NEAREST)` of a [C,H,W] (or [H,W]) tensor."""
    x = img
    lead = x.dim()
    while x.dim() < 4:
        x = x[None]
    need_cast = not x.is_floating_point()
    y = F.interpolate(x.float() if need_cast else x, size=tuple(size), mode='nearest')
    if need_cast:
        y = y.to(img.dtype)
    while y.dim() > lead:
        y = y[0]
    return y


def resize_bilinear(img, size):
    """`transforms.Resize(size, InterpolationMode.BILINEAR)` of a [C,H,W] tensor (no antialias: the pinned
    torchvision applies it to tensors only on request)."""
    x = img
    lead = x.dim()
    while x.dim() < 4:
        x = x[None]
    y = F.interpolate(x, size=tuple(size), mode='bilinear', align_corners=False)
    while y.dim() > lead:
        y = y[0]
    return y


def gaussian_kernel1d(kernel_size, sigma=None, dtype=torch.float32, device=None):
    if sigma is None:
        sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size, dtype=dtype, device=device)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(img, kernel_size, sigma=None):
    """`transforms.functional.gaussian_blur(img, kernel_size)` of a [C,H,W] tensor: separable Gaussian, reflect
    padding, every channel on its own."""
    if kernel_size % 2 != 1 or kernel_size <= 0:
        raise ValueError(f"kernel_size must be odd and positive, got {kernel_size}")
    c, h, w = img.shape
    dt = img.dtype if img.is_floating_point() else torch.float32
    k1 = gaussian_kernel1d(kernel_size, sigma, dtype=dt, device=img.device)
    k2 = torch.outer(k1, k1).expand(c, 1, kernel_size, kernel_size)
    pad = kernel_size // 2
    x = F.pad(img.to(dt)[None], [pad, pad, pad, pad], mode='reflect')
    return F.conv2d(x, k2, groups=c)[0].to(img.dtype)


# ------------------------------------------------------------------------------------------------
# the event network
# ------------------------------------------------------------------------------------------------
class _ConvPair(nn.Module):
    """3x3 conv - BN - ReLU, twice (parameters under `double_conv.{0,1,3,4}`)."""

    def __init__(self, cin, cout, cmid=None):
        super().__init__()
        cmid = cmid or cout
        layers = []
        for a, b in ((cin, cmid), (cmid, cout)):
            layers += [nn.Conv2d(a, b, 3, padding=1, bias=False), nn.BatchNorm2d(b), nn.ReLU(inplace=True)]
        self.double_conv = nn.Sequential(*layers)

    def forward(self, x):
        return self.double_conv(x)


class _Down(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _ConvPair(cin, cout))

    def forward(self, x):
        return self.maxpool_conv(x)


class _Up(nn.Module):
    """x2 up-sampling of the deep feature, centred zero padding to the skip's size, concat [skip, deep], conv pair."""

    def __init__(self, cin, cout, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = _ConvPair(cin, cout, cin // 2)
        else:
            self.up = nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2)
            self.conv = _ConvPair(cin, cout)

    def forward(self, deep, skip):
        deep = self.up(deep)
        dy, dx = skip.shape[2] - deep.shape[2], skip.shape[3] - deep.shape[3]
        if dy or dx:
            deep = F.pad(deep, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([skip, deep], dim=1))


class _Head(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


class UNet_2heads(nn.Module):
    """Shared encoder, two decoders: head 1 regresses the per-polarity event counts, head 2 the probability that a
    pixel fires at all (sigmoid).  `forward(x[B, n_channels, H, W]) -> (events [B, n_classes1, H, W],
    probabilities [B, n_classes2, H, W])` (event_net/unet_model.py:72-122)."""

    WIDTHS = (64, 128, 256, 512, 1024)

    def __init__(self, n_channels, n_classes1, n_classes2, bilinear=True):
        super().__init__()
        self.n_channels, self.n_classes1, self.n_classes2, self.bilinear = n_channels, n_classes1, n_classes2, bilinear
        w = self.WIDTHS
        f = 2 if bilinear else 1
        self.inc = _ConvPair(n_channels, w[0])
        self.down1 = _Down(w[0], w[1])
        self.down2 = _Down(w[1], w[2])
        self.down3 = _Down(w[2], w[3])
        self.down4 = _Down(w[3], w[4] // f)
        for head, ncls in ((1, n_classes1), (2, n_classes2)):         # registration order = the checkpoints' order
            setattr(self, f'up1_{head}', _Up(w[4], w[3] // f, bilinear))
            setattr(self, f'up2_{head}', _Up(w[3], w[2] // f, bilinear))
            setattr(self, f'up3_{head}', _Up(w[2], w[1] // f, bilinear))
            setattr(self, f'up4_{head}', _Up(w[1], w[0], bilinear))
            setattr(self, f'outc_{head}', _Head(w[0], ncls))

    def _decode(self, head, feats):
        x = feats[-1]
        for lvl, skip in zip((1, 2, 3, 4), reversed(feats[:-1])):
            x = getattr(self, f'up{lvl}_{head}')(x, skip)
        return getattr(self, f'outc_{head}')(x)

    def forward(self, x):
        feats = [self.inc(x)]
        for down in (self.down1, self.down2, self.down3, self.down4):
            feats.append(down(feats[-1]))
        return self._decode(1, feats), torch.sigmoid(self._decode(2, feats))


def inference_event(net, img1, img2, device, scale_factor=1, out_threshold=0.5):
    """Predicted event image of the colour pair (img1 = previous, img2 = current; [H,W,3] in [0,1]):
    `(events * P(event))` as [H,W,2] and the probability maps [1,2,H,W] (src/event_net.py:67-99).  Differentiable in
    the images; the network runs in eval mode."""
    net.eval()
    a, b = img1.permute(2, 0, 1), img2.permute(2, 0, 1)
    if a.shape != b.shape:
        raise ValueError('The sizes of the two input images are not the same!')
    if scale_factor != 1.0:
        _, h, w = a.shape
        size = (int(scale_factor * h), int(scale_factor * w))
        if size[0] <= 0 or size[1] <= 0:
            raise ValueError('Scale is too small, resized images would have no pixels')
        a, b = resize_nearest(a, size), resize_nearest(b, size)
    pair = torch.cat((a, b), dim=0)[None].to(device=device, dtype=torch.float32)
    events, probs = net(pair)
    full_events = (events * probs[:, 1][:, None])[0].squeeze().permute(1, 2, 0)
    return full_events, probs


# ------------------------------------------------------------------------------------------------
# the loss
# ------------------------------------------------------------------------------------------------
def event_loss(gt_event, full_event, blur=True, kernel_sizes=(9,), unblurred_weight=0.0, kernel_weights=(1.0,)):
    """Un-balanced event loss of Tracker.py:206-221 on [h,w,2] event images: the L2 distance of the raw images plus
    `kernel_weight` x the L2 distance of their Gaussian-blurred versions.  (`unblurred_weight` only scales the first
    entry of the returned per-term list -- the raw term itself enters the loss with weight 1, exactly like the
    reference.)  Returns (loss, gts_blurred, preds_blurred, term_values)."""
    loss = ((gt_event - full_event) ** 2).sum()
    gts, preds, terms = [], [], [unblurred_weight * loss]
    if blur:
        for k, wk in zip(kernel_sizes, kernel_weights):
            g = gaussian_blur(gt_event.permute(2, 0, 1), k).permute(1, 2, 0)
            p = gaussian_blur(full_event.permute(2, 0, 1), k).permute(1, 2, 0)
            t = ((g - p) ** 2).sum()
            loss = loss + wk * t
            gts.append(g)
            preds.append(p)
            terms.append(t)
    return loss, gts, preds, terms
