"""CPU restatement (numpy) of the image operators and the loss of the tracker's event term.
TEST INFRASTRUCTURE ONLY -- nothing under evennicer-slam_amd/ may import this.

What it follows:
  * src/Tracker.py:129-137,146: ground-truth event image, mask and previous colour image resized with torchvision
    `transforms.Resize((h, w), InterpolationMode.NEAREST)`;
  * src/utils/Renderer.py:288-291: depth image resized with `transforms.Resize(..., BILINEAR)`;
  * src/Tracker.py:206-228: L2 event loss, Gaussian-blurred L2 terms (`transforms.functional.gaussian_blur`), balancer.

Pinning: torchvision (pinned 0.12 next to pytorch 1.11, environment.yaml) is NOT part of /root/reference and is not
installed in the build image, so these three operators are restated from torchvision's published tensor algorithms:
  resize NEAREST   -> torch `interpolate(mode='nearest')`: src = min(floor(dst * in / out), in - 1), scale in float32;
  resize BILINEAR  -> `interpolate(mode='bilinear', align_corners=False)`, antialias off:
                      x = max((dst + 0.5) * in / out - 0.5, 0), the two neighbours weighted by the fraction;
  gaussian_blur(k) -> sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8; kernel exp(-0.5 (x / sigma)^2) on linspace(-(k-1)/2,
                      (k-1)/2, k), normalised; 2-D kernel = outer product; reflect padding (edge not repeated).
The reference holds no test or fixture for them: PARITY UNPINNED by the reference for these three operators; the
tests cross-check this file against scipy.ndimage (an independent implementation of the same definitions).  The rest
of the event term (pose -> rays -> render -> U-Net -> loss -> camera gradient) is pinned by
tests/golden/tiny_event_iter.npz, produced with the reference's own functions and network."""
import numpy as np


def resize_nearest(img, size):
    """img [..., H, W] -> [..., h, w]."""
    H, W = img.shape[-2:]
    h, w = size
    sy = np.float32(H) / np.float32(h)
    sx = np.float32(W) / np.float32(w)
    iy = np.minimum(np.floor(np.arange(h, dtype=np.float32) * sy).astype(np.int64), H - 1)
    ix = np.minimum(np.floor(np.arange(w, dtype=np.float32) * sx).astype(np.int64), W - 1)
    return img[..., iy[:, None], ix[None, :]]


def _bilinear_axis(n_in, n_out):
    scale = np.float32(n_in) / np.float32(n_out)
    x = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    x = np.maximum(x, np.float32(0))
    i0 = np.minimum(np.floor(x).astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    f = (x - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, f


def resize_bilinear(img, size):
    """float32 img [..., H, W] -> [..., h, w] (align_corners False, no antialias)."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape[-2:]
    y0, y1, fy = _bilinear_axis(H, size[0])
    x0, x1, fx = _bilinear_axis(W, size[1])
    fy = fy[:, None]
    fx = fx[None, :]
    a = img[..., y0[:, None], x0[None, :]]
    b = img[..., y0[:, None], x1[None, :]]
    c = img[..., y1[:, None], x0[None, :]]
    d = img[..., y1[:, None], x1[None, :]]
    one = np.float32(1)
    return ((one - fy) * ((one - fx) * a + fx * b) + fy * ((one - fx) * c + fx * d)).astype(np.float32)


def gaussian_kernel1d(k, sigma=None):
    if sigma is None:
        sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    half = (k - 1) * 0.5
    x = np.linspace(-half, half, k)
    pdf = np.exp(-0.5 * (x / sigma) ** 2)
    return pdf / pdf.sum()


def _reflect_index(i, n):
    """index of the reflect padding that does not repeat the edge sample: -1 -> 1, n -> n - 2"""
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def gaussian_blur(img, k, sigma=None):
    """img [C, H, W] -> blurred [C, H, W], float64 arithmetic."""
    img = np.asarray(img, dtype=np.float64)
    _, H, W = img.shape
    k1 = gaussian_kernel1d(k, sigma)
    r = k // 2
    out = np.zeros_like(img)
    for dy in range(-r, r + 1):
        yy = _reflect_index(np.arange(H) + dy, H)
        for dx in range(-r, r + 1):
            xx = _reflect_index(np.arange(W) + dx, W)
            out += k1[dy + r] * k1[dx + r] * img[:, yy[:, None], xx[None, :]]
    return out


def gaussian_blur_adjoint(g, k, sigma=None):
    """Transpose of gaussian_blur applied to g [C, H, W] (the blur's backward)."""
    g = np.asarray(g, dtype=np.float64)
    _, H, W = g.shape
    k1 = gaussian_kernel1d(k, sigma)
    r = k // 2
    out = np.zeros_like(g)
    for dy in range(-r, r + 1):
        yy = _reflect_index(np.arange(H) + dy, H)
        for dx in range(-r, r + 1):
            xx = _reflect_index(np.arange(W) + dx, W)
            np.add.at(out, (slice(None), yy[:, None], xx[None, :]), k1[dy + r] * k1[dx + r] * g)
    return out


def event_loss(gt_event, full_event, blur=True, kernel_sizes=(9,), kernel_weights=(1.0,), balancer=1.0):
    """Tracker.py:206-228 on [h, w, 2] images.  Returns (loss, d loss / d full_event).  The raw L2 term always has
    weight 1 (the reference's `unblurred_weight` never reaches the optimised loss)."""
    gt = np.asarray(gt_event, dtype=np.float64)
    fe = np.asarray(full_event, dtype=np.float64)
    diff = gt - fe
    loss = (diff ** 2).sum()
    grad = -2.0 * diff
    if blur:
        for k, wk in zip(kernel_sizes, kernel_weights):
            bd = gaussian_blur(diff.transpose(2, 0, 1), k)               # blur is linear: blur(gt) - blur(pred)
            loss += wk * (bd ** 2).sum()
            grad += wk * (-2.0) * gaussian_blur_adjoint(bd, k).transpose(1, 2, 0)
    return loss * balancer, grad * balancer
