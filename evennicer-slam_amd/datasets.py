"""Dataset readers of the run harness: the reference's `Replica` / `Replica_event` layouts
(src/utils/datasets.py:51-216) read with PIL + numpy instead of cv2 (absent from the image).

    <input_folder>/results/frame*.jpg   colour
    <input_folder>/results/depth*.png   16-bit depth, metres = value / cam.png_depth_scale
    <input_folder>/traj.txt             one row-major 4x4 camera-to-world per line (OpenCV axes; y and z columns flipped
                                        on load, :133-134)
    <event_folder>/*frame*.png          integrated events between frame i-1 and i: RGB png, channels (0, -, +)

Same item tuples, dtypes and conventions as the reference: colour float64 [H,W,3] in [0,1] (`cv2.imread / 255.`),
depth float32 [H,W] * scale, events uint8 [H,W,2] = (-, +) with an all-zero image for frame 0, mask int64 [H,W], pose
float32 [4,4] with the translation scaled IN PLACE on every access (the reference does the same, :112-113).

`RPG` / `RPG_event` (src/utils/datasets.py:218-319; BASELINE config 5's sequence format): grey-scale frames
`results/frame*`, 16-bit depth `results/depth*`, event pngs `<event_folder>/*.png` with channels (+, -, 0), and a lens
model (`cam.distortion` = k1, k2, p1, p2, k3, k4, k5, k6 of OpenCV) that the reference removes from the colour and event images
-- not from the depth -- with `cv2.undistort(img, K, dist)` (:85-88, :262-266).  `undistort()` below restates that call in
numpy: the inverse map of OpenCV's pinhole + rational-radial + tangential model evaluated per destination pixel, bilinear
sampling with a zero border.  cv2 interpolates 8-bit images in fixed point (1/32 pixel, 15-bit weights); this version
interpolates in float64 and rounds once, so single pixels may differ from cv2's by one grey level.  cv2 is absent from the
image, so that last level is "parity unpinned"; the geometry is pinned by a distort -> undistort round trip
(tests/test_harness_cpu.py)."""
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F


def _imread_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"))


def _imread_depth(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im)                                     # 16-bit png ('I;16') -> uint16 / int32


def _resize_bilinear(img, size_hw):
    """cv2.resize(img, (W, H)) default INTER_LINEAR (half-pixel centres, no antialiasing) for float or uint8 HxWxC."""
    if img.shape[0] == size_hw[0] and img.shape[1] == size_hw[1]:
        return img
    t = torch.from_numpy(np.ascontiguousarray(img)).double().permute(2, 0, 1)[None]
    out = F.interpolate(t, size=size_hw, mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
    return out if img.dtype.kind == 'f' else np.clip(np.rint(out), 0, 255).astype(img.dtype)


def get_dataset(cfg, args, scale, device='cuda:0'):
    return dataset_dict[cfg['dataset']](cfg, args, scale, device=device)


class BaseDataset(torch.utils.data.Dataset):
    def __init__(self, cfg, args, scale, device='cuda:0'):
        super().__init__()
        self.name = cfg['dataset']
        self.device = device
        self.scale = scale
        cam = cfg['cam']
        self.png_depth_scale = cam['png_depth_scale']
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy']
        self.distortion = np.asarray(cam['distortion'], dtype=np.float64) if cam.get('distortion') is not None else None
        self.crop_size = cam.get('crop_size')
        self.input_folder = cfg['data']['input_folder'] if getattr(args, 'input_folder', None) is None else args.input_folder
        self.crop_edge = cam['crop_edge']

    def __len__(self):
        return self.n_img

    def _color_depth(self, index):
        color = _imread_rgb(self.color_paths[index])
        if self.distortion is not None:                             # only the colour image, not the depth (:84-88)
            color = undistort(color, (self.fx, self.fy, self.cx, self.cy), self.distortion)
        color = color / 255.
        depth = _imread_depth(self.depth_paths[index]).astype(np.float32) / self.png_depth_scale
        H, W = depth.shape
        color = torch.from_numpy(_resize_bilinear(color, (H, W)))
        depth = torch.from_numpy(depth) * self.scale
        return color, depth

    def _crop(self, color, depth, event=None):
        if self.crop_size is not None:                              # :95-103 ("actually is resize")
            color = F.interpolate(color.permute(2, 0, 1)[None], self.crop_size, mode='bilinear', align_corners=True)[0]
            depth = F.interpolate(depth[None, None], self.crop_size, mode='nearest')[0, 0]
            color = color.permute(1, 2, 0).contiguous()
            if event is not None:
                event = F.interpolate(event.permute(2, 0, 1)[None].float(), self.crop_size, mode='bilinear', align_corners=True)[0]
                event = event.permute(1, 2, 0).contiguous()
        e = self.crop_edge
        if e > 0:
            color, depth = color[e:-e, e:-e], depth[e:-e, e:-e]
            if event is not None:
                event = event[e:-e, e:-e]
        return color, depth, event

    def _pose(self, index):
        pose = self.poses[index]
        pose[:3, 3] *= self.scale
        return pose

    def __getitem__(self, index):
        color, depth = self._color_depth(index)
        color, depth, _ = self._crop(color, depth)
        return index, color.to(self.device), depth.to(self.device), self._pose(index).to(self.device)


class Replica(BaseDataset):
    def __init__(self, cfg, args, scale, device='cuda:0'):
        super().__init__(cfg, args, scale, device)
        self.color_paths = sorted(glob.glob(f'{self.input_folder}/results/frame*.jpg'))
        self.depth_paths = sorted(glob.glob(f'{self.input_folder}/results/depth*.png'))
        self.n_img = len(self.color_paths)
        self.load_poses(f'{self.input_folder}/traj.txt')

    def load_poses(self, path):
        self.poses = []
        with open(path, "r") as f:
            lines = f.readlines()
        for i in range(self.n_img):
            c2w = np.array(list(map(float, lines[i].split()))).reshape(4, 4)
            c2w[:3, 1] *= -1
            c2w[:3, 2] *= -1
            self.poses.append(torch.from_numpy(c2w).float())


class Replica_event(Replica):
    def __init__(self, cfg, args, scale, device='cuda:0'):
        super().__init__(cfg, args, scale, device)
        self.event_folder = cfg['data']['event_folder'] if getattr(args, 'event_folder', None) is None else args.event_folder
        self.event_paths = sorted(glob.glob(f'{self.event_folder}/*frame*.png'))
        self.n_event = len(self.event_paths)
        assert self.n_event == self.n_img - 1, "Number of GT events does not match that of GT images!"

    def __getitem__(self, index):
        color, depth = self._color_depth(index)
        H, W = depth.shape
        if index >= 1:
            event = _imread_rgb(self.event_paths[index - 1])        # png (0, -, +)
        else:
            event = np.zeros((H, W, 3), dtype=np.uint8)             # all black for the first frame
        event = torch.from_numpy(_resize_bilinear(event, (H, W)))
        color, depth, event = self._crop(color, depth, event)
        event = event[:, :, 1:]                                     # (-, +)
        mask = torch.any(event != 0, dim=-1) * 1
        return (index, color.to(self.device), depth.to(self.device), event.to(self.device), mask.to(self.device),
                self._pose(index).to(self.device))


def distort_points(x, y, dist):
    """OpenCV's lens model on normalised image coordinates: (x, y) -> (x_d, y_d) with dist = (k1, k2, p1, p2[, k3[, k4, k5, k6]])."""
    d = np.zeros(8, dtype=np.float64)
    d[:min(len(dist), 8)] = np.asarray(dist, dtype=np.float64)[:8]
    k1, k2, p1, p2, k3, k4, k5, k6 = d
    r2 = x * x + y * y
    radial = (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) / (1 + r2 * (k4 + r2 * (k5 + r2 * k6)))
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return xd, yd


def undistort(img, K, dist):
    """`cv2.undistort(img, K, dist)` (new camera matrix = K) for an HxW or HxWxC array: every destination pixel (u, v) reads
    the source at the DISTORTED position of its ray -- u' = fx x_d + cx, v' = fy y_d + cy with (x_d, y_d) =
    distort_points((u - cx) / fx, (v - cy) / fy) -- by bilinear interpolation, zeros outside the image.  uint8 in -> uint8 out
    (rounded once), float in -> float64 out."""
    fx, fy, cx, cy = K
    a = np.asarray(img)
    H, W = a.shape[:2]
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    xd, yd = distort_points((u - cx) / fx, (v - cy) / fy, dist)
    mx, my = fx * xd + cx, fy * yd + cy
    x0, y0 = np.floor(mx).astype(np.int64), np.floor(my).astype(np.int64)
    ax, ay = mx - x0, my - y0
    src = a.reshape(H, W, -1).astype(np.float64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)] * ok[..., None]

    out = ((1 - ay) * (1 - ax))[..., None] * tap(y0, x0) + ((1 - ay) * ax)[..., None] * tap(y0, x0 + 1) + \
          (ay * (1 - ax))[..., None] * tap(y0 + 1, x0) + (ay * ax)[..., None] * tap(y0 + 1, x0 + 1)
    out = out.reshape(a.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8) if a.dtype == np.uint8 else out


class RPG(BaseDataset):
    """src/utils/datasets.py:218-240: the Replica layout with any image extension."""

    def __init__(self, cfg, args, scale, device='cuda:0'):
        super().__init__(cfg, args, scale, device)
        self.color_paths = sorted(glob.glob(f'{self.input_folder}/results/frame*'))
        self.depth_paths = sorted(glob.glob(f'{self.input_folder}/results/depth*'))
        self.n_img = len(self.color_paths)
        Replica.load_poses(self, f'{self.input_folder}/traj.txt')


class RPG_event(RPG):
    """src/utils/datasets.py:242-319: grey-scale frames (read as grey, replicated to three channels), event pngs with channels
    (+, -, 0) handed out as (-, +) like Replica_event, the lens model removed from colour and events."""

    def __init__(self, cfg, args, scale, device='cuda:0'):
        super().__init__(cfg, args, scale, device)
        self.event_folder = cfg['data']['event_folder'] if getattr(args, 'event_folder', None) is None else args.event_folder
        self.event_paths = sorted(glob.glob(f'{self.event_folder}/*.png'))
        self.n_event = len(self.event_paths)
        assert self.n_event == self.n_img - 1, "Number of GT events does not match that of GT images!"

    def __getitem__(self, index):
        from PIL import Image
        with Image.open(self.color_paths[index]) as im:
            grey = np.array(im.convert('L'))                        # cv2.IMREAD_GRAYSCALE, then GRAY2BGR (:252-253)
        color = np.repeat(grey[:, :, None], 3, axis=2)
        depth = _imread_depth(self.depth_paths[index]).astype(np.float32) / self.png_depth_scale
        event = _imread_rgb(self.event_paths[index - 1]) if index >= 1 else np.zeros_like(color)    # RGB = (+, -, 0)
        if self.distortion is not None:                             # colour and events, not the depth (:262-266)
            K = (self.fx, self.fy, self.cx, self.cy)
            color, event = undistort(color, K, self.distortion), undistort(event, K, self.distortion)
        H, W = depth.shape
        color = torch.from_numpy(_resize_bilinear(color / 255., (H, W)))
        depth = torch.from_numpy(depth) * self.scale
        event = torch.from_numpy(_resize_bilinear(event, (H, W)))
        color, depth, event = self._crop(color, depth, event)
        event = event[:, :, :-1]                                    # (+, -)
        event = event[:, :, [1, 0]]                                 # (-, +) as in Replica_event (:309-310)
        mask = torch.any(event != 0, dim=-1) * 1
        return (index, color.to(self.device), depth.to(self.device), event.to(self.device), mask.to(self.device),
                self._pose(index).to(self.device))


dataset_dict = {"replica": Replica, "replica_event": Replica_event, "rpg": RPG, "rpg_event": RPG_event}


def write_rpg_event_sequence(root, frames, poses, png_depth_scale, events):
    """Write a sequence in the RPG_event layout (tests): frames = list of (grey uint8 [H,W], depth float32 [H,W] metres),
    events = list (n-1) of uint8 [H,W,2] (-, +); written as png (+, -, 0).  Returns (input_folder, event_folder)."""
    from PIL import Image
    inp, evf = os.path.join(root, 'seq'), os.path.join(root, 'seq_events')
    os.makedirs(os.path.join(inp, 'results'), exist_ok=True)
    os.makedirs(evf, exist_ok=True)
    with open(os.path.join(inp, 'traj.txt'), 'w') as f:
        for i, ((grey, depth), pose) in enumerate(zip(frames, poses)):
            Image.fromarray(np.asarray(grey, dtype=np.uint8), 'L').save(os.path.join(inp, 'results', f'frame{i:06d}.png'))
            d16 = np.clip(np.rint(np.asarray(depth, dtype=np.float64) * png_depth_scale), 0, 65535).astype(np.uint16)
            Image.fromarray(d16).save(os.path.join(inp, 'results', f'depth{i:06d}.png'))
            p = np.array(pose, dtype=np.float64).reshape(4, 4).copy()
            p[:3, 1] *= -1
            p[:3, 2] *= -1
            f.write(' '.join(repr(float(v)) for v in p.reshape(-1)) + '\n')
    for i, ev in enumerate(events):
        rgb = np.zeros(ev.shape[:2] + (3,), dtype=np.uint8)
        rgb[..., 0], rgb[..., 1] = ev[..., 1], ev[..., 0]
        Image.fromarray(rgb, 'RGB').save(os.path.join(evf, f'event{i + 1:06d}.png'))
    return inp, evf


def write_replica_event_sequence(root, frames, poses, png_depth_scale, events=None):
    """Write a sequence in the layout above (synthetic data for tests and the harness's smoke run).
    frames: list of (color float [H,W,3] in [0,1], depth float32 [H,W] metres); poses: list of [4,4] camera-to-world IN THE
    READER'S convention (the y / z column flip of load_poses is undone here); events: list (n-1) of uint8 [H,W,2] (-, +).
    Colour is written as PNG-quality JPEG (quality 100, no chroma subsampling).  Returns (input_folder, event_folder)."""
    from PIL import Image
    inp, evf = os.path.join(root, 'seq'), os.path.join(root, 'seq_events')
    os.makedirs(os.path.join(inp, 'results'), exist_ok=True)
    os.makedirs(evf, exist_ok=True)
    with open(os.path.join(inp, 'traj.txt'), 'w') as f:
        for i, ((color, depth), pose) in enumerate(zip(frames, poses)):
            c8 = np.clip(np.rint(np.asarray(color, dtype=np.float64) * 255.), 0, 255).astype(np.uint8)
            Image.fromarray(c8, 'RGB').save(os.path.join(inp, 'results', f'frame{i:06d}.jpg'), quality=100, subsampling=0)
            d16 = np.clip(np.rint(np.asarray(depth, dtype=np.float64) * png_depth_scale), 0, 65535).astype(np.uint16)
            Image.fromarray(d16).save(os.path.join(inp, 'results', f'depth{i:06d}.png'))
            p = np.array(pose, dtype=np.float64).reshape(4, 4).copy()
            p[:3, 1] *= -1
            p[:3, 2] *= -1
            f.write(' '.join(repr(float(v)) for v in p.reshape(-1)) + '\n')
    if events is not None:
        for i, ev in enumerate(events):
            rgb = np.zeros(ev.shape[:2] + (3,), dtype=np.uint8)
            rgb[..., 1:] = ev
            Image.fromarray(rgb, 'RGB').save(os.path.join(evf, f'event_frame{i + 1:06d}.png'))
    return inp, evf
