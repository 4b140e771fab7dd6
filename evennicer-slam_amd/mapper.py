"""Mapper inner-iteration glue on the device layout.

Reference (src/Mapper.py): per `optimize_map` call the frustum-masked part of every grid becomes a compact leaf
`val_grad = val[mask]` (:343-361) optimised by `torch.optim.Adam` with per-stage learning rates (:396-419, :466-473);
EVERY iteration re-materialises the full grid (`val[mask] = val_grad`, :448-458 -- an index_put over up to 22.8 MB
per grid, whose backward gathers the dense gradient back), steps Adam (:573-575) and writes the result back
(:596-602).

Here the grid lives in the kernels' voxel-major layout for the whole call: the render kernels read it and add their
gradients into a persistent accumulator directly (functional.VoxelMajorGrid), and ONE launch per iteration applies
Adam to the masked voxels of all grids and clears the accumulators (enslam_adam_masked).  No per-iteration layout
conversion, index_put, dense-gradient gather, block marking or clearing.  `write_back()` returns the optimised
voxels to the caller's `[1,32,D,H,W]` tensors (`val[mask] = val_grad`, :596-602)."""
import ctypes

import torch

from . import _lib as L
from .functional import VoxelMajorGrid, _ptr, _require_hip, _stream, note_raw_write

GRID_KEYS = ('grid_coarse', 'grid_middle', 'grid_fine', 'grid_color')


class MaskedGridOptimizer:
    """Adam over the masked voxels of feature grids, state and parameters in voxel-major layout.

    c      : dict key -> float32 [1,32,D,H,W] HIP tensor (the shared map `self.c` of the reference)
    masks  : dict key -> bool tensor [D,H,W] or the reference's channel-repeated [1,32,D,H,W] (Mapper.py:343-346);
             a missing key / None optimises every voxel (frustum_feature_selection off, :330-341)
    keys   : the grids this mapper optimises (coarse mapper: ('grid_coarse',); else middle, fine, color -- :326-328)
    """

    def __init__(self, c, masks=None, keys=('grid_middle', 'grid_fine', 'grid_color'), betas=(0.9, 0.999), eps=1e-8):
        lib = L.lib()
        self.c, self.keys = c, tuple(keys)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.grids, self.mask, self.mask5, self.m, self.v = {}, {}, {}, {}, {}
        masks = masks or {}
        dev = None
        for key in self.keys:
            g = c[key]
            _require_hip(g, key)
            if g.dim() != 5 or g.shape[0] != 1 or g.shape[1] != 32 or g.dtype != torch.float32:
                raise L.EnslamError(f"{key}: expected float32 [1,32,D,H,W], got {tuple(g.shape)} {g.dtype}")
            dev = g.device
            D, H, W = (int(x) for x in g.shape[2:])
            V = D * H * W
            vm = torch.empty((V, 32), dtype=torch.float32, device=dev)
            L.check(lib.enslam_grid_to_voxel_major(_ptr(g.detach().contiguous()), _ptr(vm), V, _stream()),
                    "enslam_grid_to_voxel_major")
            self.grids[key] = VoxelMajorGrid((D, H, W), vm, torch.zeros((V, 32), dtype=torch.float32, device=dev))
            mk = masks.get(key)
            if mk is not None:
                mk = mk.to(dev)
                if mk.dim() == 5:                       # channel-repeated mask of the reference: any channel
                    mk = mk[0, 0]
                if tuple(mk.shape) != (D, H, W):
                    raise L.EnslamError(f"{key}: mask shape {tuple(mk.shape)} does not match the grid {(D, H, W)}")
                self.mask5[key] = mk.bool()
                self.mask[key] = mk.reshape(-1).to(torch.uint8).contiguous()
            else:
                self.mask5[key] = self.mask[key] = None
            self.m[key] = torch.zeros((V, 32), dtype=torch.float32, device=dev)
            self.v[key] = torch.zeros((V, 32), dtype=torch.float32, device=dev)
        n = len(self.keys)
        # device scalars the launch reads: learning rate (float64) and step count (int32) of each grid
        self.lr_t = torch.zeros(n, dtype=torch.float64, device=dev)
        self.step_t = torch.zeros(n, dtype=torch.int32, device=dev)
        self._lr_host = [None] * n
        self.steps = [0] * n                           # host mirror of step_t

    # ---- what the renderer gets in place of the reference's `c`
    def render_grids(self, extra=None):
        """dict to pass as `c` to Renderer.render_batch_ray: optimised grids in the device layout, every other key of
        the caller's `c` (or `extra`) unchanged."""
        out = dict(self.c)
        if extra:
            out.update(extra)
        out.update(self.grids)
        return out

    def set_lr(self, lrs):
        """lrs: dict key -> learning rate for the coming steps (cfg['mapping']['stage'][stage][...]*lr_factor,
        Mapper.py:466-473).  Only a changed value costs a (tiny, stream-ordered) copy."""
        host = [float(lrs.get(k, 0.0)) for k in self.keys]
        if host != self._lr_host:
            self.lr_t.copy_(torch.tensor(host, dtype=torch.float64), non_blocking=False)
            self._lr_host = host

    def step(self, lrs=None):
        """optimizer.step() + optimizer.zero_grad() for the grids (Mapper.py:575, 594): Adam on the masked voxels of
        every grid that has received a gradient so far (torch.optim.Adam skips parameters whose grad is None; the
        stages only ever add grids), then all accumulators are cleared."""
        if lrs is not None:
            self.set_lr(lrs)
        lib = L.lib()
        n = len(self.keys)
        inc = [1 if self.grids[k].has_grad else 0 for k in self.keys]
        if any(inc):
            if all(inc):
                self.step_t += 1
            else:
                self.step_t += torch.tensor(inc, dtype=torch.int32, device=self.step_t.device)
            self.steps = [s + i for s, i in zip(self.steps, inc)]
        arr = lambda: (ctypes.c_void_p * n)()
        p, g, m, v, mk, lr, st = arr(), arr(), arr(), arr(), arr(), arr(), arr()
        vs = (ctypes.c_int64 * n)()
        for i, k in enumerate(self.keys):
            G = self.grids[k]
            p[i], g[i], m[i], v[i] = G.vm.data_ptr(), G.grad_vm.data_ptr(), self.m[k].data_ptr(), self.v[k].data_ptr()
            mk[i] = self.mask[k].data_ptr() if self.mask[k] is not None else None
            vs[i] = G.vm.shape[0]
            lr[i] = self.lr_t.data_ptr() + 8 * i
            st[i] = self.step_t.data_ptr() + 4 * i
        L.check(lib.enslam_adam_masked(n, p, g, m, v, mk, vs, lr, st, self.betas[0], self.betas[1], self.eps, _stream()),
                "enslam_adam_masked")

    def grid(self, key):
        """Current values of one grid as a fresh float32 [1,32,D,H,W] tensor (layout conversion; for inspection)."""
        G = self.grids[key]
        out = torch.empty((1, 32) + G.dims, dtype=torch.float32, device=G.vm.device)
        L.check(L.lib().enslam_grid_from_voxel_major(_ptr(G.vm), _ptr(out), G.vm.shape[0], _stream()),
                "enslam_grid_from_voxel_major")
        return out

    def write_back(self):
        """val[mask] = val_grad (Mapper.py:596-602): optimised voxels into the caller's tensors, in place."""
        with torch.no_grad():
            for k in self.keys:
                new = self.grid(k)
                if self.mask5[k] is None:
                    self.c[k].copy_(new)
                else:
                    self.c[k].copy_(torch.where(self.mask5[k][None, None], new, self.c[k]))
        return self.c


class FusedAdam:
    """torch.optim.Adam (defaults) for a list of small dense parameters -- the decoders the mapper optimises
    (Mapper.py:363-369, group 0 of :409) -- as ONE launch per step; learning rate and step count live on the
    device, so a captured step follows `set_lr` without re-capture.  Parameters whose `.grad` is None are skipped
    only if that holds for all of them (the reference's decoder group always receives gradients together)."""

    def __init__(self, params, lr=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FusedAdam: empty parameter list")
        if len(self.params) > 72:
            raise L.EnslamError("FusedAdam: at most 72 tensors per optimiser")
        dev = self.params[0].device
        for p in self.params:
            _require_hip(p, "parameter")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise L.EnslamError("FusedAdam: parameters must be contiguous float32")
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        sizes = [p.numel() for p in self.params]
        self._m = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self._v = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.exp_avg, self.exp_avg_sq = list(self._m.split(sizes)), list(self._v.split(sizes))
        self.lr_t = torch.full((1,), float(lr), dtype=torch.float64, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self._lr_host = float(lr)

    def set_lr(self, lr):
        if float(lr) != self._lr_host:
            self.lr_t.fill_(float(lr))
            self._lr_host = float(lr)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self, lr=None):
        if lr is not None:
            self.set_lr(lr)
        grads = [p.grad for p in self.params]
        if all(g is None for g in grads):
            return
        n = len(self.params)
        P, G, M, V = ((ctypes.c_void_p * n)() for _ in range(4))
        numel = (ctypes.c_int64 * n)()
        for i, (p, g) in enumerate(zip(self.params, grads)):
            if g is None:
                raise L.EnslamError("FusedAdam.step: some parameters of the group have no gradient")
            g = g if g.is_contiguous() else g.contiguous()
            P[i], G[i], M[i], V[i] = p.data_ptr(), g.data_ptr(), self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr()
            numel[i] = p.numel()
        self.step_t += 1
        # raw kernel writes: tell torch (and the packed-decoder cache) they changed -- now, and on every replay when
        # this call is being captured (graph.GraphedStep.replay)
        note_raw_write(self.params)
        L.check(L.lib().enslam_adam_tensors(n, P, G, M, V, numel, _ptr(self.lr_t), _ptr(self.step_t), self.betas[0],
                                            self.betas[1], self.eps, _stream()), "enslam_adam_tensors")
