"""Host side of the HIP rendering path: tensor plumbing around the C ABI (include/enslam_hip.h).

PyTorch is used for device memory, streams and autograd bookkeeping only; every arithmetic step of the
path (sampling, gather, decoders, compositing and their backward) runs in libenslam_hip.so.
There is no CPU implementation here: non-HIP tensors raise.
"""
import ctypes
import os
import weakref

import torch

from . import _lib as L

_KIND_OF_GRID = {'grid_coarse': 0, 'grid_middle': 1, 'grid_fine': 2, 'grid_color': 3}


def _require_hip(t, what):
    if not t.is_cuda:
        raise L.EnslamError(f"{what} must live on a HIP device (got {t.device}); the rendering path has no CPU "
                            f"fallback")


def engine_on_calling_thread():
    """Context manager: run the backward passes started inside it on the calling thread instead of the autograd engine's
    per-device worker thread.  For a Python-implemented autograd.Function the worker thread costs two thread switches and a
    GIL hand-over per call (0.25-0.3 ms of a 0.9 ms Python-driven step, tools/hostprof.py); results are identical.  The
    switch is scoped: PyTorch's setting is restored on exit and nothing outside the block changes (round 3 flipped it
    process-wide in Renderer.__init__).  `ENSLAM_KEEP_AUTOGRAD_THREADS=1` makes this a no-op.

        with EF.engine_on_calling_thread():
            loss.backward()
    """
    if os.environ.get('ENSLAM_KEEP_AUTOGRAD_THREADS') == '1':
        import contextlib
        return contextlib.nullcontext()
    return torch.autograd.set_multithreading_enabled(False)


# ---- tensors another PROCESS writes ---------------------------------------------------------------------------------
# The cached device-side forms (voxel-major copies of contiguous grids, packed decoders) are keyed on (tensor identity,
# _version).  `_version` is a counter of THIS process's tensor object: an in-place update made by another process on the same
# device memory (the reference shares its map and decoders between three processes over CUDA IPC: EvenNICER_SLAM.py:75-95,
# 313-332) does not move it.  The reference's own flow never renders such memory directly -- the tracker clones the map and
# deep-copies the decoders every frame (Tracker.py:248-260), the mappers render the tensors they themselves update
# (Mapper.py:633-641) -- and is safe as it stands.  A caller that DOES render straight from memory another process writes
# declares it: `external_writers(True)` (or ENSLAM_EXTERNAL_WRITERS=1) makes every version-keyed cache miss, so each call
# re-converts the touched blocks and re-packs the decoders from what the memory holds now (channels_last_3d grids are read in
# place and are current either way).  tests/test_hip_multiprocess.py exercises all of it across spawned processes.
_external = [os.environ.get('ENSLAM_EXTERNAL_WRITERS') == '1', 0]


def external_writers(on=True):
    """Declare that tensors handed to the renderer may be modified in place by other processes (see above); returns the
    previous setting."""
    prev, _external[0] = _external[0], bool(on)
    return prev


def _ver(t):
    """Version of tensor t for cache keys: its `_version`, or -- with external writers declared -- a value no cache has seen."""
    if _external[0]:
        _external[1] += 1
        return -_external[1]
    return t._version


def _live_grid_guard(grids):
    """[(grid, version)] of the channels_last_3d grids a render call reads IN PLACE: its backward reads the same storage again
    (corner re-gather of the ray gradients, the recompute route), through a detached alias autograd does not version-check."""
    return [(g, g._version) for g in grids if is_native_grid(g)]


def _check_live_grids(guard):
    """Plain torch raises when a tensor saved for backward has been modified in place; the in-place read of a native grid gets
    the same protection: an optimiser step / MaskedGridOptimizer.step / any in-place write to the map between a render call
    and its backward would silently give ray and grid gradients of a different map."""
    for g, v in guard:
        if g._version != v:
            raise L.EnslamError("a feature grid read in place by this render call (channels_last_3d layout) was modified in place "
                                f"between the call and its backward (version {v} -> {g._version}): backpropagate before updating "
                                "the map, or render from a clone")


def _f32c(t):
    """detached contiguous float32 form of t (no copies, no dispatcher calls beyond detach() when it already is one)."""
    t = t.detach()
    if t.dtype is not torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _stream():
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _ptr(t):
    """Device address of t as a plain int (None -> NULL): ctypes converts both for c_void_p parameters, and an int costs a
    quarter of a c_void_p object (a colour-stage step passes ~180 addresses)."""
    return t.data_ptr() if t is not None else None


def _ptrv(t):
    return t.data_ptr() if t is not None else 0


# ------------------------------------------------------------------------------------------------
# hipGraph capture bookkeeping (graph.GraphedStep)
# ------------------------------------------------------------------------------------------------
# Cached device-side forms (voxel-major grids, packed decoders) that are CREATED while a stream capture is recording live
# in the graph's private memory pool and only hold data once that graph has been replayed; every replay rewrites them
# with the values its inputs had at the START of the replay.  They are therefore tagged with the capture they were made
# in and are invisible to everything outside that capture (an eager render after replays re-packs from the live tensors).
_capture = {'epoch': 0, 'active': 0, 'raw_writes': None, 'init_zero': None, 'persist': None, 'persist_need': {}}

# Persistent dense gradients of captured steps (see _RenderFn.backward): data_ptr of the gradient buffer -> the uint8 flags of
# the 64-voxel blocks that hold non-zeros (`prev` of enslam_step_finish_rays_prev).  The finish launch of the next replay
# rewrites exactly the blocks flagged here or touched by its own rays, so EVERY other writer of such a buffer has to add the
# blocks it fills to these flags: parallel.allreduce_gradients (the sums of the other ranks' blocks) does through
# note_foreign_blocks().  Entries live as long as the GraphedStep that captured them (graph.GraphedStep.close / __del__).
# An address alone does not identify a buffer (the allocator hands a freed address out again): an entry is
# data_ptr -> (the buffer's STORAGE, flags).  The entry holds the storage (so its address cannot be handed out again while the
# entry lives) but not the tensor: a second reference to the tensor would make AccumulateGrad clone the gradient instead of
# adopting it, and the `.grad` that callers pass back in is a detached alias (another tensor object on the same storage)
# anyway.  A look-up compares the storage; a graph that goes away removes only the entries that are still its own
# (_persist_register / forget_persistent).
_persist_prev = {}


def _persist_register(g, pv):
    _persist_prev[g.data_ptr()] = (g.untyped_storage(), pv)


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _cap_tag():
    """0 outside a capture, else the id of the capture being recorded.  (graph.GraphedStep announces its capture through
    begin_capture(): without one no stream query is made -- six of them cost 25 us of a Python-driven step.)"""
    if not _capture['active']:
        return 0
    return _capture['active'] if _capturing() else 0


def _tag_visible(tag):
    return tag == 0 or tag == _cap_tag()


def begin_capture():
    """Called by graph.GraphedStep right before it starts recording: new capture id, fresh log of raw writes.
    One capture at a time per process: the bookkeeping is process-wide (hipGraph capture itself is per stream, but the caches
    tagged here are shared), so a second capture started before the first has ended is refused instead of mixing their logs."""
    if _capture['active']:
        raise L.EnslamError("a hipGraph capture of this library is already being recorded in this process (captures do not nest "
                            "and cannot run on two threads at once)")
    _capture['epoch'] += 1
    _capture['active'] = _capture['epoch']
    _capture['raw_writes'] = []
    _capture['init_zero'] = []
    _capture['persist'] = {}          # id(grid) -> (gradient data_ptr, prev flags) of this capture
    _capture['persist_need'] = {}     # id(grid) -> the flags this step's sampler marks (persistent native gradients)


def end_capture(ok=True):
    """-> the tensors that kernels of the captured step write behind torch's back (note_raw_write).  ok=False: the capture
    raised -- its storages belong to a pool that is being torn down, nothing is touched (and nothing here may hide the
    original exception)."""
    log, _capture['raw_writes'] = _capture['raw_writes'] or [], None
    _capture['active'] = 0
    # persistent gradient tensors of the captured step and their block flags (see _RenderFn.backward): the protocol starts
    # from all-zero memory; filled here, eagerly, once -- a fill inside the capture would run at every replay
    init, _capture['init_zero'] = _capture['init_zero'] or [], None
    if not ok:
        forget_persistent(list((_capture['persist'] or {}).values()))
        _capture['persist'] = None
        _capture['persist_need'] = {}
        end_capture.persist_keys = []
        return []
    for storage in init:
        torch.empty(0, dtype=torch.uint8, device=storage.device).set_(storage).zero_()
    seen, out = set(), []
    for t in log:
        if id(t) not in seen:
            seen.add(id(t))
            out.append(t)
    persist, _capture['persist'] = _capture['persist'] or {}, None
    _capture['persist_need'] = {}
    end_capture.persist_keys = list(persist.values())                      # (ptr, flags) pairs, read by graph.GraphedStep
    return out


def forget_persistent(keys):
    """The graph that owned these persistent gradient buffers is gone (graph.GraphedStep.close): keys = its (data_ptr, flags)
    pairs.  Only entries that still carry THESE flags are removed -- a newer graph may have been given the same address."""
    for ptr, pv in keys:
        e = _persist_prev.get(ptr)
        if e is not None and e[1] is pv:
            del _persist_prev[ptr]


def note_foreign_blocks(grad, flags):
    """`grad` (a dense feature-grid gradient) has just been written in the 64-voxel blocks flagged in `flags` (uint8, one per
    block) by something other than the render backward -- the unpack of a gradient all-reduce.  If it is the persistent
    gradient buffer of a captured step, those blocks join the flags of the blocks its next replay must rewrite."""
    e = _persist_prev.get(grad.data_ptr())
    if e is None:
        return False
    st0, pv = e
    if st0._cdata != grad.untyped_storage()._cdata:     # another buffer at the same address
        return False
    if pv.numel() == flags.numel():
        torch.maximum(pv, flags.reshape(pv.shape).to(pv.dtype), out=pv)
        return True
    return False


def note_raw_write(tensors):
    """A library kernel is about to overwrite these tensors in place (mapper.FusedAdam, MaskedGridOptimizer): bump their
    version counters so the caches keyed on `_version` miss, and -- under capture -- remember them so that every replay
    of the graph bumps them again (the Python call itself only runs at capture time)."""
    torch._C._increment_version(list(tensors))      # (a LIST: handed a tensor, the call iterates over its rows -- 25 us each)
    if _capture['raw_writes'] is not None and _capturing():
        _capture['raw_writes'].extend(tensors)


_bound6_cache = {}


def bound6(bound):
    """[3,2] tensor -> (c_double*6) x_lo,x_hi,y_lo,y_hi,z_lo,z_hi.  Cached per (tensor identity, version): a scene's bound is
    set once (EvenNICER_SLAM.load_bound) and read by every render call."""
    key = id(bound)
    hit = _bound6_cache.get(key)
    if hit is not None and hit[0]() is bound and hit[1] == bound._version:
        return hit[2]
    b = bound.detach().to('cpu', torch.float64).reshape(-1).tolist()
    out = (ctypes.c_double * 6)(*b)
    if len(_bound6_cache) > 64:
        _bound6_cache.clear()
    _bound6_cache[key] = (weakref.ref(bound), bound._version, out)
    return out


# ------------------------------------------------------------------------------------------------
# decoder parameters: ordering, packing cache
# ------------------------------------------------------------------------------------------------
_layer_cache = weakref.WeakKeyDictionary()     # decoder module -> its Linear sub-modules (+ embedder), ABI order


def decoder_params(dec, kind):
    """Parameter tensors of one decoder in ABI order.  Works on this package's modules and on any
    module with the reference's attribute names (pts_linears, fc_c, output_linear, embedder._B).
    The sub-module list is cached (nn.ModuleList indexing is slow); parameters are re-read every call."""
    mods = _layer_cache.get(dec)
    if mods is None:
        mods = [dec.pts_linears[i] for i in range(5)]
        if kind != L.MLP_COARSE:
            mods += [dec.fc_c[i] for i in range(5)]
        mods.append(dec.output_linear)
        _layer_cache[dec] = mods
    ps = []
    for m in mods:
        p = m._parameters
        ps.append(p['weight'])
        ps.append(p['bias'])
    if kind != L.MLP_COARSE:
        ps.append(dec.embedder._B)
    return ps


_flat_offsets = {}


def _flat_params_struct(kind, like, base):
    """enslam_mlp_params of tensors laid out back to back (float32) from address `base` in decoder_params() order with the
    sizes of `like` -> (struct, address behind the last one).  The byte offsets are cached per (kind, sizes)."""
    key = (kind, tuple(t.numel() for t in like))
    offs = _flat_offsets.get(key)
    if offs is None:
        o, offs = 0, []
        for n in key[1]:
            offs.append(o)
            o += 4 * n
        offs.append(o)
        _flat_offsets[key] = offs = tuple(offs)
    s = L.MlpParams()
    it = iter(offs)
    for i in range(5):
        s.W[i] = base + next(it)
        s.b[i] = base + next(it)
    if kind != L.MLP_COARSE:
        for i in range(5):
            s.Wc[i] = base + next(it)
            s.bc[i] = base + next(it)
    s.Wo = base + next(it)
    s.bo = base + next(it)
    if kind != L.MLP_COARSE:
        s.B = base + next(it)
    return s, base + offs[-1]


def _fill_params_struct(kind, tensors):
    """enslam_mlp_params from tensors in decoder_params() order (entries may be None)."""
    s = L.MlpParams()
    it = iter(tensors)
    for i in range(5):
        s.W[i] = _ptrv(next(it))
        s.b[i] = _ptrv(next(it))
    if kind != L.MLP_COARSE:
        for i in range(5):
            s.Wc[i] = _ptrv(next(it))
            s.bc[i] = _ptrv(next(it))
    s.Wo = _ptrv(next(it))
    s.bo = _ptrv(next(it))
    if kind != L.MLP_COARSE:
        s.B = _ptrv(next(it))
    return s


_EXPECT_SHAPES = {
    L.MLP_COARSE: [(32, 32), (32,), (32, 32), (32,), (32, 32), (32,), (32, 64), (32,), (32, 32), (32,), (1, 32), (1,)],
}


def _xyz_shapes(cd, nout):
    s = []
    for k in (93, 32, 32, 125, 32):
        s += [(32, k), (32,)]
    for _ in range(5):
        s += [(32, cd), (32,)]
    return s + [(nout, 32), (nout,), (3, 93)]


_EXPECT_SHAPES[L.MLP_MIDDLE] = _xyz_shapes(32, 1)
_EXPECT_SHAPES[L.MLP_FINE] = _xyz_shapes(64, 1)
_EXPECT_SHAPES[L.MLP_COLOR] = _xyz_shapes(32, 4)


def _check_params(kind, ps):
    for t, shp in zip(ps, _EXPECT_SHAPES[kind]):
        if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous():
            raise L.EnslamError(f"decoder {L.MLP_NAMES[kind]}: parameter of shape {tuple(t.shape)} / {t.dtype}, "
                                f"expected contiguous float32 {shp}")
        _require_hip(t, "decoder parameters")


_ELEM_SIZE = {torch.uint8: 1, torch.int32: 4, torch.float32: 4, torch.float64: 8, torch.int64: 8}
_size_memo = {}


def _lib_size(name, *args):
    """Memoised size queries of the library (enslam_packed_floats & co.: pure functions of small integers)."""
    key = (name,) + args
    v = _size_memo.get(key)
    if v is None:
        v = _size_memo[key] = int(getattr(L.lib(), name)(*args))
    return v


class _ZeroArena:
    """One zero-filled allocation per render call, handed out in aligned typed slices (block flags, validity
    bitmaps, packed-decoder buffers): one fill node in the step's graph instead of one per consumer."""

    def __init__(self, device, nbytes, persistent=False):
        # persistent (inside a captured training step, see _RenderFn.forward): no fill node -- the memory is the same at
        # every replay, zeroed once by end_capture(), and every slice handed out is back to zero (or wholly rewritten) when
        # the step ends.  Allocated at the first take(): a call that needs nothing zeroed (tracker iterations on a cached
        # map) gets no fill node either.
        self.device, self.nbytes, self.persistent = device, max(int(nbytes), 16), persistent
        self.buf = None
        self.off = 0

    def _alloc(self):
        if self.persistent:
            self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=self.device)
            _capture['init_zero'].append(self.buf.untyped_storage())
        else:
            self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)

    def take(self, n, dtype):
        if self.buf is None:
            self._alloc()
        nb = n * _ELEM_SIZE[dtype]
        off = (self.off + 15) & ~15
        if off + nb > self.buf.numel():
            return torch.zeros(n, dtype=dtype, device=self.buf.device)
        self.off = off + nb
        return self.buf[off:off + nb].view(dtype)


class _PackCache:
    """Packed form of a decoder, refreshed when any parameter's (data_ptr, _version) changes."""

    def __init__(self):
        self.key = None
        self.packed = None
        self.tag = 0            # capture the packing was made in (0: eager memory)

    def fresh(self, key):
        return key == self.key and self.packed is not None and _tag_visible(self.tag)

    def get(self, kind, ps):
        key = _params_key(self, ps)
        if not self.fresh(key):
            _check_params(kind, ps)
            n = L.lib().enslam_packed_floats(kind)
            # fresh buffer: an earlier forward's saved packing stays valid for its backward
            packed = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
            s = _fill_params_struct(kind, ps)
            L.check(L.lib().enslam_pack_mlp(kind, ctypes.byref(s), _ptr(packed), _stream()), "enslam_pack_mlp")
            self.key, self.packed, self.tag = key, packed, _cap_tag()       # (only after the launch was accepted)
        return self.packed


_pack_caches = weakref.WeakKeyDictionary()      # decoder module -> _PackCache


def _params_key(cache, ps):
    """Cache key of a decoder's parameter list: (id, data_ptr, _version) per tensor -- 69 x 3 attribute reads per colour-stage
    call when built naively.  While the list holds the SAME tensor objects as the previous call (the normal case: a module's
    parameters) only the version counters and the storage addresses are re-read, and the address part of the previous key is
    reused when none moved (an optimiser step changes versions, never storage)."""
    prev = getattr(cache, 'ps_seen', None)
    if prev is not None and len(prev) == len(ps) and all(a is b for a, b in zip(prev, ps)):
        ptrs = tuple(p.data_ptr() for p in ps)
        if ptrs == cache.ptrs_seen:
            return (cache.ids_seen, ptrs, tuple(_ver(p) for p in ps))
    cache.ps_seen = list(ps)
    cache.ids_seen = tuple(id(p) for p in ps)
    cache.ptrs_seen = tuple(p.data_ptr() for p in ps)
    return (cache.ids_seen, cache.ptrs_seen, tuple(_ver(p) for p in ps))


def packed_decoders(items, arena=None, defer=False):
    """Packed forms of several decoders [(module, kind, params)]; stale ones are rebuilt with ONE zero-fill and ONE
    launch (up to three decoders per launch).  defer=True (at most three stale decoders): nothing is launched, the
    second return value holds the arguments (n, kinds, structs, ptrs) for enslam_step_prepare, or None, and the third
    a `commit()` the caller invokes once that launch has been accepted -- a cache entry is only marked fresh then, so a
    failure in between (allocation, a refused launch) cannot leave a "fresh" entry pointing at a zero-filled buffer."""
    out, stale = [], []
    for i, (dec, kind, ps) in enumerate(items):
        cache = _pack_caches.get(dec)
        if cache is None:
            cache = _pack_caches[dec] = _PackCache()
        key = _params_key(cache, ps)
        if cache.fresh(key):
            out.append(cache.packed)
        else:
            out.append(None)
            stale.append((i, cache, key))
    pending = []

    def commit():
        tag = _cap_tag()
        for cache, key, packed in pending:
            cache.key, cache.packed, cache.tag = key, packed, tag
        del pending[:]

    if stale:
        lib = L.lib()
        sizes = [_lib_size('enslam_packed_floats', items[i][1]) for i, _, _ in stale]
        flat = (arena.take(sum(sizes), torch.float32) if arena is not None else
                torch.zeros(sum(sizes), dtype=torch.float32, device=items[stale[0][0]][2][0].device))
        pieces = flat.split(sizes)
        for j, (i, cache, key) in enumerate(stale):
            # an optimiser step changes versions, not storage: the shape checks and the pointer struct of unchanged
            # storages are reused (they are half of the host cost of a re-pack)
            where = key[:2]
            if where != getattr(cache, 'where', None):
                _check_params(items[i][1], items[i][2])
                cache.where, cache.struct = where, _fill_params_struct(items[i][1], items[i][2])
            pending.append((cache, key, pieces[j]))
            out[i] = pieces[j]
        deferred = None
        for g0 in range(0, len(stale), 3):
            grp = stale[g0:g0 + 3]
            n = len(grp)
            kinds, structs, ptrs = (ctypes.c_int32 * n)(), (L.MlpParams * n)(), (ctypes.c_void_p * n)()
            for j, (i, cache, key) in enumerate(grp):
                kinds[j] = items[i][1]
                structs[j] = cache.struct
                ptrs[j] = out[i].data_ptr()
            if defer and len(stale) <= 3:
                deferred = (n, kinds, structs, ptrs)
            else:
                L.check(lib.enslam_pack_mlp_multi(n, kinds, structs, ptrs, _stream()), "enslam_pack_mlp_multi")
        if defer:
            if deferred is None:
                commit()
            return out, deferred, commit
        commit()
    elif defer:
        return out, None, commit
    return out


def packed_decoder(dec, kind, params=None):
    cache = _pack_caches.get(dec)
    if cache is None:
        cache = _pack_caches[dec] = _PackCache()
    return cache.get(kind, params if params is not None else decoder_params(dec, kind))


# ------------------------------------------------------------------------------------------------
# grids: voxel-major copies, cached per (storage, version)
# ------------------------------------------------------------------------------------------------
class _GridEntry:
    __slots__ = ('ref', 'version', 'vm', 'valid', 'tag')

    def __init__(self, ref, version, vm, valid, tag):
        self.ref, self.version, self.vm, self.valid, self.tag = ref, version, vm, valid, tag


def _check_grid(g):
    if g.dim() != 5 or g.shape[0] != 1 or g.shape[1] != 32 or g.dtype != torch.float32:
        raise L.EnslamError(f"feature grid must be float32 [1,32,D,H,W], got {tuple(g.shape)} {g.dtype}")
    _require_hip(g, "feature grids")


_CL3D = torch.channels_last_3d


def is_native_grid(g):
    """A feature grid whose storage already is the kernels' own [V][32] layout: a [1,32,D,H,W] tensor in torch's
    channels_last_3d memory format (`grid.contiguous(memory_format=torch.channels_last_3d)` -- same shape, same values, same
    indexing as the reference's grids; only the strides differ).  Such a grid is gathered from where it lies, nothing is
    converted per step, and its gradient comes back in the same memory format without a transposed copy."""
    return (g.dim() == 5 and g.shape[1] == 32 and g.is_contiguous(memory_format=_CL3D) and not g.is_contiguous()
            and g.data_ptr() % 16 == 0)           # (the gathers load 16 bytes at a time: an odd view offset takes the copying route)


def _native_vm(g):
    """[V,32] view of a native grid's storage."""
    return g.detach().permute(0, 2, 3, 4, 1).reshape(-1, 32)


def _native_grad_view(flat, dims):
    """[1,32,D,H,W] channels_last_3d tensor over a flat [V*32] gradient buffer."""
    D, H, W = dims
    return flat.view(1, D, H, W, 32).permute(0, 4, 1, 2, 3)


class _GridCache:
    """Voxel-major copies keyed on tensor IDENTITY + version counter.  (A data_ptr key would go stale when the
    caching allocator hands a freed grid's address to a new tensor, e.g. Tracker.update_para_from_mapping's
    per-frame clones.)  Entries die with their source tensor.  An entry is either dense (every voxel converted,
    `valid is None`) or sparse (`valid` = bitmap of the 64-voxel blocks converted so far); entries made while a
    hipGraph capture records are only visible inside that capture (see _capture)."""

    def __init__(self):
        self.items = {}          # id(tensor) -> _GridEntry

    def _lookup(self, g, sparse):
        e = self.items.get(id(g))
        if e is not None and e.ref() is g and e.version == _ver(g) and (e.valid is not None) == sparse and _tag_visible(e.tag):
            return e
        return None

    def _store(self, g, vm, valid):
        key, items = id(g), self.items
        items[key] = _GridEntry(weakref.ref(g, lambda _r, key=key, items=items: items.pop(key, None)), g._version, vm, valid,
                                _cap_tag())

    def get(self, g):
        _check_grid(g)
        if is_native_grid(g):
            return _native_vm(g)
        e = self._lookup(g, False)
        if e is not None:
            return e.vm
        src = g.detach()
        if not src.is_contiguous():
            src = src.contiguous()
        V = g.shape[2] * g.shape[3] * g.shape[4]
        # fresh buffer each refresh: an earlier forward's saved copy stays valid for its backward
        vm = torch.empty((V, 32), dtype=torch.float32, device=g.device)
        L.check(L.lib().enslam_grid_to_voxel_major(_ptr(src), _ptr(vm), V, _stream()), "grid_to_voxel_major")
        self._store(g, vm, None)
        return vm

    def get_many(self, grids):
        """Voxel-major copies of several grids; all cache misses are converted in ONE launch."""
        out, miss = [], []
        for i, g in enumerate(grids):
            if is_native_grid(g):
                _check_grid(g)
                out.append(_native_vm(g))
                continue
            e = self._lookup(g, False)
            out.append(e.vm if e is not None else None)
            if e is None:
                miss.append(i)
        if len(miss) == 1:
            out[miss[0]] = self.get(grids[miss[0]])
        elif miss:
            n = len(miss)
            srcs, dsts, vs, keep = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_int64 * n)(), []
            for j, i in enumerate(miss):
                g = grids[i]
                _check_grid(g)
                src = g.detach()
                src = src if src.is_contiguous() else src.contiguous()
                V = g.shape[2] * g.shape[3] * g.shape[4]
                vm = torch.empty((V, 32), dtype=torch.float32, device=g.device)
                srcs[j], dsts[j], vs[j] = src.data_ptr(), vm.data_ptr(), V
                keep.append(src)
                out[i] = vm
            L.check(L.lib().enslam_grids_convert(n, srcs, dsts, vs, 1, _stream()), "enslam_grids_convert")
            for i in miss:
                self._store(grids[i], out[i], None)
        return out

    def get_many_sparse(self, grids, need, arena=None, defer=False, fresh=False):
        """Voxel-major copies in which (at least) the 64-voxel blocks flagged in need[i] (uint8 tensors) are valid.
        Every entry carries a `valid` bitmap; one launch converts the blocks that are needed and not yet valid.
        fresh: no bitmap and no cache entry -- every flagged block is converted by this call's launch (a captured
        training step: the grids change between replays, and a bitmap would have to be cleared at every replay)."""
        n = len(grids)
        srcs, dsts, vs = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_int64 * n)()
        needs, valids, out, keep = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), [], []
        for i, g in enumerate(grids):
            e = None if fresh else self._lookup(g, True)
            if fresh:
                _check_grid(g)
                V = g.shape[2] * g.shape[3] * g.shape[4]
                vm, valid = torch.empty((V, 32), dtype=torch.float32, device=g.device), None
            elif e is not None:
                vm, valid = e.vm, e.valid
            else:
                _check_grid(g)
                V = g.shape[2] * g.shape[3] * g.shape[4]
                vm = torch.empty((V, 32), dtype=torch.float32, device=g.device)
                valid = (arena.take((V + 63) // 64, torch.uint8) if arena is not None else
                         torch.zeros((V + 63) // 64, dtype=torch.uint8, device=g.device))
                self._store(g, vm, valid)
            src = g.detach()
            src = src if src.is_contiguous() else src.contiguous()
            keep.append(src)
            srcs[i], dsts[i], vs[i] = src.data_ptr(), vm.data_ptr(), vm.shape[0]
            needs[i], valids[i] = need[i].data_ptr(), (valid.data_ptr() if valid is not None else None)
            out.append(vm)
        if defer:                    # the caller launches (enslam_step_prepare); `keep` pins the sources until then
            return out, (n, srcs, dsts, vs, needs, valids, keep)
        L.check(L.lib().enslam_grids_convert_sparse(n, srcs, dsts, vs, needs, valids, 1, _stream()),
                "enslam_grids_convert_sparse")
        return out


_grid_cache = _GridCache()


def clear_caches():
    _grid_cache.items.clear()
    _pack_caches.clear()


def refresh_in_place(c, decoders, stage='color'):
    """Bring the cached device-side forms of a map up to date IN PLACE after its tensors were updated in place
    (`Tracker.update_para_from_mapping` copying the mapper's state into the tracker's tensors, an optimiser step):
    the voxel-major copies of dense no-gradient grids and the packed decoders are rewritten inside their existing
    buffers.  A captured step (graph.GraphedStep, tracker.GraphedCameraIteration) that was recorded while those caches
    were warm contains no conversion / packing launch and reads exactly these buffers, so this call is what makes it
    see the new map.  Must not run between a forward and its backward.  Returns the number of buffers rewritten."""
    lib = L.lib()
    n = 0
    for k in stage_kinds(stage):
        g = c[L.GRID_NAMES[k]]
        if isinstance(g, VoxelMajorGrid):
            continue                                    # already the kernels' layout: nothing cached
        e = _grid_cache.items.get(id(g))
        if e is not None and e.ref() is g and e.valid is None and e.tag == 0 and e.version != g._version:
            src = g.detach()
            src = src if src.is_contiguous() else src.contiguous()
            L.check(lib.enslam_grid_to_voxel_major(_ptr(src), _ptr(e.vm), e.vm.shape[0], _stream()), "grid_to_voxel_major")
            e.version = g._version
            n += 1
        dec = getattr(decoders, L.MLP_NAMES[k])
        cache = _pack_caches.get(dec)
        if cache is None or cache.packed is None or cache.tag != 0:
            continue
        ps = decoder_params(dec, k)
        key = _params_key(cache, ps)
        if key != cache.key:
            where = key[:2]
            if where != getattr(cache, 'where', None) or getattr(cache, 'struct', None) is None:
                _check_params(k, ps)
                cache.where, cache.struct = where, _fill_params_struct(k, ps)
            L.check(lib.enslam_pack_mlp(k, ctypes.byref(cache.struct), _ptr(cache.packed), _stream()), "enslam_pack_mlp")
            cache.key = key
            n += 1
    return n


def _scene_struct(stage, bound, coarse_bound, grids_vm, grid_dims, packed):
    """enslam_scene for a stage. grids_vm / packed: dict kind -> tensor."""
    sc = L.Scene()
    sc.bound = bound
    sc.coarse_bound = coarse_bound
    for k in range(4):
        if k in grids_vm:
            sc.grids[k].data = grids_vm[k].data_ptr()
            sc.grids[k].D, sc.grids[k].H, sc.grids[k].W = grid_dims[k]
        if k in packed:
            sc.packed[k] = packed[k].data_ptr()
    return sc


def stage_kinds(stage):
    return L.STAGE_KINDS[stage]


# ------------------------------------------------------------------------------------------------
# the differentiable render call
# ------------------------------------------------------------------------------------------------
USE_WORK_LIST = os.environ.get('ENSLAM_WORK_LIST', '1') == '1'     # process-wide default of RenderState.use_work_list


class RenderState:
    """What a render call leaves behind for its caller, per Renderer (two renderers -- a tracker's and a mapper's in one
    process, possibly on different streams -- do not see each other's):
      flags    {id(grid tensor): uint8 flags} of the 64-voxel blocks the latest call touched (parallel.allreduce_gradients
               sends only those); under hipGraph replay the buffers are rewritten in place by every replay
      work     (counter tensor, number of tiles) of the latest call's backward work list
      profile  {'decoder_bwd': [(event, event), ...]} when a caller (bench.py) asks for per-kernel timing
      use_work_list  whether this renderer's backwards walk the work list of active tiles (default) or every tile"""

    def __init__(self):
        self.flags = {}
        self.work = (None, 0)
        self.profile = {}
        # the backward walks only the 16-sample tiles whose d_raw is not all zero (per renderer; ENSLAM_WORK_LIST=0 sets the
        # process-wide default to every tile -- a diagnostic: the results are the same)
        self.use_work_list = USE_WORK_LIST

    def last_block_flags(self):
        return dict(self.flags)

    def last_active_tile_fraction(self):
        """Share of the 16-sample tiles of the most recent render call that its backward scheduled (work list of tiles
        with non-zero d_raw); None when that call kept no list.  Synchronises."""
        wcount, n = self.work
        if wcount is None or n == 0:
            return None
        return float(wcount.item()) / n


_default_state = RenderState()      # calls made without a Renderer (functional.render with a bare plan)
_latest_state = [_default_state]


def last_block_flags():
    """`RenderState.last_block_flags()` of whichever renderer rendered last in this process (single-renderer callers)."""
    return _latest_state[0].last_block_flags()


class VoxelMajorGrid:
    """A feature grid kept in the kernels' own layout across iterations (see mapper.MaskedGridOptimizer):
    `vm` float32 [V,32] values, `grad_vm` float32 [V,32] gradient accumulator that render backwards ADD into and
    the optimiser consumes and clears.  Passing one of these in the `c` dict of Renderer.render_batch_ray skips the
    per-call layout conversion, block marking, accumulator clearing and transposed-back gradient.  `anchor` is the
    tensor that ties the object into autograd (its own gradient is always None)."""

    def __init__(self, dims, vm, grad_vm):
        self.dims = tuple(int(d) for d in dims)
        V = self.dims[0] * self.dims[1] * self.dims[2]
        for t, name in ((vm, "vm"), (grad_vm, "grad_vm")):
            _require_hip(t, name)
            if tuple(t.shape) != (V, 32) or t.dtype != torch.float32 or not t.is_contiguous():
                raise L.EnslamError(f"VoxelMajorGrid.{name}: expected contiguous float32 [{V},32], got {tuple(t.shape)} {t.dtype}")
        self.vm, self.grad_vm = vm, grad_vm
        self.anchor = torch.zeros(1, dtype=torch.float32, device=vm.device, requires_grad=True)
        self.has_grad = False           # set by a backward that added into grad_vm

    @property
    def shape(self):
        return (1, 32) + self.dims

    @property
    def device(self):
        return self.vm.device

    @property
    def requires_grad(self):
        return self.anchor.requires_grad


class RenderPlan:
    """Static description of one render_batch_ray call (everything that is not a differentiable tensor)."""

    def __init__(self, stage, bound, coarse_bound, n_lin, n_surf, lindisp, t_lin, t_surf, kinds, decoders,
                 depth_max=None):
        self.stage = stage
        self.depth_max = depth_max              # float32 [2] {max, fl32(max*1.2)} of the WHOLE batch, or None
        self.bound6 = bound6(bound)
        self.coarse_bound6 = bound6(coarse_bound)
        self.n_lin, self.n_surf, self.lindisp = n_lin, n_surf, int(bool(lindisp))
        self.t_lin, self.t_surf = t_lin, t_surf
        self.kinds = kinds                      # decoder / grid kinds used, ascending
        self.decoders = decoders                # dict kind -> module
        self.n_params = {k: (12 if k == L.MLP_COARSE else 23) for k in kinds}
        self.vm = {}                            # kind -> VoxelMajorGrid for grids passed in the device layout
        self.loss = None                        # (gt_depth [N] f32, gt_color [N,3] f32 | None, w_color): fused mapper loss
        self.state = _default_state             # the calling Renderer's RenderState
        # hierarchical sampling's second pass (Renderer.py:182-197): sample distances given by the caller instead of the
        # sampler -- float64 [N, S] with S a multiple of 16 (<= 64), of which the first s_valid per ray are real samples
        # (the rest pad the last tile: evaluated, but left out of the compositing and given zero gradient)
        self.z_given = None
        self.s_valid = None


class _Accumulators:
    """The buffers a render backward adds into: voxel-major grid gradients (only the touched blocks are ever cleared,
    read or transposed back) and one flat buffer of packed-layout decoder gradients and ray gradients.  Laid out in
    the forward, because the launch that prepares the forward's inputs clears them as well."""

    def __init__(self, plan, dims, needs, N, dev, lib, native=(), grid_ids=None):
        nk = len(plan.kinds)
        self.need_rays = bool(needs[1] or needs[2])
        self.need_grid = {k: bool(needs[5 + i]) for i, k in enumerate(plan.kinds)}
        off, self.need_par = 5 + nk, {}
        for k in plan.kinds:
            n = plan.n_params[k]
            self.need_par[k] = any(needs[off:off + n])
            off += n
        # Gradients of grids that arrive in the kernels' own layout (channels_last_3d tensors, `native`): the buffer the backward
        # adds into IS the tensor autograd receives.  Eagerly it is a slice of the flat buffer below (fresh and cleared as a whole
        # per call); under hipGraph capture it is the same memory at every replay and only the blocks the previous replay
        # touched are cleared (nat_persist: kind -> (gradient buffer [V*32], prev flags); the flags move there in the finish
        # launch).  A second render of the same grid inside one captured step is not persistent (see _RenderFn.backward).
        self.native = {k: (k in native and self.need_grid[k]) for k in plan.kinds}
        self.nat_persist, self.nat_again = {}, {}
        persist_ok = _capturing() and _capture['init_zero'] is not None
        sizes = []
        for k in plan.kinds:
            D, H, W = dims[k]
            sizes.append(D * H * W * 32 if (self.need_grid[k] and k not in plan.vm and not self.native[k]) else 0)
        for k in plan.kinds:
            sizes.append(_lib_size('enslam_packed_grad_floats', k) if self.need_par[k] else 0)
        sizes.append(6 * N if self.need_rays else 0)
        pers = []
        for i, k in enumerate(plan.kinds):
            D, H, W = dims[k]
            n = D * H * W * 32 if self.native[k] else 0
            if n and persist_ok:
                if grid_ids[i] in _capture['persist']:
                    self.nat_again[k] = grid_ids[i]
                else:
                    pers.append((i, k, n))
                    n = 0
            sizes.append(n)
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        self.sizes, self.offs = sizes, offs
        self.n_grid = offs[nk]
        self.n_flat = offs[-1] - self.n_grid
        self.gbuf = torch.empty(max(self.n_grid, 1), dtype=torch.float32, device=dev)
        # flat part + 16 bytes the same launch clears: the work-list counter (int32) and the fused loss (float64) of a
        # captured training step live here instead of in a zero-filled arena
        self.tail = (self.n_flat + 1) & ~1
        self.n_zero = self.tail + 4
        self.zbuf = torch.empty(self.n_zero, dtype=torch.float32, device=dev)
        self.clean = False          # set by the launch that cleared them; a backward consumes it
        self.pv_flat = None
        if pers:
            nblk = [(n // 32 + 63) // 64 for _i, _k, n in pers]
            self.pv_flat = torch.empty(sum(nblk), dtype=torch.uint8, device=dev)
            _capture['init_zero'].append(self.pv_flat.untyped_storage())
            for (i, k, n), pv in zip(pers, self.pv_flat.split(nblk)):
                g = torch.empty(n, dtype=torch.float32, device=dev)
                _capture['init_zero'].append(g.untyped_storage())
                _capture['persist'][grid_ids[i]] = (g.data_ptr(), pv)
                _persist_register(g, pv)
                self.nat_persist[k] = (g, pv)

    def native_grad(self, plan, k):
        """flat [V*32] gradient buffer of native grid kind k"""
        if k in self.nat_persist:
            return self.nat_persist[k][0]
        i = plan.kinds.index(k)
        j = 2 * len(plan.kinds) + 1 + i
        o = self.offs[j] - self.n_grid
        return self.zbuf[o:o + self.sizes[j]]

    def counter(self):
        return self.zbuf[self.tail:self.tail + 1].view(torch.int32)

    def loss_slot(self):
        return self.zbuf[self.tail + 2:self.tail + 4].view(torch.float64)

    def zero_args(self, plan, flags):
        """(n, dsts, n_voxels, need flags) of the grid accumulators for enslam_zero_blocks / enslam_step_prepare."""
        zl = [(i, k) for i, k in enumerate(plan.kinds) if self.need_grid[k] and k not in plan.vm and not self.native[k]]
        n = len(zl) + len(self.nat_persist)
        dsts, vs, nd = (ctypes.c_void_p * max(n, 1))(), (ctypes.c_int64 * max(n, 1))(), (ctypes.c_void_p * max(n, 1))()
        for j, (i, k) in enumerate(zl):
            dsts[j], vs[j], nd[j] = self.gbuf.data_ptr() + 4 * self.offs[i], self.sizes[i] // 32, flags[i].data_ptr()
        for j, (g, pv) in enumerate(self.nat_persist.values()):      # persistent native gradients: the blocks touched one replay ago
            dsts[len(zl) + j], vs[len(zl) + j], nd[len(zl) + j] = g.data_ptr(), g.numel() // 32, pv.data_ptr()
        return n, dsts, vs, nd


class _RenderFn(torch.autograd.Function):
    """inputs: plan, rays_o, rays_d, gt_depth|None, t_rand|None, then for each kind in plan.kinds: grid,
    then for each kind: its parameters.  Outputs depth f64 [N], var f64 [N], rgb f32 [N,3]; with plan.loss the
    fused mapper loss (f64 scalar) comes first and is the only differentiable output."""

    @staticmethod
    def forward(ctx, plan, rays_o, rays_d, gt_depth, t_rand, *tensors):
        lib = L.lib()
        ctx.set_materialize_grads(False)          # unused outputs (the variance) arrive as None, not as a zero fill
        nk = len(plan.kinds)
        grids = tensors[:nk]
        N = rays_o.shape[0]
        dev = rays_o.device
        S = plan.n_lin + (plan.n_surf if gt_depth is not None else 0)
        st = _stream()
        ro, rd = _f32c(rays_o), _f32c(rays_d)
        gd = _f32c(gt_depth).reshape(-1) if gt_depth is not None else None
        SV = None                                   # real samples per ray when the last tile is padded
        if plan.z_given is not None:
            z = plan.z_given.detach().to(torch.float64).contiguous()
            S = int(z.shape[1])
            if tuple(z.shape) != (N, S) or S % 16 != 0 or S > 64 or plan.loss is not None:
                raise L.EnslamError(f"given sample distances must be float64 [N, 16k <= 64] without a fused loss, got {tuple(z.shape)}")
            if plan.s_valid is not None and plan.s_valid < S:
                SV = int(plan.s_valid)
        else:
            z = torch.empty((N, S), dtype=torch.float64, device=dev)
        scratch = plan.depth_max if plan.depth_max is not None else torch.empty(2, dtype=torch.float32, device=dev)
        # blocks of 64 voxels this batch touches, per grid (one zeroed byte buffer for all grids).  Grids that
        # arrive in the device layout (plan.vm) need none of this.
        vmg = plan.vm
        dims = {k: (vmg[k].dims if k in vmg else tuple(g.shape[2:])) for k, g in zip(plan.kinds, grids)}
        # grids without gradient (tracker, render_img, Mesher): one full conversion per grid version, cached -- the map
        # does not change between the camera iterations of a frame, so nothing is marked or converted per call
        # grids in the kernels' own layout (channels_last_3d tensors): read where they lie, never converted
        native = {k for i, k in enumerate(plan.kinds) if k not in vmg and is_native_grid(grids[i])}
        static = [(i, k) for i, k in enumerate(plan.kinds) if k not in vmg and k not in native and not ctx.needs_input_grad[5 + i]]
        dense = [(i, k) for i, k in enumerate(plan.kinds) if k not in vmg and ctx.needs_input_grad[5 + i]]
        nblk = [(dims[k][0] * dims[k][1] * dims[k][2] + 63) // 64 for _, k in dense]
        # A captured training step (fused render + loss: its backward always follows) needs no zero-fill node at all: the
        # block flags are cleared by the finish launch that consumes them (enslam_step_finish_rays_prev), the packed
        # decoders' padding stays zero, the conversion runs without a validity bitmap and the two accumulating scalars sit
        # in the flat buffer the prepare launch clears.  Everywhere else: one zero-filled arena per call.
        cap = bool(_capturing() and _capture['init_zero'] is not None and plan.loss is not None and plan.z_given is None
                   and any(ctx.needs_input_grad[5:5 + nk]))
        arena = _ZeroArena(dev, 2 * sum(nblk) + 4 * sum(_lib_size('enslam_packed_floats', k) for k in plan.kinds) + 256 + 32, persistent=cap)
        flags = [None] * nk
        grids_vm, packed = {k: vmg[k].vm for k in vmg}, {}
        for i, k in enumerate(plan.kinds):
            if k in native:
                _check_grid(grids[i])
                grids_vm[k] = grids[i].detach()
        state = plan.state
        _latest_state[0] = state
        state.flags = {}
        accum = None
        if any(ctx.needs_input_grad):
            accum = _Accumulators(plan, dims, ctx.needs_input_grad, N, dev, lib, native, [id(g) for g in grids])
        msc, fptr = None, None
        flag_move = None                # (flags, prev flags) of the persistent native gradients: one contiguous range each
        if dense:                       # the sampler marks the blocks of the samples it places
            # the flags of persistent native gradients sit together, in the order of their `prev` flags (the finish launch moves
            # them there as one range)
            order = [j for j, (_i, k) in enumerate(dense) if k in accum.nat_persist] + \
                    [j for j, (_i, k) in enumerate(dense) if k not in accum.nat_persist]
            flag_buf = arena.take(sum(nblk), torch.uint8)
            fptr = (ctypes.c_void_p * 4)()
            msc = L.Scene()
            msc.bound, msc.coarse_bound = plan.bound6, plan.coarse_bound6
            for j, fl in zip(order, flag_buf.split([nblk[j] for j in order])):
                i, k = dense[j]
                flags[i] = fl
                fptr[k] = fl.data_ptr()
                msc.grids[k].D, msc.grids[k].H, msc.grids[k].W = dims[k]
                state.flags[id(grids[i])] = fl
            if accum.pv_flat is not None:
                flag_move = (flag_buf[:accum.pv_flat.numel()], accum.pv_flat)
        conv = [(i, k) for i, k in dense if k not in native]
        merged = plan.z_given is None and not conv        # nothing to convert: the sampler and the prepare roles in one launch
        if merged:
            pass
        elif plan.z_given is None:
            L.check(lib.enslam_sample_rays(N, plan.n_lin, plan.n_surf, _ptr(ro), _ptr(rd), _ptr(gd), plan.bound6,
                                           _ptr(plan.t_lin), _ptr(plan.t_surf), plan.lindisp, _ptr(t_rand),
                                           _ptr(scratch), int(plan.depth_max is not None), _ptr(z), L.STAGE[plan.stage],
                                           ctypes.byref(msc) if msc is not None else None, fptr, st),
                    "enslam_sample_rays")
        elif msc is not None:                       # the samples are given: block marking as a launch of its own
            L.check(lib.enslam_mark_blocks(L.STAGE[plan.stage], N, S, _ptr(ro), _ptr(rd), _ptr(z), ctypes.byref(msc), fptr, st),
                    "enslam_mark_blocks")
        if static:
            for (i, k), vm in zip(static, _grid_cache.get_many([grids[i] for i, _ in static])):
                grids_vm[k] = vm
        conv_args = None
        if conv:
            conv_grids = [grids[i] for i, _ in conv]
            vms, conv_args = _grid_cache.get_many_sparse(conv_grids, [flags[i] for i, _ in conv], arena, defer=True, fresh=cap)
            for (i, k), vm in zip(conv, vms):
                grids_vm[k] = vm
        po, items = nk, []
        for k in plan.kinds:
            items.append((plan.decoders[k], k, tensors[po:po + plan.n_params[k]]))
            po += plan.n_params[k]
        pks, pack_args, pack_commit = packed_decoders(items, arena, defer=True)
        for k, pk in zip(plan.kinds, pks):
            packed[k] = pk
        # ONE launch: pack the stale decoders, convert the touched blocks, clear the backward's accumulators
        nd_, kinds_, structs_, ptrs_ = pack_args if pack_args is not None else (0, None, None, None)
        nc_, srcs_, dsts_, vs_, needs_, valids_, keep_ = conv_args if conv_args is not None else (0, None, None, None, None, None, None)
        nz_, zd_, zv_, zn_ = accum.zero_args(plan, flags) if accum is not None else (0, None, None, None)
        if merged:
            L.check(lib.enslam_sample_prepare(N, plan.n_lin, plan.n_surf, _ptr(ro), _ptr(rd), _ptr(gd), plan.bound6,
                                              _ptr(plan.t_lin), _ptr(plan.t_surf), plan.lindisp, _ptr(t_rand),
                                              _ptr(scratch), int(plan.depth_max is not None), _ptr(z), L.STAGE[plan.stage],
                                              ctypes.byref(msc) if msc is not None else None, fptr, 64, None,
                                              nd_, kinds_, structs_, ptrs_, nz_, zd_, zv_, zn_,
                                              _ptr(accum.zbuf) if accum is not None else None,
                                              accum.n_zero if accum is not None else 0, st), "enslam_sample_prepare")
            if accum is not None:
                accum.clean = True
        elif nd_ or nc_ or accum is not None:
            L.check(lib.enslam_step_prepare(nd_, kinds_, structs_, ptrs_, nc_, srcs_, dsts_, vs_, needs_, valids_, nz_, zd_, zv_, zn_,
                                            _ptr(accum.zbuf) if accum is not None else None,
                                            accum.n_zero if accum is not None else 0, st), "enslam_step_prepare")
            if accum is not None:
                accum.clean = True
        pack_commit()
        del keep_
        sc = _scene_struct(plan.stage, plan.bound6, plan.coarse_bound6, grids_vm, dims, packed)
        depth = torch.empty(N, dtype=torch.float64, device=dev)
        var = torch.empty(N, dtype=torch.float64, device=dev)
        rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
        raw = torch.empty((N * S, 4), dtype=torch.float32, device=dev)
        # a backward will follow: let the forward park the backward's operands (else the backward recomputes them)
        act = None
        # no decoder parameter wants a gradient (tracker; mapper stages with fixed decoders): the light workspace
        act_light = int(not any(ctx.needs_input_grad[5 + nk:]))
        if any(ctx.needs_input_grad):
            n_act = _lib_size('enslam_activation_floats', L.STAGE[plan.stage], N, S, act_light)
            if 0 < n_act * 4 <= ACT_WORKSPACE_LIMIT_BYTES and max(d[0] * d[1] * d[2] for d in dims.values()) < (1 << 29):
                act = torch.empty(n_act, dtype=torch.float32, device=dev)
        loss = None
        # work list of the backward (tiles with non-zero d_raw): filled by whichever kernel produces d_raw; the counter
        # comes zeroed out of this call's arena
        work = wcount = None
        if act is not None and state.use_work_list and any(ctx.needs_input_grad) and SV is None:
            work = torch.empty(N * (S // 16), dtype=torch.int32, device=dev)
            wcount = accum.counter() if accum is not None else arena.take(1, torch.int32)    # (cleared by the prepare launch)
        if plan.loss is None:
            L.check(lib.enslam_render_fwd(L.STAGE[plan.stage], N, S, _ptr(ro), _ptr(rd), _ptr(z), ctypes.byref(sc),
                                          _ptr(depth), _ptr(var), _ptr(rgb), _ptr(raw), _ptr(act), act_light, st),
                    "enslam_render_fwd")
        else:
            lgd, lgc, lw = plan.loss[:3]
            loss = accum.loss_slot() if accum is not None else arena.take(1, torch.float64)
            # the compositing launch also leaves d(loss)/d(raw) for a unit loss gradient: the backward starts at the decoders
            d_raw_unit = torch.empty((N * S, 4), dtype=torch.float32, device=dev) if any(ctx.needs_input_grad) else None
            if len(plan.loss) > 3:              # the tracker's loss (Tracker.py:176-195): (gd, gc, w, inside mask | None, handle_dynamic, 'tracker')
                linside, ldyn = plan.loss[3], plan.loss[4]
                tmp = torch.empty(N, dtype=torch.float64, device=dev)
                L.check(lib.enslam_render_tracker_loss_fwd(L.STAGE[plan.stage], N, S, _ptr(ro), _ptr(rd), _ptr(z), ctypes.byref(sc),
                                                           _ptr(depth), _ptr(var), _ptr(rgb), _ptr(raw), _ptr(act), act_light, _ptr(lgd),
                                                           _ptr(lgc), ctypes.c_float(lw), _ptr(linside), int(bool(ldyn)), _ptr(tmp),
                                                           _ptr(loss), _ptr(d_raw_unit),
                                                           _ptr(work) if d_raw_unit is not None else None,
                                                           _ptr(wcount) if d_raw_unit is not None else None, st),
                        "enslam_render_tracker_loss_fwd")
            else:
                L.check(lib.enslam_render_loss_fwd(L.STAGE[plan.stage], N, S, _ptr(ro), _ptr(rd), _ptr(z), ctypes.byref(sc),
                                                   _ptr(depth), _ptr(var), _ptr(rgb), _ptr(raw), _ptr(act), act_light, _ptr(lgd),
                                                   _ptr(lgc), ctypes.c_float(lw), _ptr(loss), _ptr(d_raw_unit),
                                                   _ptr(work) if d_raw_unit is not None else None,
                                                   _ptr(wcount) if d_raw_unit is not None else None, st),
                        "enslam_render_loss_fwd")
        ctx.sv = None
        if SV is not None:
            # compositing over the real samples only (the kernels above composited the padded rays): contiguous [N, SV] views
            raw_v = raw.view(N, S, 4)[:, :SV].contiguous()
            z_v = z[:, :SV].contiguous()
            L.check(lib.enslam_composite_fwd(N, SV, _ptr(raw_v), _ptr(z_v), _ptr(depth), _ptr(var), _ptr(rgb), None, st),
                    "enslam_composite_fwd")
            ctx.sv = (SV, raw_v, z_v)
        ctx.plan, ctx.S, ctx.dims, ctx.act_light, ctx.accum = plan, S, dims, act_light, accum
        ctx.cap_arena = cap
        ctx.flag_move = flag_move
        if accum is not None:
            for i, k in enumerate(plan.kinds):
                if k in accum.nat_persist:
                    _capture['persist_need'][id(grids[i])] = flags[i]
                elif k in accum.nat_again:
                    # a second render of this grid inside one captured step: its gradient is a fresh buffer that AccumulateGrad adds
                    # into the first call's persistent one, so the blocks it touches must reach that buffer's `prev` flags whatever the
                    # order of the two backwards -- joined to the first call's flags here (moved to `prev` by its finish launch if
                    # that is still to come) and to `prev` itself after this call's backward
                    need_first = _capture['persist_need'].get(id(grids[i]))
                    if need_first is not None:
                        torch.maximum(need_first, flags[i], out=need_first)
        ctx.keep = (ro, rd, z, raw, depth, grids_vm, packed, act, flags)
        ctx.live_guard = _live_grid_guard(grids) if any(ctx.needs_input_grad) and not _cap_tag() else []
        ctx.rgb = rgb if plan.loss is not None else None
        ctx.d_raw_unit = d_raw_unit if plan.loss is not None else None
        ctx.work, ctx.wcount = work, wcount
        state.work = (wcount, N * (S // 16))
        ctx.work_filled = plan.loss is not None and work is not None and d_raw_unit is not None      # (by the forward)
        ctx.grid_shapes = [tuple(g.shape) for g in grids]
        ctx.grid_ids = [id(g) for g in grids]
        ctx.param_like = tensors[nk:]
        if loss is not None:
            ctx.mark_non_differentiable(depth, var, rgb)
            return loss[0], depth, var, rgb
        return depth, var, rgb

    @staticmethod
    def backward(ctx, *gouts):
        lib = L.lib()
        plan, S = ctx.plan, ctx.S
        g_loss = None
        if plan.loss is not None:
            g_loss, g_depth, g_var, g_rgb = gouts[0], None, None, None
        else:
            g_depth, g_var, g_rgb = gouts
        ro, rd, z, raw, depth, grids_vm, packed, act, flags = ctx.keep
        _check_live_grids(ctx.live_guard)
        N, dev, st = ro.shape[0], ro.device, _stream()
        nk = len(plan.kinds)
        needs = ctx.needs_input_grad            # (plan, ro, rd, gd, t_rand, grids..., params...)
        accum = ctx.accum
        need_rays, need_grid, need_par = accum.need_rays, accum.need_grid, accum.need_par

        def prep(g, dtype, shape):
            if g is None:
                return None
            g = g.detach()
            if g.dtype is not dtype:
                g = g.to(dtype)
            if tuple(g.shape) != shape:
                g = g.expand(shape)
            return g if g.is_contiguous() else g.contiguous()

        gD, gV, gC = prep(g_depth, torch.float64, (N,)), prep(g_var, torch.float64, (N,)), prep(g_rgb, torch.float32, (N, 3))
        gL = prep(g_loss, torch.float64, (1,))
        if gD is None and gV is None and gC is None and gL is None:
            return (None,) * len(needs)
        sc = _scene_struct(plan.stage, plan.bound6, plan.coarse_bound6, grids_vm, ctx.dims, packed)
        gg = (L.Grid * 4)()
        gpk = (ctypes.c_void_p * 4)()
        # accumulators: laid out and cleared by the forward's prepare launch (cleared again here on a repeated backward)
        vmg = plan.vm               # grids in the device layout accumulate into their own grad_vm (kept clear by the optimiser)
        sizes, offs, n_grid, n_flat, gbuf, zbuf = accum.sizes, accum.offs, accum.n_grid, accum.n_flat, accum.gbuf, accum.zbuf
        for k in plan.kinds:
            D, H, W = ctx.dims[k]
            gg[k].D, gg[k].H, gg[k].W = D, H, W
            if need_grid[k] and k in vmg:
                gg[k].data = vmg[k].grad_vm.data_ptr()
                vmg[k].has_grad = True
        gbase = gbuf.data_ptr()
        g_grids_vm, g_packed = {}, {}
        for i, k in enumerate(plan.kinds):
            if need_grid[k] and k not in vmg:
                g_grids_vm[k] = gbase + 4 * offs[i]
                gg[k].data = g_grids_vm[k]
        nat = accum.native
        if not accum.clean:
            # a repeated backward of this call (retain_graph): the flat buffer of the first one backs tensors it returned (ray
            # gradients, native grid gradients) -- this one adds into a fresh, zero-filled buffer
            if accum.nat_persist:
                raise L.EnslamError("a second backward through the same render call inside one captured step is not supported for "
                                    "channels_last_3d feature grids (their gradient buffer is the graph's own)")
            zbuf = accum.zbuf = torch.zeros(accum.n_zero, dtype=torch.float32, device=dev)
            n, dsts, vs, need_ptrs = accum.zero_args(plan, flags)
            if n:
                L.check(lib.enslam_zero_blocks(n, dsts, vs, need_ptrs, None, 0, st), "enslam_zero_blocks")
        accum.clean = False
        zbase = zbuf.data_ptr() - 4 * n_grid
        nat_buf = {}
        for k in plan.kinds:
            if nat[k]:
                nat_buf[k] = accum.native_grad(plan, k)
                gg[k].data = nat_buf[k].data_ptr()
        # decoder-parameter gradients leave the backward kernel as per-workgroup partial images (summed by the finish launch)
        # instead of 4.4 M float atomics at its tail; ENSLAM_DW_PARTIALS=0 restores the atomics (A/B aid)
        gpart, part_keep = (ctypes.c_void_p * 4)(), {}
        for i, k in enumerate(plan.kinds):
            if need_par[k]:
                g_packed[k] = zbase + 4 * offs[nk + i]
                gpk[k] = g_packed[k]
                if USE_DW_PARTIALS:
                    part_keep[k] = torch.empty(_lib_size('enslam_bwd_partial_floats', k), dtype=torch.float32, device=dev)
                    gpart[k] = part_keep[k].data_ptr()
        g_ro = g_rd = None
        p_ro = p_rd = None
        if need_rays:
            r0 = offs[2 * nk] - n_grid
            g_ro = zbuf[r0:r0 + 3 * N].view(N, 3)
            g_rd = zbuf[r0 + 3 * N:r0 + 6 * N].view(N, 3)
            p_ro, p_rd = _ptr(g_ro), _ptr(g_rd)
        d_scale = None
        work, wcount = ctx.work, ctx.wcount
        if work is not None and not (gL is not None and ctx.d_raw_unit is not None):
            if ctx.work_filled:                             # a repeated backward appends again: start from an empty list
                wcount.zero_()
            ctx.work_filled = True
        if gL is not None and ctx.d_raw_unit is not None:
            d_raw, d_scale = ctx.d_raw_unit, gL             # unit gradients from the forward, scaled inside the decoder backward
        elif gL is not None:
            d_raw = torch.empty((N * S, 4), dtype=torch.float32, device=dev)
            if len(plan.loss) > 3:
                raise L.EnslamError("the tracker's fused loss keeps its unit gradients from the forward; none were kept")
            lgd, lgc, lw = plan.loss
            L.check(lib.enslam_composite_loss_bwd(N, S, _ptr(raw), _ptr(z), _ptr(depth), _ptr(ctx.rgb), _ptr(lgd), _ptr(lgc),
                                                  ctypes.c_float(lw), _ptr(gL), _ptr(d_raw), _ptr(work), _ptr(wcount), st),
                    "enslam_composite_loss_bwd")
        elif ctx.sv is not None:                            # padded last tile: gradient of the real samples, zeros for the pad
            SV, raw_v, z_v = ctx.sv
            d_raw_v = torch.empty((N, SV, 4), dtype=torch.float32, device=dev)
            L.check(lib.enslam_composite_bwd(N, SV, _ptr(raw_v), _ptr(z_v), _ptr(depth), _ptr(gD), _ptr(gV), _ptr(gC),
                                             _ptr(d_raw_v), st), "enslam_composite_bwd")
            d_raw = torch.zeros((N, S, 4), dtype=torch.float32, device=dev)
            d_raw[:, :SV] = d_raw_v
            d_raw = d_raw.view(N * S, 4)
        else:
            d_raw = torch.empty((N * S, 4), dtype=torch.float32, device=dev)
            L.check(lib.enslam_composite_bwd_list(N, S, _ptr(raw), _ptr(z), _ptr(depth), _ptr(gD), _ptr(gV), _ptr(gC),
                                                  _ptr(d_raw), _ptr(work), _ptr(wcount), st), "enslam_composite_bwd")
        # Ray gradients of the saved-activation path: handed to the ray-gradient role of the finish launch through dgw when
        # that launch exists anyway (grid gradients to transpose back, decoder gradients to unpack: in-kernel ray gradients
        # cost the mapper step's backward 15 us to save 11 in the finish launch), else computed by the decoder kernel itself at
        # the end of each round -- a tracker iteration (fixed map and decoders) then has no finish launch at all: 104 -> 100 us.
        # ENSLAM_INLINE_RAY_GRAD = 1 / 0 forces one or the other.
        finish_needed = any((need_grid[k] and k not in plan.vm and not nat[k]) or need_par[k] for k in plan.kinds) or bool(accum.nat_persist)
        inline_rays = (not finish_needed) if INLINE_RAY_GRAD is None else INLINE_RAY_GRAD
        dgw = None
        # ENSLAM_DEFER_SCATTER=1 / 2 (opt-in, csrc/grid_scatter.hip): the decoder kernels leave dC in the same hand-off and a launch of its
        # own scatters it, first summing in LDS what neighbouring rays add to the same voxel rows -- always / when a decoder WITHOUT
        # parameter gradients has a grid gradient (the reference mapper's fixed occupancy decoders).  Same rule here as in the library.
        defer = False
        if DEFER_SCATTER and act is not None and plan.stage != 'coarse':
            any_grid = any(need_grid[k] for k in plan.kinds)
            light_grid = any(need_grid[k] and not need_par[k] for k in plan.kinds)
            defer = any_grid if DEFER_SCATTER == 1 else light_grid
        if act is not None and ((need_rays and not inline_rays) or defer):
            dgw = torch.empty(_lib_size('enslam_grid_handoff_floats', L.STAGE[plan.stage], N, S), dtype=torch.float32, device=dev)
        ev = plan.state.profile.get('decoder_bwd')
        if ev is not None:                      # bench.py: HIP events around the dominant kernel, on this stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(lib.enslam_decoder_bwd_partials(L.STAGE[plan.stage], N, S, _ptr(ro), _ptr(rd), _ptr(z), ctypes.byref(sc),
                                                _ptr(d_raw), _ptr(d_scale), _ptr(act), ctx.act_light, _ptr(dgw), gg, gpk, gpart, p_ro, p_rd,
                                                _ptr(work), _ptr(wcount), st), "enslam_decoder_bwd")
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        # saved-activation path: the ray gradients (corner re-gather from the hand-off in dgw) ride in the finish launch
        ray_pending = dgw is not None and need_rays and plan.stage != 'coarse'
        out = [None, g_ro if needs[1] else None, g_rd if needs[2] else None, None, None]
        # ONE launch: grid gradients back to the callers' [1,32,D,H,W] layout + decoder gradients unpacked into views of
        # one flat buffer shaped like the parameters
        conv = [(i, k) for i, k in enumerate(plan.kinds) if need_grid[k] and k not in vmg and not nat[k]]
        grid_out = {}
        for i, k in enumerate(plan.kinds):
            if nat[k]:                                      # (the buffer the backward kernel has just added into)
                grid_out[k] = _native_grad_view(nat_buf[k], ctx.dims[k])
                if k in accum.nat_persist:
                    plan.state.flags[ctx.grid_ids[i]] = accum.nat_persist[k][1]     # readers use `prev` from the finish launch on
        del nat_buf
        nc = len(conv)
        srcs, dsts, vs = (ctypes.c_void_p * max(nc, 1))(), (ctypes.c_void_p * max(nc, 1))(), (ctypes.c_int64 * max(nc, 1))()
        need_ptrs = (ctypes.c_void_p * max(nc, 1))()
        # Under hipGraph capture the dense gradient of a grid is the same memory at every replay: the finish launch then only
        # rewrites the blocks touched now or one replay earlier (enslam_step_finish_rays_prev) instead of zero-filling the
        # ~85 % of the tensor no ray came near.  The buffers start zeroed by end_capture(); only their STORAGES are kept
        # there (a second reference to the tensor itself would make AccumulateGrad clone the 16 MB instead of adopting it).
        persistent = _capturing() and _capture['init_zero'] is not None and nc > 0
        # A second backward into the same grid inside one captured step (AccumulateGrad adds its gradient into the first
        # one's persistent buffer) must not be persistent itself: it returns a fresh, fully written gradient, and the blocks
        # it touched join the first buffer's flags (a captured elementwise max below), so the next replay clears them there.
        again = [ctx.grid_ids[i] for i, _k in conv if persistent and ctx.grid_ids[i] in _capture['persist']]
        if again:
            persistent = False
        prev_ptrs = (ctypes.c_void_p * max(nc, 1))() if persistent else None
        prev_keep = []
        for j, (i, k) in enumerate(conv):
            g = torch.empty(ctx.grid_shapes[i], dtype=torch.float32, device=dev)
            grid_out[k] = g
            srcs[j], dsts[j], vs[j], need_ptrs[j] = g_grids_vm[k], g.data_ptr(), sizes[i] // 32, flags[i].data_ptr()
            if persistent:
                pv = torch.empty(flags[i].numel(), dtype=torch.uint8, device=dev)
                prev_ptrs[j] = pv.data_ptr()
                prev_keep.append(pv)
                plan.state.flags[ctx.grid_ids[i]] = pv      # the launch moves the flags there (last_block_flags)
                _capture['init_zero'] += [g.untyped_storage(), pv.untyped_storage()]
                _capture['persist'][ctx.grid_ids[i]] = (g.data_ptr(), pv)
                _persist_register(g, pv)
        for k in plan.kinds:
            out.append(grid_out.get(k))
        kinds_p = [k for k in plan.kinds if need_par[k]]
        views_by_kind = {}
        npk = len(kinds_p)
        kind_arr, pk_arr, structs = (ctypes.c_int32 * max(npk, 1))(), (ctypes.c_void_p * max(npk, 1))(), (L.MlpParams * max(npk, 1))()
        part_arr = (ctypes.c_void_p * max(npk, 1))()
        if kinds_p:
            # the gradients of all decoder parameters are views of ONE fresh flat buffer, shaped like the parameters by one
            # C++ call (69 x split + view in Python cost 0.19 ms per step); the pointer structs come from the flat base and
            # the parameters' (fixed) sizes
            like, po = [], nk
            like_by_kind = {}
            for k in plan.kinds:
                if need_par[k]:
                    like_by_kind[k] = ctx.param_like[po - nk:po - nk + plan.n_params[k]]
                    like += like_by_kind[k]
                po += plan.n_params[k]
            total = sum(t.numel() for t in like)
            pflat = torch.empty(total, dtype=torch.float32, device=dev)
            views = torch._C._nn.unflatten_dense_tensors(pflat, like)
            base, vo = pflat.data_ptr(), 0
            for j, k in enumerate(kinds_p):
                n = plan.n_params[k]
                views_by_kind[k] = list(views[vo:vo + n])
                vo += n
                kind_arr[j], pk_arr[j] = k, g_packed[k]
                part_arr[j] = gpart[k]
                structs[j], base = _flat_params_struct(k, like_by_kind[k], base)
        mv = ctx.flag_move
        if nc or npk or ray_pending or mv is not None:
            L.check(lib.enslam_step_finish_native(nc, srcs, dsts, vs, need_ptrs, prev_ptrs, npk, kind_arr, pk_arr, part_arr, structs,
                                                  L.STAGE[plan.stage], N if ray_pending else 0, S, _ptr(ro), _ptr(rd), _ptr(z),
                                                  ctypes.byref(sc), _ptr(dgw) if ray_pending else None, p_ro, p_rd, _ptr(work),
                                                  _ptr(wcount), _ptr(mv[0]) if mv is not None else None,
                                                  _ptr(mv[1]) if mv is not None else None, mv[0].numel() if mv is not None else 0, st),
                    "enslam_step_finish")
        for i, k in enumerate(plan.kinds):
            gid = ctx.grid_ids[i]
            if (need_grid[k] and k not in vmg and not nat[k] and gid in again) or k in accum.nat_again:
                # (captured: runs at every replay, after the finish launch)
                pv_first = _capture['persist'][gid][1] if _capture['persist'] is not None and gid in _capture['persist'] else None
                if pv_first is not None:
                    torch.maximum(pv_first, flags[i], out=pv_first)
                    if ctx.cap_arena:
                        flags[i].zero_()                    # no fill node re-zeroes a captured step's arena
        for k in plan.kinds:
            out += views_by_kind.get(k, [None] * plan.n_params[k])
        return tuple(out)       # (the saved buffers go with the graph; kept so that retain_graph backwards work)


# The forward keeps the backward's operands (1.3 KB per sample and decoder: 172 MB at 1000 rays x 48, colour
# stage) when their total stays under this limit; above it the backward recomputes them (slower, no extra memory).
ACT_WORKSPACE_LIMIT_BYTES = 8 << 30

INLINE_RAY_GRAD = {'1': True, '0': False}.get(os.environ.get('ENSLAM_INLINE_RAY_GRAD', ''), None)     # None: decided per call
DEFER_SCATTER = {'1': 1, '2': 2}.get(os.environ.get('ENSLAM_DEFER_SCATTER', ''), 0)    # feature-gradient scatter as its own launch (csrc/grid_scatter.hip): off (default) / always / by policy
# weight gradients as per-workgroup partial images summed by the finish launch instead of float atomics at the backward's tail:
# backward 138 -> 132 us, finish launch 22 -> 28 us (17.6 MB more to read), step unchanged -- off by default (round 3)
USE_DW_PARTIALS = os.environ.get('ENSLAM_DW_PARTIALS', '0') == '1'


def last_active_tile_fraction():
    """`RenderState.last_active_tile_fraction()` of whichever renderer rendered last in this process."""
    return _latest_state[0].last_active_tile_fraction()


# torch.autograd.Function.apply first scans every argument for functorch wrappers (30-35 us for the ~80 arguments of a
# colour-stage call); no transform is ever active on this path, so the C-level apply is bound directly.  Falls back to the
# public entry when the private attribute is missing or a functorch transform IS active.
try:
    _render_apply = torch._C._FunctionBase.__dict__['apply'].__get__(None, _RenderFn)
except Exception:           # pragma: no cover
    _render_apply = None


# ------------------------------------------------------------------------------------------------
# step plans: the Python-driven call with its host work cached (include/enslam_hip.h, "Step plans")
# ------------------------------------------------------------------------------------------------
# A render call of a loop repeats with the same tensors (the optimiser changes their values in place): everything the host
# derives from them -- which grids convert, where every buffer lies, the parameter pointer tables -- is built once into an
# enslam_step_plan and found again by identity; a call then allocates three blobs and makes ONE library call per direction
# (0.2 ms of Python per direction before).  Taken for plain calls only: sampled rays with gt_depth, whole 16-sample tiles, no
# hipGraph capture of this library in progress (its persistent gradients are _RenderFn's), no VoxelMajorGrid, no given sample
# distances, no per-kernel profiling.  ENSLAM_STEP_PLANS=0 switches it off (same numbers either way).
STEP_PLANS = os.environ.get('ENSLAM_STEP_PLANS', '1') == '1'
_N_PARAMS = 23


class _PlanEntry:
    __slots__ = ('plan', 'layout', 'tensors', 'ptrs', 'rgrad', 'kinds', 'modes', 'dims', 'par_grad', 'need_rays', 'like', 'keep',
                 'n_in', 'flag_spec', 'ntiles')


plan_stats = {'built': 0, 'hits': 0, 'declined': 0}      # (diagnostic counters; tests read them)
# > 0 while graph.GraphedStep runs the eager warm-up iterations of a step it is about to capture: those must leave the caches the
# capture will read (packed decoders of fixed decoders, converted grids) warm, which only _RenderFn's route fills
plans_suspended = [0]
_plan_entries = {}              # (stage, N, n_lin, n_surf, lindisp, loss key, work list, t_lin ptr, t_surf ptr, bound key) -> [entries]


def _plan_build(rplan, N, rays_o, rays_d, grids, params_flat, loss_key):
    lib = L.lib()
    e = _PlanEntry()
    P, kinds = L.StepPlan(), rplan.kinds
    nk = len(kinds)
    P.stage, P.n_rays, P.n_lin, P.n_surf, P.lindisp = L.STAGE[rplan.stage], N, rplan.n_lin, rplan.n_surf, rplan.lindisp
    P.need_rays = int(rays_o.requires_grad or rays_d.requires_grad)
    P.use_work_list = int(rplan.state.use_work_list)
    P.loss_kind, P.use_color, P.w_color = loss_key
    e.modes, e.dims, e.par_grad, like = {}, {}, {}, []
    po = 0
    off = 0
    for i, k in enumerate(kinds):
        g = grids[i]
        _check_grid(g)
        nat = is_native_grid(g)
        e.modes[k] = (2 if nat else 3) if g.requires_grad else 1
        e.dims[k] = tuple(int(x) for x in g.shape[2:])
        P.grid_mode[k] = e.modes[k]
        P.grid_D[k], P.grid_H[k], P.grid_W[k] = e.dims[k]
        ps = params_flat[po:po + _N_PARAMS]
        po += _N_PARAMS
        _check_params(k, ps)
        P.params[k] = _fill_params_struct(k, ps)
        e.par_grad[k] = any(p.requires_grad for p in ps)
        P.par_grad[k] = int(e.par_grad[k])
        if e.par_grad[k]:
            for j, t in enumerate(ps):
                P.pgrad_off[k][j] = off
                off += t.numel()
            like += list(ps)
    P.pgrad_floats = off
    P.act_light = int(not any(e.par_grad.values()))
    P.bound, P.coarse_bound = rplan.bound6, rplan.coarse_bound6
    P.t_lin, P.t_surf = _ptr(rplan.t_lin), _ptr(rplan.t_surf)
    Lay = L.StepLayout()
    rc = lib.enslam_plan_layout(ctypes.byref(P), ctypes.byref(Lay))
    if rc != 0:
        return None
    n_act = _lib_size('enslam_activation_floats', P.stage, N, Lay.n_samples, P.act_light)
    if not (0 < n_act * 4 <= ACT_WORKSPACE_LIMIT_BYTES) or max(d[0] * d[1] * d[2] for d in e.dims.values()) >= (1 << 29):
        return None
    e.plan, e.layout, e.kinds, e.like, e.need_rays = P, Lay, kinds, like, bool(P.need_rays)
    # (grids by weak reference: a cached plan must not keep a replaced 45 MB map alive; the decoders' parameters are small)
    e.tensors = [weakref.ref(g) for g in grids] + list(params_flat)
    e.ptrs = [t.data_ptr() for t in params_flat] + [rplan.t_lin.data_ptr(), rplan.t_surf.data_ptr() if rplan.t_surf is not None else 0]
    e.rgrad = [t.requires_grad for t in [rays_o, rays_d] + list(grids) + list(params_flat)]
    e.keep = (rplan.t_lin, rplan.t_surf)
    e.n_in = 4 + nk + len(params_flat)
    e.ntiles = N * (Lay.n_samples // 16)
    return e


def _plan_lookup(rplan, N, rays_o, rays_d, grids, params_flat, loss_key):
    tl, ts = rplan.t_lin, rplan.t_surf
    key = (rplan.stage, N, rplan.n_lin, rplan.n_surf, rplan.lindisp, loss_key, rplan.state.use_work_list, id(rplan.state),
           tuple(rplan.bound6), tuple(rplan.coarse_bound6))
    lst = _plan_entries.get(key)
    tensors = [rays_o, rays_d] + list(grids) + list(params_flat)
    nk = len(grids)
    if lst is not None:
        for e in lst:
            t0 = e.tensors
            # rays may be new tensors every step (only their gradient flag matters); grids and parameters are found by identity
            if len(t0) + 2 == len(tensors) and all(r() is g for r, g in zip(t0[:nk], grids)) and \
                    all(a is b for a, b in zip(t0[nk:], params_flat)) and \
                    e.rgrad == [t.requires_grad for t in tensors] and \
                    e.ptrs == [t.data_ptr() for t in params_flat] + [tl.data_ptr(), ts.data_ptr() if ts is not None else 0] and \
                    all(tuple(g.shape[2:]) == e.dims[k] and ((e.modes[k] == 3) != is_native_grid(g) or e.modes[k] == 1)
                        for g, k in zip(grids, rplan.kinds)):
                plan_stats['hits'] += 1
                return e
    e = _plan_build(rplan, N, rays_o, rays_d, grids, params_flat, loss_key)
    if e is None:
        plan_stats['declined'] += 1
        return None
    plan_stats['built'] += 1
    if len(_plan_entries) > 64:
        _plan_entries.clear()
    _plan_entries.setdefault(key, []).insert(0, e)
    del _plan_entries[key][4:]
    return e


def _blob_view(blob, off, nbytes, dtype):
    return blob[off:off + nbytes].view(dtype)


class _PlanFn(torch.autograd.Function):
    """_RenderFn for calls served by a step plan: inputs (entry, rplan, rays_o, rays_d, gt_depth, gt_color | None, *grids, *params)."""

    @staticmethod
    def forward(ctx, e, rplan, rays_o, rays_d, gt_depth, gt_color, *tensors):
        lib = L.lib()
        ctx.set_materialize_grads(False)
        P, Lay = e.plan, e.layout
        nk = len(e.kinds)
        dev = rays_o.device
        ro, rd = _f32c(rays_o), _f32c(rays_d)
        gd = _f32c(gt_depth).reshape(-1)
        gc = gt_color                               # (already detached contiguous float32 [N,3] or None: Renderer)
        u8 = torch.uint8
        scratch = torch.empty(Lay.scratch_bytes, dtype=u8, device=dev)
        gradb = torch.empty(Lay.grad_bytes, dtype=u8, device=dev)
        outb = torch.empty(Lay.out_bytes, dtype=u8, device=dev)
        gv = (ctypes.c_void_p * 4)()
        hold = []
        static = [(i, k) for i, k in enumerate(e.kinds) if e.modes[k] == 1 and not is_native_grid(tensors[i])]
        if static:
            for (i, k), vm in zip(static, _grid_cache.get_many([tensors[i] for i, _ in static])):
                gv[k] = vm.data_ptr()
                hold.append(vm)
        for i, k in enumerate(e.kinds):
            if gv[k]:
                continue
            g = tensors[i].detach()
            if e.modes[k] == 3 and not g.is_contiguous():
                g = g.contiguous()
            gv[k] = g.data_ptr()
            hold.append(g)
        st = _stream()
        L.check(lib.enslam_plan_forward(ctypes.byref(P), ctypes.byref(Lay), scratch.data_ptr(), gradb.data_ptr(), outb.data_ptr(),
                                        _ptr(ro), _ptr(rd), _ptr(gd), _ptr(gc), _ptr(rplan.depth_max), gv, st), "enslam_plan_forward")
        N = P.n_rays
        depth = _blob_view(outb, Lay.o_depth, 8 * N, torch.float64)
        var = _blob_view(outb, Lay.o_var, 8 * N, torch.float64)
        rgb = _blob_view(outb, Lay.o_rgb, 12 * N, torch.float32).view(N, 3)
        state = rplan.state
        _latest_state[0] = state
        state.flags = {}
        for i, k in enumerate(e.kinds):
            if e.modes[k] >= 2:
                state.flags[id(tensors[i])] = scratch[Lay.s_flags[k]:Lay.s_flags[k] + (e.dims[k][0] * e.dims[k][1] * e.dims[k][2] + 63) // 64]
        state.work = (_blob_view(gradb, Lay.g_counter, 4, torch.int32), e.ntiles) if P.use_work_list else (None, 0)
        ctx.e, ctx.rplan = e, rplan
        ctx.keep = (ro, rd, gd, gc, scratch, gradb, outb, gv, hold)
        ctx.live_guard = _live_grid_guard(tensors[:len(e.kinds)]) if any(ctx.needs_input_grad) else []
        ctx.calls = 0
        if P.loss_kind == 1:
            loss = _blob_view(outb, Lay.o_loss, 8, torch.float64)
            ctx.mark_non_differentiable(depth, var, rgb)
            return loss[0], depth, var, rgb
        return depth, var, rgb

    @staticmethod
    def backward(ctx, *gouts):
        lib = L.lib()
        e = ctx.e
        P, Lay = e.plan, e.layout
        ro, rd, gd, gc, scratch, gradb, outb, gv, hold = ctx.keep
        _check_live_grids(ctx.live_guard)
        N, dev = P.n_rays, ro.device
        n_in = e.n_in + 2
        if P.loss_kind == 1:
            g_loss, g_depth, g_var, g_rgb = gouts[0], None, None, None
        else:
            g_loss = None
            g_depth, g_var, g_rgb = gouts

        def prep(g, dtype, shape):
            if g is None:
                return None
            g = g.detach()
            if g.dtype is not dtype:
                g = g.to(dtype)
            if tuple(g.shape) != shape:
                g = g.expand(shape)
            return g if g.is_contiguous() else g.contiguous()

        gD, gV, gC = prep(g_depth, torch.float64, (N,)), prep(g_var, torch.float64, (N,)), prep(g_rgb, torch.float32, (N, 3))
        gL = prep(g_loss, torch.float64, (1,))
        if gD is None and gV is None and gC is None and gL is None:
            return (None,) * n_in
        st = _stream()
        if ctx.calls > 0:
            # a repeated backward of this call (retain_graph): the first one's gradient blob backs tensors it returned -- this
            # one adds into a fresh, cleared blob (the work list of a fused-loss forward keeps its counter) and cleared
            # channel-major accumulators
            old = gradb
            gradb = torch.zeros(Lay.grad_bytes, dtype=torch.uint8, device=dev)
            if P.loss_kind == 1 and P.use_work_list:
                gradb[Lay.g_counter:Lay.g_counter + 4].copy_(old[Lay.g_counter:Lay.g_counter + 4])
            m3 = [k for k in e.kinds if e.modes[k] == 3]
            if m3:
                n = len(m3)
                dsts, vs, nd = (ctypes.c_void_p * n)(), (ctypes.c_int64 * n)(), (ctypes.c_void_p * n)()
                for j, k in enumerate(m3):
                    dsts[j] = scratch.data_ptr() + Lay.s_gacc[k]
                    vs[j] = e.dims[k][0] * e.dims[k][1] * e.dims[k][2]
                    nd[j] = scratch.data_ptr() + Lay.s_flags[k]
                L.check(lib.enslam_zero_blocks(n, dsts, vs, nd, None, 0, st), "enslam_zero_blocks")
        ctx.calls += 1
        L.check(lib.enslam_plan_backward(ctypes.byref(P), ctypes.byref(Lay), scratch.data_ptr(), gradb.data_ptr(), outb.data_ptr(),
                                         _ptr(ro), _ptr(rd), _ptr(gd), _ptr(gc), gv, _ptr(gD), _ptr(gV), _ptr(gC), _ptr(gL), st),
                "enslam_plan_backward")
        needs = ctx.needs_input_grad
        out = [None, None, None, None, None, None]      # (entry, rplan, rays_o, rays_d, gt_depth, gt_color)
        if e.need_rays:
            if needs[2]:
                out[2] = _blob_view(gradb, Lay.g_ro, 12 * N, torch.float32).view(N, 3)
            if needs[3]:
                out[3] = _blob_view(gradb, Lay.g_rd, 12 * N, torch.float32).view(N, 3)
        for k in e.kinds:
            D, H, W = e.dims[k]
            V = D * H * W
            if e.modes[k] == 2:
                out.append(_native_grad_view(_blob_view(gradb, Lay.g_nat[k], 128 * V, torch.float32), e.dims[k]))
            elif e.modes[k] == 3:
                out.append(_blob_view(gradb, Lay.g_dense[k], 128 * V, torch.float32).view(1, 32, D, H, W))
            else:
                out.append(None)
        if e.like:
            views = torch._C._nn.unflatten_dense_tensors(_blob_view(gradb, Lay.g_params, 4 * P.pgrad_floats, torch.float32), e.like)
            vo = 0
            for k in e.kinds:
                if e.par_grad[k]:
                    out += list(views[vo:vo + _N_PARAMS])
                    vo += _N_PARAMS
                else:
                    out += [None] * _N_PARAMS
        else:
            out += [None] * (_N_PARAMS * len(e.kinds))
        return tuple(out)


try:
    _plan_apply = torch._C._FunctionBase.__dict__['apply'].__get__(None, _PlanFn)
except Exception:           # pragma: no cover
    _plan_apply = None


def _plan_render(plan, rays_o, rays_d, gt_depth, t_rand, grids, params_flat):
    """-> outputs of _PlanFn, or None when the call is not one a step plan serves."""
    if (not STEP_PLANS or _capture['active'] or plans_suspended[0] or plan.z_given is not None or t_rand is not None or gt_depth is None or plan.vm
            or plan.stage == 'coarse' or plan.state.profile or not rays_o.is_cuda or not torch.is_grad_enabled()):
        return None
    N = rays_o.shape[0]
    S = plan.n_lin + plan.n_surf
    if N <= 0 or S % 16 != 0 or S > 64 or len(params_flat) != _N_PARAMS * len(plan.kinds):
        return None
    loss_key, gc = (0, 0, 0.0), None
    if plan.loss is not None:
        if len(plan.loss) != 3:
            return None
        lgd, gc, lw = plan.loss
        if lgd.shape[0] != N:
            return None
        loss_key = (1, int(gc is not None), float(lw))
    if not any(t.requires_grad for t in grids) and not any(t.requires_grad for t in params_flat) and \
            not (rays_o.requires_grad or rays_d.requires_grad):
        return None                                 # (forward-only calls: nothing to save; _RenderFn's route is as cheap)
    e = _plan_lookup(plan, N, rays_o, rays_d, grids, params_flat, loss_key)
    if e is None:
        return None
    if _plan_apply is not None and not torch._C._are_functorch_transforms_active():
        return _plan_apply(e, plan, rays_o, rays_d, gt_depth, gc, *grids, *params_flat)
    return _PlanFn.apply(e, plan, rays_o, rays_d, gt_depth, gc, *grids, *params_flat)


def render(plan, rays_o, rays_d, gt_depth, t_rand, grids, params_flat):
    out = _plan_render(plan, rays_o, rays_d, gt_depth, t_rand, grids, params_flat)
    if out is not None:
        return out
    if _render_apply is not None and not torch._C._are_functorch_transforms_active():
        return _render_apply(plan, rays_o, rays_d, gt_depth, t_rand, *grids, *params_flat)
    return _RenderFn.apply(plan, rays_o, rays_d, gt_depth, t_rand, *grids, *params_flat)


# ------------------------------------------------------------------------------------------------
# forward-only helpers
# ------------------------------------------------------------------------------------------------
def eval_points(p, decoders, c, stage, bound, apply_mask=True, coarse_bound=None):
    """raw [P,4] float32 for points p [P,3] (any float dtype; evaluated as float64 like the reference)."""
    lib = L.lib()
    _require_hip(p, "points")
    if torch.is_grad_enabled() and (p.requires_grad or any(
            c[L.GRID_NAMES[k]].requires_grad for k in stage_kinds(stage))):
        raise NotImplementedError("eval_points is forward-only on the HIP path (its differentiable use, "
                                  "Renderer.regulation, belongs to the iMAP mode); wrap the call in torch.no_grad()")
    pts = p.detach().to(torch.float64).contiguous()
    P = pts.shape[0]
    kinds = stage_kinds(stage)
    grids_vm, dims, packed = {}, {}, {}
    for k in kinds:
        g = c[L.GRID_NAMES[k]]
        grids_vm[k] = g.vm if isinstance(g, VoxelMajorGrid) else _grid_cache.get(g)
        dims[k] = tuple(g.shape[2:])
        packed[k] = packed_decoder(getattr(decoders, L.MLP_NAMES[k]), k)
    if coarse_bound is None:
        coarse_bound = decoders.coarse_decoder.bound if 0 in kinds else bound
    sc = _scene_struct(stage, bound6(bound), bound6(coarse_bound), grids_vm, dims, packed)
    raw = torch.empty((P, 4), dtype=torch.float32, device=p.device)
    L.check(lib.enslam_eval_points(L.STAGE[stage], P, _ptr(pts), ctypes.byref(sc), int(apply_mask), _ptr(raw),
                                   _stream()), "enslam_eval_points")
    return raw


def voxel_index(points, bound, shape):
    """Parity helper: (ix,iy,iz int32, fx,fy,fz float32) the gather uses for float64 points [P,3]."""
    lib = L.lib()
    pts = points.detach().to(torch.float64).contiguous()
    P, dev = pts.shape[0], pts.device
    outs = [torch.empty(P, dtype=torch.int32, device=dev) for _ in range(3)] + \
           [torch.empty(P, dtype=torch.float32, device=dev) for _ in range(3)]
    D, H, W = shape
    L.check(lib.enslam_voxel_index(P, _ptr(pts), bound6(bound), D, H, W, *[_ptr(t) for t in outs], _stream()),
            "enslam_voxel_index")
    return outs


def gather_pixels(idx, H0, W0, ww, depth, color):
    """(pix_i, pix_j, depth samples, colour samples) of the window pixels `idx` (int64 [n], row-major inside the window that
    starts at (H0, W0) and is ww wide): what common.get_sample_uv computes with 2 linspace + 2 index-arithmetic + 4 indexing
    launches, in one (enslam_gather_pixels).  depth float32 [H, W], color float32 / float64 [H, W, 3], all on the GPU."""
    n = int(idx.shape[0])
    dev = idx.device
    depth, color, idx = depth.contiguous(), color.contiguous(), idx.contiguous()
    oi = torch.empty(n, dtype=torch.float32, device=dev)
    oj = torch.empty(n, dtype=torch.float32, device=dev)
    od = torch.empty(n, dtype=torch.float32, device=dev)
    oc = torch.empty((n, 3), dtype=color.dtype, device=dev)
    L.check(L.lib().enslam_gather_pixels(n, _ptr(idx), int(H0), int(W0), int(ww), int(depth.shape[1]), int(depth.shape[0]), _ptr(depth),
                                         _ptr(color), int(color.dtype == torch.float64), _ptr(oi), _ptr(oj), _ptr(od), _ptr(oc),
                                         _stream()), "enslam_gather_pixels")
    return oi, oj, od, oc


def gather_pixels_ok(idx, depth, color):
    return (idx.is_cuda and depth.is_cuda and color.is_cuda and idx.dtype == torch.int64 and depth.dtype == torch.float32
            and depth.dim() == 2 and color.dim() == 3 and color.shape[2] == 3 and tuple(color.shape[:2]) == tuple(depth.shape)
            and color.dtype in (torch.float32, torch.float64) and not depth.requires_grad and not color.requires_grad)


def fourier_sincos(x):
    """Parity helper: (sin, cos) float32 of float32 arguments as the embedding kernels evaluate them."""
    _require_hip(x, "x")
    xx = x.detach().contiguous().float().reshape(-1)
    s, c = torch.empty_like(xx), torch.empty_like(xx)
    L.check(L.lib().enslam_fourier_sincos(xx.numel(), _ptr(xx), _ptr(s), _ptr(c), _stream()), "enslam_fourier_sincos")
    return s.reshape(x.shape), c.reshape(x.shape)


def ray_points(rays_o, rays_d, z_vals, bound):
    """Parity helper: float64 sample points [N*S,3] and the strict in-bound mask."""
    lib = L.lib()
    N, S = z_vals.shape
    pts = torch.empty((N * S, 3), dtype=torch.float64, device=z_vals.device)
    mask = torch.empty(N * S, dtype=torch.uint8, device=z_vals.device)
    L.check(lib.enslam_ray_points(N, S, _ptr(rays_o.contiguous().float()), _ptr(rays_d.contiguous().float()),
                                  _ptr(z_vals.contiguous()), bound6(bound), _ptr(pts), _ptr(mask), _stream()),
            "enslam_ray_points")
    return pts, mask.bool()


def sample_rays(rays_o, rays_d, gt_depth, bound, n_lin, n_surf, lindisp=False, t_rand=None, depth_max=None):
    """z_vals float64 [N,S] (Renderer.py:83-171)."""
    lib = L.lib()
    _require_hip(rays_o, "rays")
    N, dev = rays_o.shape[0], rays_o.device
    S = n_lin + (n_surf if gt_depth is not None else 0)
    t_lin = torch.linspace(0., 1., steps=n_lin, device=dev)
    t_surf = torch.linspace(0., 1., steps=max(n_surf, 1), device=dev).double() if n_surf > 0 else None
    z = torch.empty((N, S), dtype=torch.float64, device=dev)
    scratch = torch.empty(2, dtype=torch.float32, device=dev)
    gd = gt_depth.contiguous().float().reshape(-1) if gt_depth is not None else None
    if depth_max is not None:
        scratch = depth_max
    L.check(lib.enslam_sample_rays(N, n_lin, n_surf, _ptr(rays_o.contiguous().float()),
                                   _ptr(rays_d.contiguous().float()), _ptr(gd), bound6(bound), _ptr(t_lin),
                                   _ptr(t_surf), int(bool(lindisp)), _ptr(t_rand), _ptr(scratch),
                                   int(depth_max is not None), _ptr(z), 0, None, None, _stream()), "enslam_sample_rays")
    return z


def batch_depth_max(gt_depth):
    """{max(gt_depth), fl32(max*1.2)} float32 [2]: what a ray-sharded caller passes to every shard
    (Renderer.py:110,145 take these maxima over the whole batch)."""
    m = gt_depth.detach().float().max().reshape(1)
    return torch.cat([m, m * 1.2]).contiguous()


class _CompositeFn(torch.autograd.Function):
    """raw2outputs_nerf_color with occupancy=True as one HIP kernel each way; gradient to `raw` only."""

    @staticmethod
    def forward(ctx, raw, z_vals):
        lib = L.lib()
        _require_hip(raw, "raw")
        N, S = z_vals.shape
        if S > 64:
            raise L.EnslamError("composite kernels handle at most 64 samples per ray")
        r = raw.detach().contiguous().float()
        z = z_vals.detach().contiguous().double()
        dev = raw.device
        depth = torch.empty(N, dtype=torch.float64, device=dev)
        var = torch.empty(N, dtype=torch.float64, device=dev)
        rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
        w = torch.empty((N, S), dtype=torch.float32, device=dev)
        L.check(lib.enslam_composite_fwd(N, S, _ptr(r), _ptr(z), _ptr(depth), _ptr(var), _ptr(rgb), _ptr(w), _stream()),
                "enslam_composite_fwd")
        ctx.keep = (r, z, depth)
        ctx.mark_non_differentiable(w)
        return depth, var, rgb, w

    @staticmethod
    def backward(ctx, g_depth, g_var, g_rgb, _gw):
        lib = L.lib()
        r, z, depth = ctx.keep
        N, S = z.shape
        gD = g_depth.detach().double().contiguous() if g_depth is not None else None
        gV = g_var.detach().double().contiguous() if g_var is not None else None
        gC = g_rgb.detach().float().contiguous() if g_rgb is not None else None
        d_raw = torch.empty((N, S, 4), dtype=torch.float32, device=r.device)
        L.check(lib.enslam_composite_bwd(N, S, _ptr(r), _ptr(z), _ptr(depth), _ptr(gD), _ptr(gV), _ptr(gC),
                                         _ptr(d_raw), _stream()), "enslam_composite_bwd")
        return d_raw, None


def composite(raw, z_vals):
    """(depth f64 [N], var f64 [N], rgb f32 [N,3], weights f32 [N,S]) from raw [N,S,4], z_vals [N,S]."""
    return _CompositeFn.apply(raw, z_vals)
