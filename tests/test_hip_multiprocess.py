"""The reference's process topology around the rendering path (EvenNICER_SLAM.py:75-95,313-332): the map and the decoders live
in device memory that SPAWNED processes share over CUDA IPC -- a mapper process updates them in place (Mapper.py:633-641, the
optimiser on `shared_decoders`), a tracker process renders from them (Tracker.py:248-260: from clones / a deep copy taken
every frame).  These tests start fresh child processes (spawn context), let a "mapper" child rewrite shared grids and
`share_memory()` decoders in place on the device, and let a "tracker" child render before and after -- from clones as the
reference does, and straight from the shared memory -- with the grids in the reference's contiguous layout and as
channels_last_3d tensors.  Every child-side render must equal a single-process render of the same values bit for bit (the
forward has no atomics).

What this pins: the library's caches are keyed on (tensor identity, `_version`), and `_version` is per process -- a write by
another process is invisible to it.  The reference's own flow (clones) never meets that; rendering straight from shared
memory does, and has to be declared (`functional.external_writers(True)`): the last test shows the stale result without the
declaration and the correct one with it."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRIDS = ('grid_middle', 'grid_fine', 'grid_color')


def _render(E, renderer, grids, model, rays):
    with torch.no_grad():
        d, v, c = renderer.render_batch_ray(grids, model, rays['rays_d'], rays['rays_o'], 'cuda:0', 'color', gt_depth=rays['gt_depth'])
    torch.cuda.synchronize()
    return d.cpu().numpy(), v.cpu().numpy(), c.cpu().numpy()


def _tracker(shared_c, shared_decoders, ev_first, ev_updated, q, declare):
    """Child process: render before and after the mapper's update -- from clones (Tracker.update_para_from_mapping) and
    directly from the shared tensors."""
    import copy
    import evennicer_slam_amd as E
    import evennicer_slam_amd.functional as EF
    from tests.hip_util import tiny_on_gpu
    s, bound, _model, _grids, rays, renderer = tiny_on_gpu()
    if declare:
        EF.external_writers(True)
    out = {}
    for phase in ('before', 'after'):
        # the reference's tracker: its own copy of the map and of the decoders, taken now
        c = {k: v.clone() for k, v in shared_c.items()}
        decoders = copy.deepcopy(shared_decoders)
        out['clone_' + phase] = _render(E, renderer, c, decoders, rays)
        # ... and straight from the shared memory
        out['direct_' + phase] = _render(E, renderer, shared_c, shared_decoders, rays)
        if phase == 'before':
            ev_first.set()
            assert ev_updated.wait(300)
    q.put(out)


def _mapper(shared_c, shared_decoders, ev_first, ev_updated):
    """Child process: once the tracker has rendered, rewrite the map and the decoders IN PLACE on the device."""
    assert ev_first.wait(300)
    with torch.no_grad():
        for k in GRIDS:
            g = shared_c[k]
            mask = torch.zeros(g.shape[2:], dtype=torch.bool, device=g.device)
            mask[::2] = True
            val = g[0].permute(1, 2, 3, 0)                           # the reference's masked write-back (Mapper.py:633-641)
            val[mask] = val[mask] * 1.5 + 0.01
        for p in shared_decoders.parameters():
            p.mul_(1.0 + 1.0 / 64)
    torch.cuda.synchronize()
    ev_updated.set()


def _run(layout, declare):
    import torch.multiprocessing as mp
    import evennicer_slam_amd as E
    from tests.hip_util import as_layout, tiny_on_gpu
    s, bound, model, grids, rays, renderer = tiny_on_gpu()
    shared_c = {k: as_layout(v, layout) for k, v in grids.items()}
    shared_c = {k: v.share_memory_() for k, v in shared_c.items()}              # EvenNICER_SLAM.py:92-93
    model.share_memory()                                                         # EvenNICER_SLAM.py:95
    before = _render(E, renderer, {k: v.clone() for k, v in shared_c.items()}, model, rays)
    ctx = mp.get_context('spawn')
    ev_first, ev_updated, q = ctx.Event(), ctx.Event(), ctx.Queue()
    pt = ctx.Process(target=_tracker, args=(shared_c, model, ev_first, ev_updated, q, declare))
    pm = ctx.Process(target=_mapper, args=(shared_c, model, ev_first, ev_updated))
    pt.start(); pm.start()
    out = q.get(timeout=600)
    for p in (pt, pm):
        p.join(timeout=120)
        assert p.exitcode == 0
    # this process sees the mapper child's writes in the same memory: a single-process render of the updated values
    import evennicer_slam_amd.functional as EF
    EF.clear_caches()
    after = _render(E, renderer, {k: v.clone() for k, v in shared_c.items()}, model, rays)
    assert not np.array_equal(before[0], after[0]) and not np.array_equal(before[2], after[2])     # the update changes the picture
    return out, before, after


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("layout", ['contiguous', 'channels_last_3d'])
def test_tracker_process_renders_what_a_mapper_process_wrote(layout):
    """The reference's flow (a fresh clone / deep copy per frame) and direct renders with external writers declared: both see
    the mapper process's in-place update, bit for bit the single-process render of the same values."""
    out, before, after = _run(layout, declare=True)
    assert _same(out['clone_before'], before) and _same(out['direct_before'], before)
    assert _same(out['clone_after'], after)
    assert _same(out['direct_after'], after)


def test_undeclared_direct_render_of_contiguous_shared_grids_is_the_documented_hazard():
    """Without the declaration the clone flow is still right (new tensors every frame: nothing to go stale), channels_last_3d
    grids would be too (read in place), but a DIRECT render of contiguous shared grids and shared decoders serves this process's
    cached copies, made before the other process wrote: the result equals the OLD picture.  INTEGRATION.md section 3c names
    `functional.external_writers(True)` / ENSLAM_EXTERNAL_WRITERS=1 for that case."""
    out, before, after = _run('contiguous', declare=False)
    assert _same(out['clone_before'], before) and _same(out['clone_after'], after)
    assert _same(out['direct_before'], before)
    assert _same(out['direct_after'], before), "the caches noticed another process's write: the hazard this test documents is gone -- update INTEGRATION.md"
