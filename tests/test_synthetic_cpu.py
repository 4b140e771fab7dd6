"""CPU tests of the pieces round 3 added around the path: the analytic view-consistent scene (synthetic.BoxRoom), the helper
that starts bench.py's ranks, the bench's host-core query, flag coarsening of the gradient bucket."""
import sys
import types

import numpy as np
import torch

from evennicer_slam_amd.synthetic import BoxRoom, look_at, trajectory


def _room():
    return BoxRoom([-1.0, -0.8, -0.6], [1.0, 0.8, 0.6], [0.2, -0.8, -0.3], [0.6, -0.3, 0.1], seed=3)


def test_box_room_depth_lies_on_surfaces_and_colour_follows_the_hit_point():
    room = _room()
    cam = dict(H=30, W=40, fx=35.0, fy=35.0, cx=19.5, cy=14.5)
    c2w = look_at([-0.5, 0.3, 0.1], [0.4, -0.5, -0.1])
    color, depth = room.render(c2w, cam)
    assert tuple(color.shape) == (30, 40, 3) and tuple(depth.shape) == (30, 40) and depth.dtype == torch.float32
    assert float(depth.min()) > 0.1 and float(color.min()) >= 0.05 - 1e-9 and float(color.max()) <= 0.95 + 1e-9
    j, i = torch.meshgrid(torch.arange(30, dtype=torch.float64), torch.arange(40, dtype=torch.float64), indexing='ij')
    dirs = torch.stack([(i - cam['cx']) / cam['fx'], -(j - cam['cy']) / cam['fy'], -torch.ones_like(i)], -1).reshape(-1, 3)
    pts = c2w[:3, 3] + (dirs @ c2w[:3, :3].T) * depth.double().reshape(-1, 1)
    # every hit point lies on a wall of the room or on a face of the box
    lo, hi, blo, bhi = room.room_lo, room.room_hi, room.box_lo, room.box_hi
    d_wall = torch.minimum((pts - lo).abs(), (pts - hi).abs()).min(dim=1).values
    inside_box = ((pts >= blo - 1e-5) & (pts <= bhi + 1e-5)).all(dim=1)
    d_box = torch.minimum((pts - blo).abs(), (pts - bhi).abs()).min(dim=1).values
    on_surface = (d_wall < 1e-5) | (inside_box & (d_box < 1e-5))
    assert bool(on_surface.all())
    assert bool(inside_box.any()) and not bool(inside_box.all())          # the box is in view, and so are walls
    assert torch.allclose(color.reshape(-1, 3), room.color_at(pts), atol=1e-6)


def test_box_room_is_view_consistent():
    """A surface point seen in view A re-projects into view B at the depth view B records there (where it is visible)."""
    room = _room()
    cam = dict(H=48, W=64, fx=55.0, fy=55.0, cx=31.5, cy=23.5)
    a, b = look_at([-0.5, 0.3, 0.1], [0.4, -0.5, -0.1]), look_at([-0.42, 0.34, 0.12], [0.45, -0.45, -0.1])
    _, da = room.render(a, cam)
    _, db = room.render(b, cam)
    j, i = torch.meshgrid(torch.arange(48, dtype=torch.float64), torch.arange(64, dtype=torch.float64), indexing='ij')
    dirs = torch.stack([(i - cam['cx']) / cam['fx'], -(j - cam['cy']) / cam['fy'], -torch.ones_like(i)], -1).reshape(-1, 3)
    pts = a[:3, 3] + (dirs @ a[:3, :3].T) * da.double().reshape(-1, 1)
    pc = (pts - b[:3, 3]) @ b[:3, :3]                                       # into camera B's frame
    z = -pc[:, 2]
    u, v = cam['fx'] * pc[:, 0] / z + cam['cx'], -cam['fy'] * pc[:, 1] / z + cam['cy']
    ui, vi = torch.round(u).long(), torch.round(v).long()
    ok = (z > 0.05) & (ui >= 0) & (ui < 64) & (vi >= 0) & (vi < 48) & ((u - ui).abs() < 0.05) & ((v - vi).abs() < 0.05)
    assert int(ok.sum()) > 20
    seen = db.double()[vi[ok], ui[ok]]
    same = (seen - z[ok]).abs() < 0.02                                      # equal unless another surface occludes it in B
    assert float(same.double().mean()) > 0.9
    assert bool((seen[~same] < z[ok][~same]).all())                         # a mismatch is always an occluder in FRONT


def test_trajectory_is_smooth_and_inside_the_room():
    room = _room()
    poses = trajectory(room, 20, step=0.01, yaw_deg=0.5)
    assert len(poses) == 20 and poses[0].dtype == torch.float32
    t = torch.stack([p[:3, 3] for p in poses]).double()
    assert bool(((t > room.room_lo) & (t < room.room_hi)).all())
    steps = (t[1:] - t[:-1]).norm(dim=1)
    assert float(steps.max()) < 0.0125 and float(steps.min()) > 0.0075
    for p in poses:
        R = p[:3, :3].double()
        assert torch.allclose(R @ R.T, torch.eye(3, dtype=torch.float64), atol=1e-5) and float(torch.det(R)) > 0.999


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE: one child `python -m torch.distributed.run --nproc-per-node N bench.py <args>`
    on the loopback interface, its exit code handed back; nothing else in this process touches a device."""
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '5', '--warmup', '2'])
    rc = bench.spawn_ranks(4)
    cmd = seen['cmd']
    assert rc == 7 and cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and 1024 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    assert cmd[-6:] == ['--gpus', '4', '--steps', '5', '--warmup', '2'] and cmd[-7].endswith('bench.py')
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'


def test_bench_physical_core_query():
    import bench
    cores, model = bench.physical_cores()
    assert isinstance(cores, int) and cores >= 1 and isinstance(model, str)


def test_bucket_flag_coarsening():
    from evennicer_slam_amd.parallel import _coarsen
    f = torch.tensor([0, 0, 1, 0, 0, 0, 0, 0, 0, 1], dtype=torch.uint8)      # 10 fine blocks -> 3 coarse (4 fine each), padded
    assert _coarsen(f, 4).tolist() == [1, 0, 1]
    assert _coarsen(torch.zeros(8, dtype=torch.uint8), 4).tolist() == [0, 0]
