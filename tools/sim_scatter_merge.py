"""CPU simulation (no kernels): how many feature-gradient atomics the bench batch needs under different merge policies.
A cell's contribution = 4 x-pair row atomics (256 B each).  Policies: per ray run-length merge (today), plus merging across the
R neighbouring rays (rays ordered along a Morton curve of their pixels) that share a round of the backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import render_oracle as RO

n_rays = int(os.environ.get('RAYS', 1000))
sc = bench.build_scene_cpu('room0', 0)
ro, rd, gd, gc = bench.make_rays(sc, n_rays, 1000)
z = RO.sample_depths(ro, rd, gd, sc['bound'], 32, 16, 'color')
pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
# active tiles: the oracle's forward weights
params = {k: v for k, v in sc['model'].state_dict().items()}
with torch.no_grad():
    raw = RO.eval_points(params, sc['grids'], pts.reshape(-1, 3).float(), 'color', sc['bound']).reshape(n_rays, 48, 4)
    depth, var, color, w = RO.composite(raw, z)
act = (w.reshape(n_rays, 3, 16) != 0).any(-1).numpy()          # [ray, tile]
print("active tile fraction (weights != 0):", act.mean())

def morton(u, v):
    def spread(x):
        x = x.astype(np.uint64)
        x = (x | (x << 8)) & 0x00FF00FF; x = (x | (x << 4)) & 0x0F0F0F0F; x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555
        return x
    return spread(u) | (spread(v) << 1)

# pixel of each ray from its camera-space direction
c2w = sc['c2w'].double().numpy(); cam = sc.get('cam', bench.CAM)
dc = (rd.double().numpy() @ c2w[:3, :3])          # camera-space direction (R^T d)
u = dc[:, 0] / -dc[:, 2] * cam['fx'] + cam['cx']; v = dc[:, 1] / dc[:, 2] * cam['fy'] + cam['cy']
print("pixel range", u.min(), u.max(), v.min(), v.max())
order = np.argsort(morton(np.clip(u, 0, 4095).astype(np.int64), np.clip(v, 0, 4095).astype(np.int64)), kind='stable')

for key in ('grid_middle', 'grid_fine'):
    shape = tuple(sc['grids'][key].shape[2:])
    (ix, iy, iz), _fr, _in = RO.voxel_coords(pts.reshape(-1, 3).float(), sc["bound"], shape)
    cell = ((iz.long() * shape[1] + iy.long()) * shape[2] + ix.long()).reshape(n_rays, 3, 16).numpy()
    def count(order, R, piece):
        """rounds of R rays (same tile index k, consecutive in `order`); within a round the unit of merging is `piece` samples per ray
        (16: the whole tiles of the round merged together; 4: quarter tiles as the dW waves deal them)"""
        total = 0
        for k in range(3):
            rays = [r for r in order if act[r, k]]
            for i in range(0, len(rays), R):
                grp = rays[i:i + R]
                for p0 in range(0, 16, piece):
                    total += len(np.unique(cell[grp, k, p0:p0 + piece]))
        return total
    unmerged = int(act.sum()) * 16
    ident = np.arange(n_rays)
    print(f"{key} {shape}: samples in active tiles {unmerged}; distinct cells overall {len(np.unique(cell[act]))}")
    print(f"   per ray, whole tile (today)         {count(ident, 1, 16)}")
    for R in (4, 8, 16):
        print(f"   R={R:2d} random order  tile {count(ident, R, 16)}  quarter {count(ident, R, 4)}    Morton order  tile {count(order, R, 16)}  quarter {count(order, R, 4)}")

print("\n-- deferred scatter: one workgroup per G Morton-consecutive rays and grid, distinct CORNER ROWS (128 B each) per workgroup")
for key in ('grid_middle', 'grid_fine'):
    shape = tuple(sc['grids'][key].shape[2:])
    (ix, iy, iz), _fr, _in = RO.voxel_coords(pts.reshape(-1, 3).float(), sc["bound"], shape)
    ix = ix.reshape(n_rays, 3, 16).numpy(); iy = iy.reshape(n_rays, 3, 16).numpy(); iz = iz.reshape(n_rays, 3, 16).numpy()
    D, H, W = shape
    def rows_of(sel):
        out = []
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    out.append(((np.minimum(iz[sel] + dz, D - 1)) * H + np.minimum(iy[sel] + dy, H - 1)) * W + np.minimum(ix[sel] + dx, W - 1))
        return np.concatenate([o.reshape(-1) for o in out])
    allrows = len(np.unique(rows_of(act)))
    for G in (8, 16, 32, 64):
        for name, od in (("random", np.arange(n_rays)), ("Morton", order)):
            tot = 0; mx = 0
            for i in range(0, n_rays, G):
                grp = od[i:i + G]
                m = np.zeros_like(act); m[grp] = act[grp]
                n = len(np.unique(rows_of(m))); tot += n; mx = max(mx, n)
            print(f"   {key} G={G:2d} {name:6s}: rows {tot:6d} (max per workgroup {mx}) = {tot * 128 / 1e6:.2f} MB of atomics; overall distinct rows {allrows}")

print("\n-- one wave = sample index s of R Morton-consecutive rays (no table): distinct CELLS per wave-unit, 4 atomics each")
for key in ('grid_middle', 'grid_fine'):
    shape = tuple(sc['grids'][key].shape[2:])
    (ix, iy, iz), _fr, _in = RO.voxel_coords(pts.reshape(-1, 3).float(), sc["bound"], shape)
    cell = ((iz.long() * shape[1] + iy.long()) * shape[2] + ix.long()).reshape(n_rays, 48).numpy()
    acts = np.repeat(act, 16, axis=1)          # [ray, 48]
    for R in (16, 32, 64):
        for name, od in (("random", np.arange(n_rays)), ("Morton", order)):
            tot = 0
            for i in range(0, n_rays, R):
                grp = od[i:i + R]
                for s_ in range(48):
                    m = acts[grp, s_]
                    if m.any():
                        tot += len(np.unique(cell[grp, s_][m]))
            print(f"   {key} R={R:2d} {name:6s}: cells {tot:6d} -> {tot * 4} atomics = {tot * 4 * 256 / 1e6:.1f} MB")
