// Ray gradients of the saved-activation backward: d(loss)/d(rays_o, rays_d) from the hand-off the decoder backward leaves per
// (tile, decoder slot) -- dC in register layout and the embedding's position gradient.  Shared by the stand-alone
// grid_bwd_kernel (render_bwd.hip) and the finish launch (step_kernel, util_kernels.hip), where it runs beside the
// gradient conversion it is independent of.
#pragma once
#include "common.hpp"

struct RayGradArgs {
    int n_rays, ntl, n_slots;
    const float* ro;
    const float* rd;
    const double* z;
    const float* dgrid_ws;       // [tile][ACT_SLOTS][DG_STRIDE]
    double lo[3], hi[3];         // Renderer.bound
    DevGrid grid[4];             // voxel-major values
    float* g_ro;
    float* g_rd;
    const int* work;             // optional work list of active tiles (null: every tile)
    const int* n_work;
};

struct TileGeo {
    double pw[3];
    float zf;
    int64_t sidx;
    int ray;
};

ENS_DEV TileGeo tile_geo(int64_t tile, int ntl, int S, const float* ro, const float* rd, const double* z, int p) {
    TileGeo g;
    g.ray = __builtin_amdgcn_readfirstlane((int)(tile / ntl));            // tile is wave-uniform
    const int tl = __builtin_amdgcn_readfirstlane((int)(tile - (int64_t)g.ray * ntl));
    g.sidx = (int64_t)g.ray * S + 16 * tl + p;
    const double zz = z[g.sidx];
    g.zf = (float)zz;
#pragma unroll
    for (int a = 0; a < 3; ++a) g.pw[a] = (double)ro[g.ray * 3 + a] + (double)rd[g.ray * 3 + a] * zz;
    return g;
}

// Coordinate gradient through the trilinear weights (ATen grid_sampler_3d_backward, gix/giy/giz) for the lane's
// 8 channels; partial over channels -> caller reduces over the 4 q lanes.
ENS_DEV void coord_grad_partial(const Vox& v, const DevGrid& g, int q, const f32x4& d0, const f32x4& d1, float& gx,
                                float& gy, float& gz) {
    gx = gy = gz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
        int x = v.ix + dx, y = v.iy + dy, z = v.iz + dz;
        const bool ok = (x < g.W) && (y < g.H) && (z < g.D);
        x = min(x, g.W - 1); y = min(y, g.H - 1); z = min(z, g.D - 1);
        const float* src = g.data + (((int64_t)z * g.H + y) * g.W + x) * 32 + 4 * q;
        const f32x4 a = ld4(src), b = ld4(src + 16);
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) dot = fmaf(a[r], d0[r], fmaf(b[r], d1[r], dot));
        dot = ok ? dot : 0.f;
        const float wx = dx ? v.fx : (1.f - v.fx), wy = dy ? v.fy : (1.f - v.fy), wz = dz ? v.fz : (1.f - v.fz);
        gx += (dx ? dot : -dot) * wy * wz;
        gy += (dy ? dot : -dot) * wx * wz;
        gz += (dz ? dot : -dot) * wx * wy;
    }
}

// reduce the per-sample position gradient over the tile and add it to the ray gradients
ENS_DEV void add_ray_grad(float dpx, float dpy, float dpz, float zf, int ray, float* g_ro, float* g_rd, int lane) {
    float v[6] = {dpx, dpy, dpz, dpx * zf, dpy * zf, dpz * zf};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicAdd(g_ro + ray * 3 + a, v[a]); atomicAdd(g_rd + ray * 3 + a, v[3 + a]); }
    }
}

// one wave: unit = tile * n_slots + slot
ENS_DEV void ray_grad_unit(const RayGradArgs& A, int64_t unit, int lane) {
    const int p = lane & 15, q = lane >> 4;
    int64_t tile = unit / A.n_slots;
    const int slot_idx = (int)(unit - tile * A.n_slots);
    if (A.work != nullptr) {                     // unit counts over the list of active tiles
        if (tile >= (int64_t)A.n_work[0]) return;
        tile = A.work[tile];
    }
    const DevGrid grid = A.grid[slot_idx + 1];
    const float* dgw = A.dgrid_ws + (tile * ACT_SLOTS + slot_idx) * DG_STRIDE;
    const f32x4 dc0 = ld4(dgw + lane * 4), dc1 = ld4(dgw + 256 + lane * 4);
    const f32x4 dpe = ld4(dgw + DG_DPE + lane * 4);
    // (the geometry's loads leave with the hand-off's, not behind the zero test: one memory round trip less in this latency chain)
    const int S = 16 * A.ntl;
    const TileGeo G = tile_geo(tile, A.ntl, S, A.ro, A.rd, A.z, p);
    bool nz = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) nz = nz || dc0[r] != 0.f || dc1[r] != 0.f || dpe[r] != 0.f;
    if (!__any(nz)) return;
    const Vox v = make_vox(G.pw, A.lo, A.hi, grid);
    float gx, gy, gz;
    coord_grad_partial(v, grid, q, dc0, dc1, gx, gy, gz);
    gx += __shfl_xor(gx, 16); gx += __shfl_xor(gx, 32);
    gy += __shfl_xor(gy, 16); gy += __shfl_xor(gy, 32);
    gz += __shfl_xor(gz, 16); gz += __shfl_xor(gz, 32);
    float dpx = gx * v.gx + dpe[0], dpy = gy * v.gy + dpe[1], dpz = gz * v.gz + dpe[2];
    if (q != 0) { dpx = dpy = dpz = 0.f; }
    add_ray_grad(dpx, dpy, dpz, G.zf, G.ray, A.g_ro, A.g_rd, lane);
}
