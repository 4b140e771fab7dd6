// Fused forward of the rendering hot path for gfx950: one 64-lane wave renders one ray
// (S = 16*NTL samples): sample points (float64) -> bound mask -> trilinear gather from voxel-major
// grids -> decoders as chained v_mfma_f32_16x16x4_f32 with activations kept in registers
// -> alpha compositing with wave shuffles.
//
// Replaces (reference paths): src/utils/Renderer.py:173-181 (points, eval_points call),
// Renderer.py:24-62 (bound mask), src/conv_onet/models/decoder.py:168-203,254-274,312-342
// (grid_sample + MLPs + stage combine), src/common.py:256-297 (raw2outputs, occupancy branch).
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"
#include "lds_util.hpp"
#include "stamps.hpp"

#define IC(n) std::integral_constant<int, n>{}

namespace {

ENS_DEV unsigned fwd_pos_bits(const f32x4& v) {
    return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
}
// Write one register tile to the workspace in the backward's deposit layout: transpose through a wave-private
// 1 KB LDS tile, then one coalesced 16-byte store per lane.
ENS_DEV void ws_store_dep(float* __restrict__ ws_tile, const f32x4& x, float* stage, int lane, int p, int q) {
#ifdef ENS_EXP_MFMA_TRANSPOSE
    // A/B aid: the transposition as an MFMA against the identity -- D[sample][feature] = sum_k X[sample][k] I[k][feature]
    // with the register tile as the A operand (step r, k-slot q carries feature 4q+r) and the matching rows of I as B.
    // Exact (products with 1 and 0); 4 MFMAs per tile instead of an LDS round trip with two waits.
    f32x4 t = splat4(0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) t = MFMA16(x[r], (p == 4 * q + r) ? 1.f : 0.f, t);
    *reinterpret_cast<f32x4*>(ws_tile + (lane ^ (lane >> 4)) * 4) = t;          // (swizzled tile, as below)
#else
    // (storing component-major [r][lane] with plain dword stores and turning the tile in the backward's
    // global_load_lds by addressing was tried: forward -2.6 us, backward +10 us -- the strided 16-byte source runs cost
    // more than the LDS round trip here, which other waves hide.  Four scattered dword stores per tile straight into
    // the deposit layout, without the LDS round trip: no gain either, 0.346 vs 0.339 ms per step)
    // The 16-byte chunk (4q + r) of sample group P = p >> 2 sits at chunk (4q + r) ^ P of its 64-float row: written plainly,
    // the 32 lanes of a ds_write_b32 group would land on 8 banks (4-way); with the XOR they cover all 32, and the
    // plain ds_read_b128 below hands lane L the fragment of lane L ^ (L >> 4): the workspace tile is stored SWIZZLED, which is
    // how the backward's slots want it (render_bwd.hip, dep_bases: global_load_lds copies the tile verbatim).
    const int P = p >> 2;
    float* d = stage + P * 64 + (p & 3) + 16 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[4 * (r ^ P)] = x[r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + lane * 4);     // the tile goes to the workspace swizzled, as the backward reads it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef ENS_EXP_NT_WS          // A/B aid: the workspace written with non-temporal stores
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ws_tile + lane * 4));
#else
    *reinterpret_cast<f32x4*>(ws_tile + lane * 4) = v;
#endif
#endif
}

ENS_DEV f32x4 sin4(f32x4 v) { return f32x4{ens_sinf(v[0]), ens_sinf(v[1]), ens_sinf(v[2]), ens_sinf(v[3])}; }

// One block of MLP.forward (decoder.py:193-199): h = relu(W_i x + b_i) + (Wc_i c + bc_i).
template <int I, int CT, int NTL>
ENS_DEV void xyz_layer(const float* __restrict__ pk, const f32x4 (&emb)[NTL][6], const f32x4 (&c)[NTL][CT],
                       f32x4 (&h)[NTL][2], unsigned (&mb)[NTL][2], float* const (&ws)[NTL], bool wl, float* stage, int lane,
                       int p, int q) {
    constexpr XyzLay L{CT * 16};
    f32x4 acc[NTL][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const f32x4 b = ld4(pk + L.ob(I) + 16 * rt + 4 * q);
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) acc[tl][rt] = b;
    }
    if constexpr (I == 0) {
        linear32<6, NTL, 6, true>(acc, pk + L.oW(0), 96, emb, 0, p, q);
    } else if constexpr (I == 3) {
        linear32<6, NTL, 6, true>(acc, pk + L.oW(3), 128, emb, 0, p, q);
        linear32<2, NTL, 2, true>(acc, pk + L.oW(3) + 6 * 256, 128, h, 0, p, q);
    } else {
        linear32<2, NTL, 2, true>(acc, pk + L.oW(I), 32, h, 0, p, q);
    }
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
        const unsigned bits = fwd_pos_bits(acc[tl][0]) | (fwd_pos_bits(acc[tl][1]) << 4);
        if constexpr (I < 4) mb[tl][0] |= bits << (8 * I); else mb[tl][1] = bits;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const f32x4 bc = ld4(pk + L.obc(I) + 16 * rt + 4 * q);
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) acc[tl][rt] = relu4(acc[tl][rt]) + bc;
    }
    linear32<CT, NTL, CT, true>(acc, pk + L.oWc(I), CT * 16, c, 0, p, q);
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
        h[tl][0] = acc[tl][0];
        h[tl][1] = acc[tl][1];
        if (ws[tl] != nullptr) {                    // backward operands: slot order EMB 0..5 | h2 6 | h0 8 | h1 10 | h3 12
            constexpr int T = I == 2 ? 6 : (I == 0 ? 8 : (I == 1 ? 10 : 12));
            if constexpr (I < 4) {
                if (!wl) {
                    ws_store_dep(ws[tl] + T * 256, h[tl][0], stage, lane, p, q);
                    ws_store_dep(ws[tl] + (T + 1) * 256, h[tl][1], stage, lane, p, q);
                }
            } else {                                // h4 feeds the VALU dWo: register layout
                if (!wl) {
                    *reinterpret_cast<f32x4*>(ws[tl] + ACT_H4 + lane * 4) = h[tl][0];
                    *reinterpret_cast<f32x4*>(ws[tl] + ACT_H4 + 256 + lane * 4) = h[tl][1];
                }
                *reinterpret_cast<uint2*>(ws[tl] + (wl ? ACTL_MASK : ACT_MASK) + lane * 2) = make_uint2(mb[tl][0], mb[tl][1]);
            }
        }
    }
}

// MLP (middle/fine/color), decoder.py:177-203.  c: CT feature tiles; o: output tile (rows 0..n_out-1
// valid on lanes q == 0).
// ws[tl]: activation-workspace block of (tile tl, this decoder) or nullptr (nothing saved)
template <int CT, int NTL>
ENS_DEV void mlp_xyz_fwd(const float* __restrict__ pk, const float (&pc)[NTL], const f32x4 (&c)[NTL][CT],
                         f32x4 (&o)[NTL], float* const (&ws)[NTL], bool wl, float* stage, int lane, int p, int q) {
    constexpr XyzLay L{CT * 16};
    f32x4 emb[NTL][6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const float a = pk[L.oBT() + (16 * t + p) * 4 + q];
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) emb[tl][t] = sin4(MFMA16(a, pc[tl], splat4(0.f)));
    }
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
        if (ws[tl] != nullptr) {
            if (!wl) {
#pragma unroll
                for (int t = 0; t < 6; ++t) ws_store_dep(ws[tl] + t * 256, emb[tl][t], stage, lane, p, q);
#pragma unroll
                for (int t = 0; t < CT; ++t) ws_store_dep(ws[tl] + (14 + t) * 256, c[tl][t], stage, lane, p, q);
            }
            // sample coordinates as a feature tile (features 0..2 = x,y,z live on q == 0 lanes): dB^T operand
            const float cx = __shfl(pc[tl], p), cy = __shfl(pc[tl], 16 + p), cz = __shfl(pc[tl], 32 + p);
            ws_store_dep(ws[tl] + (wl ? ACTL_Q : (14 + CT) * 256), q == 0 ? f32x4{cx, cy, cz, 0.f} : splat4(0.f), stage, lane, p, q);
        }
    }
    f32x4 h[NTL][2];
    unsigned mb[NTL][2];
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) mb[tl][0] = mb[tl][1] = 0u;
    xyz_layer<0, CT, NTL>(pk, emb, c, h, mb, ws, wl, stage, lane, p, q);
    xyz_layer<1, CT, NTL>(pk, emb, c, h, mb, ws, wl, stage, lane, p, q);
    xyz_layer<2, CT, NTL>(pk, emb, c, h, mb, ws, wl, stage, lane, p, q);
    xyz_layer<3, CT, NTL>(pk, emb, c, h, mb, ws, wl, stage, lane, p, q);
    xyz_layer<4, CT, NTL>(pk, emb, c, h, mb, ws, wl, stage, lane, p, q);
    out_layer<NTL>(o, pk + L.oWo(), pk + L.obo(), h, p, q);
}

template <int I, int NTL>
ENS_DEV void feat_layer(const float* __restrict__ pk, const f32x4 (&c)[NTL][2], f32x4 (&h)[NTL][2], int p, int q) {
    constexpr FeatLay L{};
    f32x4 acc[NTL][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const f32x4 b = ld4(pk + L.ob(I) + 16 * rt + 4 * q);
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) acc[tl][rt] = b;
    }
    if constexpr (I == 0) {
        linear32<2, NTL, 2>(acc, pk + L.oW(0), 32, c, 0, p, q);
    } else if constexpr (I == 3) {
        linear32<2, NTL, 2>(acc, pk + L.oW(3), 64, c, 0, p, q);
        linear32<2, NTL, 2>(acc, pk + L.oW(3) + 32, 64, h, 0, p, q);
    } else {
        linear32<2, NTL, 2>(acc, pk + L.oW(I), 32, h, 0, p, q);
    }
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) { h[tl][0] = relu4(acc[tl][0]); h[tl][1] = relu4(acc[tl][1]); }
}

// MLP_no_xyz (coarse), decoder.py:262-274.
template <int NTL>
ENS_DEV void mlp_feat_fwd(const float* __restrict__ pk, const f32x4 (&c)[NTL][2], f32x4 (&o)[NTL], int p, int q) {
    constexpr FeatLay L{};
    f32x4 h[NTL][2];
    feat_layer<0, NTL>(pk, c, h, p, q);
    feat_layer<1, NTL>(pk, c, h, p, q);
    feat_layer<2, NTL>(pk, c, h, p, q);
    feat_layer<3, NTL>(pk, c, h, p, q);
    feat_layer<4, NTL>(pk, c, h, p, q);
    out_layer<NTL>(o, pk + L.oWo(), pk + L.obo(), h, p, q);
}

// value held by lane (p, q=0) for tile tl  ->  lane 16*tl + p
template <int NTL>
ENS_DEV float to_sample_lane(const f32x4 (&o)[NTL], int comp, int lane) {
    float r = 0.f;
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
        const float v = __shfl(o[tl][comp], lane & 15);
        r = ((lane >> 4) == tl) ? v : r;
    }
    return r;
}

// MODE 0: one wave per ray (S = 16*NTL samples), compositing fused.
// MODE 1: explicit points (eval_points), 16*NTL points per wave, raw only.
// MODE 2: one wave per 16-sample tile of a ray (NTL == 1, tiles_per_ray > 0): raw only; compositing runs as its
//         own kernel.  3x the waves of MODE 0 at less than half the registers: latency is hidden by occupancy.
template <int STAGE, int NTL>
__global__ __launch_bounds__(64, NTL == 1 ? 3 : 1) void render_fwd_kernel(int64_t n_units, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d,
                                                        const double* __restrict__ z_vals,
                                                        const double* __restrict__ points, int64_t n_points,
                                                        int apply_mask, int tiles_per_ray, DevScene sc, double* __restrict__ depth,
                                                        double* __restrict__ var, float* __restrict__ rgb,
                                                        float* __restrict__ raw_out, float* __restrict__ act_ws, int wl) {
    __shared__ __attribute__((aligned(16))) float ws_stage[256];
    const int WSS = wl ? ACTL_STRIDE : ACT_STRIDE, WSV = wl ? ACTL_VOX : ACT_VOX;
    constexpr int S = 16 * NTL;
    const int lane = threadIdx.x, p = lane & 15, q = lane >> 4;
    const bool tile_mode = tiles_per_ray > 0;
    const int64_t unit = blockIdx.x;
    const bool ray_mode = (points == nullptr) && !tile_mode;
    const int64_t ray = tile_mode ? unit / tiles_per_ray : unit;           // wave-uniform

    double pw[NTL][3];
    float pc[NTL];
    if (points == nullptr) {
        double o[3], d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { o[a] = (double)rays_o[ray * 3 + a]; d[a] = (double)rays_d[ray * 3 + a]; }
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) {
            const double z = z_vals[unit * S + 16 * tl + p];
#pragma unroll
            for (int a = 0; a < 3; ++a) pw[tl][a] = o[a] + d[a] * z;     // Renderer.py:173-174 (float64)
        }
    } else {
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) {
            int64_t i = unit * S + 16 * tl + p;
            i = i < n_points ? i : n_points - 1;
#pragma unroll
            for (int a = 0; a < 3; ++a) pw[tl][a] = points[i * 3 + a];
        }
    }
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
        pc[tl] = q == 0 ? (float)pw[tl][0] : (q == 1 ? (float)pw[tl][1] : (q == 2 ? (float)pw[tl][2] : 0.f));
    }

    f32x4 occ[NTL], col[NTL];
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) { occ[tl] = splat4(0.f); col[tl] = splat4(0.f); }
    // activation workspace blocks of this wave's tiles (tile index = unit*NTL + tl), one per decoder slot
    float* ws0[NTL];
    float* ws1[NTL];
    float* ws2[NTL];
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
        float* b = (act_ws != nullptr && points == nullptr) ? act_ws + ((unit * NTL + tl) * ACT_SLOTS) * (int64_t)WSS : nullptr;
        ws0[tl] = b;
        ws1[tl] = b ? b + WSS : nullptr;
        ws2[tl] = b ? b + 2 * WSS : nullptr;
    }

    if constexpr (STAGE == 0) {
        f32x4 c[NTL][2];
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) {
            const Vox v = make_vox(pw[tl], sc.clo, sc.chi, sc.grid[0]);
            gather8(v, sc.grid[0], q, c[tl][0], c[tl][1]);
        }
        mlp_feat_fwd<NTL>(sc.packed[0], c, occ, p, q);
    } else {
        f32x4 cm[NTL][2];
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) {
            const Vox v = make_vox(pw[tl], sc.lo, sc.hi, sc.grid[1]);
            gather8(v, sc.grid[1], q, cm[tl][0], cm[tl][1]);
            if (ws0[tl] != nullptr && q == 0) *reinterpret_cast<f32x4*>(ws0[tl] + WSV + p * 4) = vox_record(v, sc.grid[1]);
        }
        mlp_xyz_fwd<2, NTL>(sc.packed[1], pc, cm, occ, ws0, wl != 0, ws_stage, lane, p, q);
        if constexpr (STAGE >= 2) {
            f32x4 cf[NTL][4];
#pragma unroll
            for (int tl = 0; tl < NTL; ++tl) {
                const Vox v = make_vox(pw[tl], sc.lo, sc.hi, sc.grid[2]);
                gather8(v, sc.grid[2], q, cf[tl][0], cf[tl][1]);
                if (ws1[tl] != nullptr && q == 0) *reinterpret_cast<f32x4*>(ws1[tl] + WSV + p * 4) = vox_record(v, sc.grid[2]);
                cf[tl][2] = cm[tl][0];                                    // decoder.py:184-187 concat
                cf[tl][3] = cm[tl][1];
            }
            f32x4 of[NTL];
            mlp_xyz_fwd<4, NTL>(sc.packed[2], pc, cf, of, ws1, wl != 0, ws_stage, lane, p, q);
#pragma unroll
            for (int tl = 0; tl < NTL; ++tl) occ[tl][0] = of[tl][0] + occ[tl][0];   // fine_occ + middle_occ
        }
        if constexpr (STAGE == 3) {
            f32x4 cc[NTL][2];
#pragma unroll
            for (int tl = 0; tl < NTL; ++tl) {
                const Vox v = make_vox(pw[tl], sc.lo, sc.hi, sc.grid[3]);
                gather8(v, sc.grid[3], q, cc[tl][0], cc[tl][1]);
                if (ws2[tl] != nullptr && q == 0) *reinterpret_cast<f32x4*>(ws2[tl] + WSV + p * 4) = vox_record(v, sc.grid[3]);
            }
            mlp_xyz_fwd<2, NTL>(sc.packed[3], pc, cc, col, ws2, wl != 0, ws_stage, lane, p, q);
        }
    }

    // ---- per-sample values on lane k = 16*tile + point
    const bool valid = lane < S;
    float occ_k = to_sample_lane<NTL>(occ, 0, lane);
    const float r_k = to_sample_lane<NTL>(col, 0, lane);
    const float g_k = to_sample_lane<NTL>(col, 1, lane);
    const float b_k = to_sample_lane<NTL>(col, 2, lane);
    const int64_t sidx = unit * S + lane;
    double zk = 0.0;
    {   // strict in-bound test of this lane's own sample (Renderer.py:44-47); outside -> occ = 100 (:58)
        double pk3[3] = {0.0, 0.0, 0.0};
        if (points == nullptr) {
            zk = valid ? z_vals[sidx] : 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) pk3[a] = (double)rays_o[ray * 3 + a] + (double)rays_d[ray * 3 + a] * zk;
        } else {
            const int64_t i = sidx < n_points ? sidx : n_points - 1;
#pragma unroll
            for (int a = 0; a < 3; ++a) pk3[a] = points[i * 3 + a];
        }
        bool in = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) in = in && (pk3[a] < sc.hi[a]) && (pk3[a] > sc.lo[a]);
        if (!in && apply_mask) occ_k = 100.f;
    }
    if (raw_out != nullptr && valid && (points == nullptr || sidx < n_points))
        *reinterpret_cast<f32x4*>(raw_out + sidx * 4) = f32x4{r_k, g_k, b_k, occ_k};
    if (!ray_mode) return;

    // ---- raw2outputs_nerf_color, occupancy branch (common.py:284-296)
    const float alpha = valid ? 1.f / (1.f + expf(-(10.f * occ_k))) : 0.f;
    const float m = valid ? (1.f - alpha) + 1e-10f : 1.f;
    float incl = m;                                                        // inclusive prefix product
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off);
        incl = lane >= off ? incl * t : incl;
    }
    float T = __shfl_up(incl, 1);
    T = lane == 0 ? 1.f : T;
    const float w = alpha * T;
    const float cr = wave_sum(w * r_k), cg = wave_sum(w * g_k), cb = wave_sum(w * b_k);
    const double dep = wave_sum((double)w * zk);
    const double tmp = zk - dep;
    const double vr = wave_sum(((double)w * tmp) * tmp);
    if (lane == 0) {
        depth[unit] = dep;
        var[unit] = vr;
        rgb[unit * 3 + 0] = cr;
        rgb[unit * 3 + 1] = cg;
        rgb[unit * 3 + 2] = cb;
    }
}

// ------------------------------------------------------------------------------------------------
// Tile-mode forward with a shared LDS weight ring: one workgroup = 4 waves = 4 tiles of 16 samples, in lockstep.
// Each layer's chunk (W|b|Wc|bc of the packed decoder) is streamed into a 2-slot LDS ring by async global_load_lds
// one layer ahead and read by all 4 waves with  base-VGPR + immediate  ds_read_b128; this replaces 3000 per-wave
// passes over the 210 KB of weights through L2 by 750 per-workgroup passes and takes the L2 latency off every layer.
// ------------------------------------------------------------------------------------------------
constexpr int fwd_ring_floats(int stage) { return stage >= 2 ? 32 * 128 + 32 + 32 * 64 + 32 : 32 * 128 + 32 + 32 * 32 + 32; }
constexpr int fwd_ring_lds_bytes(int stage) { return (2 * fwd_ring_floats(stage) + 4 * 256) * 4; }

// decoder `kind` through the ring; its layer i is global chunk C0+i (buffer (C0+i)&1).  pk_next: packed decoder whose
// layer 0 follows in the ring (nullptr: none).  NEXT_CD: its fc_c width.
// RES >= 0: the decoder's whole forward section ([B^T | chunk 0 .. chunk 4 | Wo | bo], the packed layout verbatim) is RESIDENT in
// LDS at byte offset RES -- no ring traffic and no barrier anywhere in the call (render_fwd_res_kernel: waves run independently).
template <int CT, int C0, int RB, int NEXT_CD, int RES = -1>
ENS_DEV void mlp_xyz_ring(const float* __restrict__ pk, const float* __restrict__ pk_next, float* ring, float pc,
                          const f32x4 (&c)[CT], f32x4& o, float* ws, bool wl, float* stage, unsigned wt, unsigned wq, int wave, int lane, int p, int q,
                          StampCtx& sx) {
    constexpr XyzLay L{CT * 16};
    constexpr int CD = CT * 16;
    f32x4 emb[6];
    // B^T [96][4] sits right in front of layer 0's chunk in the packed decoder and rides in with it (ring slot of chunk C0:
    // [B^T | W0 | b0 | Wc0 | bc0]): six LDS reads instead of six global loads in front of the embedding MFMAs
    constexpr int RO0 = RES >= 0 ? RES : ((C0 & 1) ? RB * 4 : 0);
    const unsigned bt = wq - (unsigned)q * 16u + RO0 + (unsigned)(p * 4 + q) * 4u;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const float a = *reinterpret_cast<const lds_float*>(static_cast<uintptr_t>(bt + 16 * t * 16));
        emb[t] = sin4(MFMA16(a, pc, splat4(0.f)));
    }
    FST(sx, 1)      // embedding: B^T reads, MFMA, sin
    if (ws != nullptr) {
        if (!wl) {
#pragma unroll
            for (int t = 0; t < 6; ++t) ws_store_dep(ws + t * 256, emb[t], stage, lane, p, q);
#pragma unroll
            for (int t = 0; t < CT; ++t) ws_store_dep(ws + (14 + t) * 256, c[t], stage, lane, p, q);
        }
        const float cx = __shfl(pc, p), cy = __shfl(pc, 16 + p), cz = __shfl(pc, 32 + p);
        ws_store_dep(ws + (wl ? ACTL_Q : (14 + CT) * 256), q == 0 ? f32x4{cx, cy, cz, 0.f} : splat4(0.f), stage, lane, p, q);
    }
    FST(sx, 2)      // workspace stores of emb / c / coordinates
    f32x4 h[5][2];
    f32x4 wo_a0 = splat4(0.f), wo_a1 = splat4(0.f), wo_b = splat4(0.f);
    unsigned mb0 = 0u, mb1 = 0u;
    auto layer = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int RO = RES >= 0 ? RES + 4 * L.oW(i)
                                    : (((C0 + i) & 1) ? RB * 4 : 0) + (i == 0 ? 384 * 4 : 0);      // (layer 0: behind B^T)
        constexpr int K = L.K(i);
        constexpr int OB = RO + 32 * K * 4, OC = OB + 32 * 4, OBC = OC + 32 * CD * 4;
        if constexpr (RES < 0) {
            float* nxt = ring + (((C0 + i + 1) & 1) ? RB : 0);
            if constexpr (i < 4) {
                // (layer 4's chunk brings the output layer along: Wo [16][32] | bo [16] follow it in the packed decoder)
                ring_load(nxt, pk + L.oW(i + 1), (L.oW(i + 2) - L.oW(i + 1) + (i == 3 ? 528 : 0)) / 4, wave, lane);
            } else if constexpr (NEXT_CD > 0) {
                constexpr XyzLay LN{NEXT_CD};
                ring_load(nxt, pk_next, LN.oW(1) / 4, wave, lane);                   // B^T | chunk 0 of the next decoder
            }
        }
        f32x4 acc[2];
        acc[0] = lds4(wq + OB);
        acc[1] = lds4(wq + OB + 64);
        if constexpr (i == 0) {
            lin_lds_tm<2, 6, 96, RO>(acc, wt, emb);
        } else if constexpr (i == 3) {
            lin_lds_tm<2, 6, 128, RO>(acc, wt, emb);
            lin_lds_tm<2, 2, 128, RO + 6 * 256 * 4>(acc, wt, h[2]);
        } else {
            lin_lds_tm<2, 2, 32, RO>(acc, wt, h[i - 1]);
        }
        if (ws != nullptr) {                 // ReLU masks for the backward (forward-only calls skip the 8 % of vector work)
            const unsigned bits = fwd_pos_bits(acc[0]) | (fwd_pos_bits(acc[1]) << 4);
            if constexpr (i < 4) mb0 |= bits << (8 * i); else mb1 = bits;
        }
        acc[0] = relu4(acc[0]) + lds4(wq + OBC);
        acc[1] = relu4(acc[1]) + lds4(wq + OBC + 64);
        lin_lds_tm<2, CT, CD, OC>(acc, wt, c);
        h[i][0] = acc[0];
        h[i][1] = acc[1];
        if constexpr (i == 4) {             // output-layer fragments out of the same ring slot, before the barrier that frees it
            constexpr int OWO = RO + (L.oW(5) - L.oW(4)) * 4;
            const unsigned wr = (unsigned)(p * 32 + 4 * q) * 4u;
            wo_a0 = lds4(wq - (unsigned)q * 16u + wr + OWO);
            wo_a1 = lds4(wq - (unsigned)q * 16u + wr + OWO + 64);
            wo_b = lds4(wq + OWO + 512 * 4);
        }
        FST(sx, 3)  // layer: prefetch issue, bias, fragment reads + MFMAs, relu
        if (ws != nullptr) {
            constexpr int T = i == 2 ? 6 : (i == 0 ? 8 : (i == 1 ? 10 : 12));
            if constexpr (i < 4) {
                if (!wl) {
                    ws_store_dep(ws + T * 256, h[i][0], stage, lane, p, q);
                    ws_store_dep(ws + (T + 1) * 256, h[i][1], stage, lane, p, q);
                }
            } else {
                if (!wl) {
                    *reinterpret_cast<f32x4*>(ws + ACT_H4 + lane * 4) = h[4][0];
                    *reinterpret_cast<f32x4*>(ws + ACT_H4 + 256 + lane * 4) = h[4][1];
                }
                *reinterpret_cast<uint2*>(ws + (wl ? ACTL_MASK : ACT_MASK) + lane * 2) = make_uint2(mb0, mb1);
            }
        }
        FST(sx, 4)  // workspace stores of h
        if constexpr (RES >= 0) return;             // resident weights: nothing to wait for, nobody to wait for
#ifndef ENS_FULL_VMCNT
        // The layer barrier waits for the ring chunk only, not for the workspace stores issued behind it: vmcnt counts
        // loads, stores and LDS-DMA together in issue order (MI355X_MICROARCH.md), so leaving the N youngest operations --
        // the 2 h tiles of this layer; layer 4: 2 h4 tiles + the mask words -- in flight still guarantees that the older
        // chunk request has landed.  The order is pinned: the request sits in front of the first sched_barrier(0) of
        // lin_lds_tm, the stores behind the last one.  Spill traffic would only add younger operations (a longer wait).
        // Forward-only calls and the light workspace store other things: full wait.  -DENS_FULL_VMCNT restores
        // __syncthreads() everywhere (75.9 / 76.2 / 76.0 against 75.0 / 74.8 / 75.4 us).
        if (ws != nullptr && !wl) {
            if constexpr (i < 4) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            __syncthreads();
        }
#else
        __syncthreads();                // next chunk landed; all waves done with this buffer
#endif
        FST(sx, 5)  // barrier (+ wait for the next chunk)
    };
    layer(IC(0)); layer(IC(1)); layer(IC(2)); layer(IC(3)); layer(IC(4));
    // output layer (tiny): its weights rode in layer 4's ring chunk (from global they cost 2 k cycles of load latency per decoder)
    o = wo_b;
#pragma unroll
    for (int r = 0; r < 4; ++r) o = MFMA16(wo_a0[r], h[4][0][r], o);
#pragma unroll
    for (int r = 0; r < 4; ++r) o = MFMA16(wo_a1[r], h[4][1][r], o);
    FST(sx, 6)      // output layer
}

// In the colour stage the launch is split into two ROLES of workgroups: the occupancy decoders (middle + fine, 588 MFMAs
// per tile) and the colour decoder (254) run in different workgroups on the same 4 tiles.  The three workgroups a CU
// holds then differ in length and drift out of phase, so one workgroup's VALU-bound stretches (embedding sin, gathers,
// workspace stores) run under another's MFMA stretches instead of all three hitting the same unit at the same time; the
// two roles write disjoint bytes of raw (w | x,y,z).  role = bit 3 of blockIdx (blocks are dealt round-robin over the 8
// XCDs: a role chosen by blockIdx & 1 would load the XCDs 2.3 : 1).
// (Measured, round 2, same box: split 109 us vs 87 us unsplit.  The de-phasing itself raised the per-workgroup rate by ~11 %,
// but 1500 workgroups on 768 resident slots run in two staggered rounds whose tail costs more.  Kept behind
// -DENS_EXP_FWD_SPLIT for A/B runs; the shipped kernel runs all three decoders of a tile group in one workgroup.)
#ifdef ENS_EXP_FWD_SPLIT
constexpr bool fwd_split_roles(int stage) { return stage == 3; }
#else
constexpr bool fwd_split_roles(int) { return false; }
#endif

#ifndef ENS_FWD_STAGGER
#define ENS_FWD_STAGGER 0          // units of 2048 cycles (A/B aid, see the kernel)
#endif
template <int STAGE>
__global__ __launch_bounds__(256, 3) void render_fwd_ring_kernel(int64_t n_tiles, int tiles_per_ray,
                                                                 const float* __restrict__ rays_o,
                                                                 const float* __restrict__ rays_d,
                                                                 const double* __restrict__ z_vals, DevScene sc,
                                                                 float* __restrict__ raw_out, float* __restrict__ act_ws, int wli, int stag_cus,
                                                                 const double* __restrict__ points, int64_t n_points, int apply_mask) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    constexpr bool SPLIT = fwd_split_roles(STAGE);
    const int role = SPLIT ? (int)((blockIdx.x >> 3) & 1) : 0;               // 0: occupancy decoders, 1: colour decoder
    const int64_t grp = SPLIT ? (int64_t)(blockIdx.x >> 4) * 8 + (blockIdx.x & 7) : (int64_t)blockIdx.x;
    if (grp * 4 >= n_tiles) return;                                            // (whole workgroup: padding of the split grid)
    // A/B aid (-DENS_FWD_STAGGER=n, off by default): in a one-round launch (every workgroup resident at once, three per CU)
    // start the second and third workgroup of a CU n x 2048 cycles later, to keep their MFMA and gather / sin / store
    // stretches apart.  Measured per kernel on one box (tools/ab_kernels.sh): 82.6 us without, 85.6 / 85.8 / 86.9 us with
    // n = 1 / 2 / 3 -- the late starters finish later by more than the others gain.
    if (stag_cus > 0) {
        const int ph = (int)(blockIdx.x / (unsigned)stag_cus) % 3;
        for (int i = 0; i < ph * ENS_FWD_STAGGER; ++i) __builtin_amdgcn_s_sleep(32);
    }
    const bool wl = wli != 0;
    const int WSS = wl ? ACTL_STRIDE : ACT_STRIDE, WSV = wl ? ACTL_VOX : ACT_VOX;
    StampCtx sx;
    FST_INIT(sx)
    constexpr int RB = fwd_ring_floats(STAGE);
    float* ring = fsm;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = lane & 15, q = lane >> 4;
    float* stage = fsm + 2 * RB + wave * 256;
    const int64_t tile_raw = grp * 4 + wave;
    const bool tvalid = tile_raw < n_tiles;
    const int64_t tile = tvalid ? tile_raw : n_tiles - 1;
    const int64_t ray = tile / tiles_per_ray;
    const int64_t sidx = tile * 16 + p;

    if (role == 0) ring_load(ring, sc.packed[1], XyzLay{32}.oW(1) / 4, wave, lane);   // B^T | chunk 0
    else ring_load(ring, sc.packed[3], XyzLay{32}.oW(1) / 4, wave, lane);

    double pw[3];
    if (points != nullptr) {                                   // eval_points: the samples are given (n_tiles = ceil(n_points / 16))
        const int64_t pi = sidx < n_points ? sidx : n_points - 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) pw[a] = points[pi * 3 + a];
    } else {
        const double z = z_vals[sidx];
#pragma unroll
        for (int a = 0; a < 3; ++a) pw[a] = (double)rays_o[ray * 3 + a] + (double)rays_d[ray * 3 + a] * z;
    }
    bool inb = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) inb = inb && (pw[a] < sc.hi[a]) && (pw[a] > sc.lo[a]);
    const float pc = q == 0 ? (float)pw[0] : (q == 1 ? (float)pw[1] : (q == 2 ? (float)pw[2] : 0.f));
    const VoxNorm vn = vox_norm(pw, sc.lo, sc.hi, sc.gs);             // shared by the middle, fine and colour grids (one bound)

    const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)fsm;
    // lane base of the tile-major weight images (lds_util.hpp) and of the bias rows
    unsigned wt = lds0 + frag_off(p, q) * 4, wq = lds0 + q * 16;
    opaque(wt); opaque(wq);

    float* wsb = (act_ws != nullptr && tvalid) ? act_ws + (tile * ACT_SLOTS) * (int64_t)WSS : nullptr;
    f32x4 occ = splat4(0.f), col = splat4(0.f);

    if (role == 0) {
        f32x4 cm[2];
        {
            const Vox v = make_vox_n(vn, sc.grid[1]);
            gather8(v, sc.grid[1], q, cm[0], cm[1]);
            if (wsb != nullptr && q == 0) *reinterpret_cast<f32x4*>(wsb + WSV + p * 4) = vox_record(v, sc.grid[1]);
        }
        FST(sx, 0)      // geometry + first gather
        __syncthreads();                                                                   // chunk 0 landed
        FST(sx, 5)
        mlp_xyz_ring<2, 0, RB, (STAGE >= 2 ? 64 : 0)>(sc.packed[1], STAGE >= 2 ? sc.packed[2] : nullptr, ring, pc, cm, occ, wsb,
                                                       wl, stage, wt, wq, wave, lane, p, q, sx);
        if constexpr (STAGE >= 2) {
            f32x4 cf[4];
            {
                const Vox v = make_vox_n(vn, sc.grid[2]);
                gather8(v, sc.grid[2], q, cf[0], cf[1]);
                if (wsb != nullptr && q == 0) *reinterpret_cast<f32x4*>(wsb + WSS + WSV + p * 4) = vox_record(v, sc.grid[2]);
            }
            cf[2] = cm[0];
            cf[3] = cm[1];
            FST(sx, 7)  // later gathers
            f32x4 of;
            mlp_xyz_ring<4, 5, RB, ((STAGE == 3 && !SPLIT) ? 32 : 0)>(sc.packed[2], (STAGE == 3 && !SPLIT) ? sc.packed[3] : nullptr,
                                                                     ring, pc, cf, of, wsb ? wsb + WSS : nullptr, wl, stage, wt,
                                                                     wq, wave, lane, p, q, sx);
            occ[0] = of[0] + occ[0];                                                        // fine_occ + middle_occ
        }
    }
    if constexpr (STAGE == 3) {
        if (role == 1 || !SPLIT) {
            f32x4 cc[2];
            {
                const Vox v = make_vox_n(vn, sc.grid[3]);
                gather8(v, sc.grid[3], q, cc[0], cc[1]);
                if (wsb != nullptr && q == 0) *reinterpret_cast<f32x4*>(wsb + 2 * WSS + WSV + p * 4) = vox_record(v, sc.grid[3]);
            }
            FST(sx, 7)
            if (SPLIT) {
                __syncthreads();                                                           // chunk 0 (colour layer 0) landed
                FST(sx, 5)
            }
            mlp_xyz_ring<2, 10, RB, 0>(sc.packed[3], nullptr, ring, pc, cc, col, wsb ? wsb + 2 * WSS : nullptr, wl, stage,
                                       wt, wq, wave, lane, p, q, sx);
        }
    }
    if (q == 0 && tvalid && (points == nullptr || sidx < n_points)) {                   // rows 0..3 live on q == 0 lanes
        const float o = (inb || (points != nullptr && !apply_mask)) ? occ[0] : 100.f;   // Renderer.py:58
        if (!SPLIT) {
            *reinterpret_cast<f32x4*>(raw_out + sidx * 4) = f32x4{col[0], col[1], col[2], o};
        } else if (role == 0) {
            raw_out[sidx * 4 + 3] = o;
        } else {
            raw_out[sidx * 4 + 0] = col[0];
            raw_out[sidx * 4 + 1] = col[1];
            raw_out[sidx * 4 + 2] = col[2];
        }
    }
    FST(sx, 8)
    FST_FLUSH(sx, blockIdx.x, wave, lane)
}

// ------------------------------------------------------------------------------------------------
// Weight-stationary forward for LARGE forward-only batches (render_img, eval_points / Mesher lattices, the event term's
// rescaled image): one workgroup of 12 waves per CU keeps the forward sections of its decoders RESIDENT in LDS -- workgroups of
// role 0 the two occupancy decoders (middle 66 KB + fine 87 KB), role 1 the colour decoder (66 KB) -- and every wave walks its
// own tiles (static stride) through mlp_xyz_ring<.., RES>: no weight ring (the ring kernel streams 210 KB per 4 tiles through
// LDS), no per-layer workgroup barrier, waves of one SIMD drift apart so that one wave's gathers / sines run under another's
// MFMAs.  The two roles write disjoint components of raw (occupancy | colour).  Colour stage, no activation workspace.
// ------------------------------------------------------------------------------------------------
constexpr int fwd_res_lds_bytes() { return (XyzLay{32}.fwd_floats() + XyzLay{64}.fwd_floats()) * 4; }
constexpr int FWD_RES_WAVES = 12;
constexpr int64_t FWD_RES_MIN_TILES = 16384;        // (5 tiles per wave and more: below that the ring kernel's residency rounds win)

ENS_DEV void res_load(float* dst_lds, const float* __restrict__ src, int n4, int wave, int lane) {      // all 12 waves, 1 KB per wave instruction
    for (int j = 0; j * (64 * FWD_RES_WAVES) < n4; ++j) {
        const int e = j * (64 * FWD_RES_WAVES) + wave * 64 + lane;
        if (e < n4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * e),
                                             (__attribute__((address_space(3))) void*)(dst_lds + 4 * (j * (64 * FWD_RES_WAVES) + wave * 64)),
                                             16, 0, 0);
    }
}

__global__ __launch_bounds__(64 * FWD_RES_WAVES, 1) void render_fwd_res_kernel(int64_t n_tiles, int tiles_per_ray,
                                                                               const float* __restrict__ rays_o,
                                                                               const float* __restrict__ rays_d,
                                                                               const double* __restrict__ z_vals, DevScene sc,
                                                                               float* __restrict__ raw_out,
                                                                               const double* __restrict__ points, int64_t n_points,
                                                                               int apply_mask, int n_wg_occ) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    constexpr int MID_BYTES = XyzLay{32}.fwd_floats() * 4;
    const int role = (int)blockIdx.x < n_wg_occ ? 0 : 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = lane & 15, q = lane >> 4;
    if (role == 0) {
        res_load(fsm, sc.packed[1], XyzLay{32}.fwd_floats() / 4, wave, lane);
        res_load(fsm + XyzLay{32}.fwd_floats(), sc.packed[2], XyzLay{64}.fwd_floats() / 4, wave, lane);
    } else {
        res_load(fsm, sc.packed[3], XyzLay{32}.fwd_floats() / 4, wave, lane);
    }
    __syncthreads();                                                  // (vmcnt(0): the resident images have landed)
    StampCtx sx;
    FST_INIT(sx)
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)fsm;
    unsigned wt = lds0 + frag_off(p, q) * 4, wq = lds0 + q * 16;
    opaque(wt); opaque(wq);
    const int n_wg = role == 0 ? n_wg_occ : (int)gridDim.x - n_wg_occ;
    const int wg = role == 0 ? (int)blockIdx.x : (int)blockIdx.x - n_wg_occ;
    const int64_t stride = (int64_t)n_wg * FWD_RES_WAVES;
    for (int64_t tile = (int64_t)wave * n_wg + wg; tile < n_tiles; tile += stride) {      // (neighbouring tiles to neighbouring CUs)
        const int64_t ray = tile / tiles_per_ray;
        const int64_t sidx = tile * 16 + p;
        double pw[3];
        if (points != nullptr) {
            const int64_t pi = sidx < n_points ? sidx : n_points - 1;
#pragma unroll
            for (int a = 0; a < 3; ++a) pw[a] = points[pi * 3 + a];
        } else {
            const double z = z_vals[sidx];
#pragma unroll
            for (int a = 0; a < 3; ++a) pw[a] = (double)rays_o[ray * 3 + a] + (double)rays_d[ray * 3 + a] * z;
        }
        bool inb = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) inb = inb && (pw[a] < sc.hi[a]) && (pw[a] > sc.lo[a]);
        const float pc = q == 0 ? (float)pw[0] : (q == 1 ? (float)pw[1] : (q == 2 ? (float)pw[2] : 0.f));
        const VoxNorm vn = vox_norm(pw, sc.lo, sc.hi, sc.gs);
        const bool live = points == nullptr || sidx < n_points;
        if (role == 0) {
            f32x4 cm[2], occ = splat4(0.f);
            {
                const Vox v = make_vox_n(vn, sc.grid[1]);
                gather8(v, sc.grid[1], q, cm[0], cm[1]);
            }
            mlp_xyz_ring<2, 0, 0, 0, 0>(nullptr, nullptr, nullptr, pc, cm, occ, nullptr, false, nullptr, wt, wq, wave, lane, p, q, sx);
            f32x4 cf[4], of;
            {
                const Vox v = make_vox_n(vn, sc.grid[2]);
                gather8(v, sc.grid[2], q, cf[0], cf[1]);
            }
            cf[2] = cm[0];
            cf[3] = cm[1];
            mlp_xyz_ring<4, 0, 0, 0, MID_BYTES>(nullptr, nullptr, nullptr, pc, cf, of, nullptr, false, nullptr, wt, wq, wave, lane, p, q, sx);
            if (q == 0 && live) raw_out[sidx * 4 + 3] = (inb || (points != nullptr && !apply_mask)) ? of[0] + occ[0] : 100.f;   // Renderer.py:58
        } else {
            f32x4 cc[2], col = splat4(0.f);
            {
                const Vox v = make_vox_n(vn, sc.grid[3]);
                gather8(v, sc.grid[3], q, cc[0], cc[1]);
            }
            mlp_xyz_ring<2, 0, 0, 0, 0>(nullptr, nullptr, nullptr, pc, cc, col, nullptr, false, nullptr, wt, wq, wave, lane, p, q, sx);
            if (q == 0 && live) {
                raw_out[sidx * 4 + 0] = col[0];
                raw_out[sidx * 4 + 1] = col[1];
                raw_out[sidx * 4 + 2] = col[2];
            }
        }
    }
}

#ifdef ENS_STAMPS
}  // namespace
extern "C" int enslam_debug_set_stamp_buffer_fwd(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &v, sizeof(v)) == hipSuccess ? 0 : -2;
}
namespace {
#endif

// raw2outputs_nerf_color on its own (common.py:256-297, occupancy branch): one wave per ray; blockDim.x / 64 rays per
// workgroup (16 when the mapper loss is fused in: its terms are summed in LDS first, one fp64 atomic per workgroup --
// one per ray, all on one address, tripled the kernel's time).
__global__ __launch_bounds__(1024) void composite_fwd_kernel(int n_rays, int S, const float* __restrict__ raw,
                                                           const double* __restrict__ z_vals,
                                                           double* __restrict__ depth, double* __restrict__ var,
                                                           float* __restrict__ rgb, float* __restrict__ weights,
                                                           LossSpec ls, WorkList wk) {
    __shared__ double red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ray_raw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool rvalid = ray_raw < n_rays;
    const int64_t ray = rvalid ? ray_raw : n_rays - 1, sidx = ray * S + lane;
    const bool valid = lane < S;
    const f32x4 rw = valid ? *reinterpret_cast<const f32x4*>(raw + sidx * 4) : splat4(0.f);
    const double zk = valid ? z_vals[sidx] : 0.0;
    const float alpha = valid ? 1.f / (1.f + expf(-(10.f * rw[3]))) : 0.f;
    const float m = valid ? (1.f - alpha) + 1e-10f : 1.f;
    float incl = m;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off);
        incl = lane >= off ? incl * t : incl;
    }
    float T = __shfl_up(incl, 1);
    T = lane == 0 ? 1.f : T;
    const float w = alpha * T;
    const float cr = wave_sum(w * rw[0]), cg = wave_sum(w * rw[1]), cb = wave_sum(w * rw[2]);
    const double dep = wave_sum((double)w * zk);
    const double tmp = zk - dep;
    const double vr = wave_sum(((double)w * tmp) * tmp);
    if (weights != nullptr && valid && rvalid) weights[sidx] = w;
    double term = 0.0;
    if (lane == 0 && rvalid) {
        depth[ray] = dep;
        var[ray] = vr;
        rgb[ray * 3 + 0] = cr;
        rgb[ray * 3 + 1] = cg;
        rgb[ray * 3 + 2] = cb;
        if (ls.gd != nullptr) {                      // this ray's term of the mapper loss
            const float g = ls.gd[ray];
            term = g > 0.f ? fabs((double)g - dep) : 0.0;
            if (ls.gc != nullptr)
                term += (double)(ls.w * ((fabsf(ls.gc[ray * 3] - cr) + fabsf(ls.gc[ray * 3 + 1] - cg)) + fabsf(ls.gc[ray * 3 + 2] - cb)));
        }
    }
    if (ls.d_raw_unit != nullptr) {                  // backward of the compositing for d(total)/d(loss) = 1, same arithmetic
        const float t = ls.gd[ray];                  // as composite_bwd_kernel (render_bwd.hip) with the loss's gradients
        const double diff = (double)t - dep;
        const double gD = t > 0.f ? (diff > 0.0 ? -1.0 : (diff < 0.0 ? 1.0 : 0.0)) : 0.0;
        float gcl[3] = {0.f, 0.f, 0.f};
        if (ls.gc != nullptr) {
            const float cc[3] = {cr, cg, cb};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float d = ls.gc[ray * 3 + a] - cc[a];
                gcl[a] = -ls.w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            }
        }
        float gw = (float)(gD * zk);
        gw += gcl[0] * rw[0] + gcl[1] * rw[1] + gcl[2] * rw[2];
        gw = valid ? gw : 0.f;
        const float gww = gw * w;
        float suf = gww;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float tt = __shfl_down(suf, off);
            suf = lane + off < 64 ? suf + tt : suf;
        }
        suf -= gww;
        const float ga = gw * T - suf / m;
        const float gocc = ga * (1.f - alpha) * alpha * 10.f;
        const f32x4 dr = f32x4{gcl[0] * w, gcl[1] * w, gcl[2] * w, gocc};
        if (valid && rvalid) *reinterpret_cast<f32x4*>(ls.d_raw_unit + sidx * 4) = dr;
        if (wk.tiles != nullptr)                     // (uniform over the launch)
            append_active_tiles_wg(wk.tiles, wk.count, ray, S / 16, rvalid,
                                   valid && (dr[0] != 0.f || dr[1] != 0.f || dr[2] != 0.f || dr[3] != 0.f));
    }
    if (ls.gd != nullptr) {                          // (uniform over the launch)
        if (lane == 0) red[wave] = term;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
            atomicAdd(ls.loss, t);
        }
    }
}

// The tracker's RGB-D loss on top of the compositing, forward and backward, in TWO launches of 16 rays per workgroup (batches of
// up to ENS_TRACKER_TAIL_MAX_RAYS rays):
//   tracker_composite_kernel: raw2outputs_nerf_color per ray (one wave per ray) and
//       tmp = |gd - depth| / sqrt(var + 1e-10)                               (Tracker.py:179-181, float64 like the reference)
//   tracker_loss_kernel:
//       keep = inside & (tmp < 10 * median(tmp over the inside rays))         (:164-174 as a mask, :180-182; tracker_median_wg)
//       loss = sum_{keep & gd > 0} tmp + w * sum_{keep & gd > 0} |gc - color| (:187-195)
//     and d(loss)/d(raw) for a unit loss gradient (the variance is detached, :179), the work list of active tiles appended on
//     the way.
// The torch formulation of the same tail is 25-30 launches (sort, gather, where, two loss kernels, compositing backward).  (One
// single-workgroup launch for everything was 60 us at 200 rays: 13 dependent rounds of 16 rays, each a chain of global loads, a
// returning atomic and two barriers.)
__global__ __launch_bounds__(1024) void tracker_composite_kernel(int n_rays, int S, const float* __restrict__ raw,
                                                                 const double* __restrict__ z_vals, double* __restrict__ depth,
                                                                 double* __restrict__ var, float* __restrict__ rgb,
                                                                 const float* __restrict__ gd, TrackerSpec ts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valid = lane < S;
    const int ray = blockIdx.x * 16 + wave;
    if (ray < n_rays) {
        const int64_t sidx = (int64_t)ray * S + lane;
        const f32x4 rw = valid ? *reinterpret_cast<const f32x4*>(raw + sidx * 4) : splat4(0.f);
        const double zk = valid ? z_vals[sidx] : 0.0;
        const float alpha = valid ? 1.f / (1.f + expf(-(10.f * rw[3]))) : 0.f;
        const float m = valid ? (1.f - alpha) + 1e-10f : 1.f;
        float incl = m;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float t = __shfl_up(incl, off);
            incl = lane >= off ? incl * t : incl;
        }
        float T = __shfl_up(incl, 1);
        T = lane == 0 ? 1.f : T;
        const float w = alpha * T;
        const float cr = wave_sum(w * rw[0]), cg = wave_sum(w * rw[1]), cb = wave_sum(w * rw[2]);
        const double dep = wave_sum((double)w * zk);
        const double d0 = zk - dep;
        const double vr = wave_sum(((double)w * d0) * d0);
        if (lane == 0) {
            depth[ray] = dep;
            var[ray] = vr;
            rgb[ray * 3 + 0] = cr; rgb[ray * 3 + 1] = cg; rgb[ray * 3 + 2] = cb;
            ts.tmp[ray] = fabs((double)gd[ray] - dep) / sqrt(vr + 1e-10);
        }
    }
}

// median of tmp over the inside rays (torch.median: the lower middle element), computed by EVERY workgroup of the loss launch for
// itself from the n values the compositing launch left in memory: redundant, but parallel, and it needs neither a launch of its
// own nor a hand-over between workgroups of one launch (a "last workgroup done" ticket was tried: its two device-scope fences
// write back / invalidate the XCD's L2 and made the compositing launch 16 us instead of 5).  Up to 1024 rays by counting ranks
// (1024 / P adjacent lanes share one entry and rank it against a slice of the others, eight LDS reads in flight); above that a
// bitonic network in LDS.  Every thread of the workgroup must call this.
ENS_DEV double tracker_median_wg(const TrackerSpec& ts, int n_rays, double* srt, int* s_cnt, double* s_med) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int P = 64;
    while (P < n_rays) P <<= 1;
    int kept = 0;
    for (int i = threadIdx.x; i < P; i += 1024) {
        const bool in = i < n_rays && (ts.inside == nullptr || ts.inside[i] != 0);
        srt[i] = in ? ts.tmp[i] : INFINITY;
        kept += in ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
    if (lane == 0) s_cnt[wave] = kept;
    if (threadIdx.x == 0) *s_med = INFINITY;
    __syncthreads();
    int c = 0;
    for (int w = 0; w < 16; ++w) c += s_cnt[w];
    if (n_rays <= 1024) {
        const int parts = 1024 / P, i = threadIdx.x / parts, part = threadIdx.x - i * parts;
        const int per = (n_rays + parts - 1) / parts, j0 = part * per, j1 = min(j0 + per, n_rays);
        const double mine = srt[i];
        int rank = 0;
        int j = j0;
        for (; j + 8 <= j1; j += 8) {
            double o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = srt[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (o[u] < mine || (o[u] == mine && j + u < i)) ? 1 : 0;
        }
        for (; j < j1; ++j) {
            const double o = srt[j];
            rank += (o < mine || (o == mine && j < i)) ? 1 : 0;
        }
        for (int o = 1; o < parts; o <<= 1) rank += __shfl_xor(rank, o);          // (parts is a power of two <= 16: lanes of one wave)
        // (+inf entries: rays outside the mask, never the median unless nothing finite is kept -- then the default stands)
        if (part == 0 && i < n_rays && mine != INFINITY && c > 0 && rank == ((c - 1) >> 1)) *s_med = mine;
        __syncthreads();
        return *s_med;
    }
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += 1024) {
                const int o = i ^ j;
                if (o > i) {
                    const double a = srt[i], b = srt[o];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) { srt[i] = b; srt[o] = a; }
                }
            }
            __syncthreads();
        }
    }
    return c > 0 ? srt[(c - 1) >> 1] : INFINITY;
}

// COMPOSITE: handle_dynamic off -- no median, so the compositing (outputs, tmp) happens here too: one launch
template <bool COMPOSITE>
__global__ __launch_bounds__(1024) void tracker_loss_kernel(int n_rays, int S, const float* __restrict__ raw,
                                                            const double* __restrict__ z_vals, double* __restrict__ depth,
                                                            double* __restrict__ var, float* __restrict__ rgb,
                                                            LossSpec ls, TrackerSpec ts, WorkList wk) {
    __shared__ double red[16];
    __shared__ double srt[COMPOSITE ? 1 : ENS_TRACKER_TAIL_MAX_RAYS];
    __shared__ int s_cnt[16];
    __shared__ double s_med;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valid = lane < S;
    const int ray_raw = blockIdx.x * 16 + wave;
    const bool rvalid = ray_raw < n_rays;
    const int ray = rvalid ? ray_raw : n_rays - 1;
    const int64_t sidx = (int64_t)ray * S + lane;
    double med = INFINITY;
    if constexpr (!COMPOSITE) {
        if (ts.dynamic) med = tracker_median_wg(ts, n_rays, srt, s_cnt, &s_med);
    }
    const f32x4 rw = valid ? *reinterpret_cast<const f32x4*>(raw + sidx * 4) : splat4(0.f);
    const double zk = valid ? z_vals[sidx] : 0.0;
    const float alpha = valid ? 1.f / (1.f + expf(-(10.f * rw[3]))) : 0.f;
    const float m = valid ? (1.f - alpha) + 1e-10f : 1.f;
    float incl = m;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off);
        incl = lane >= off ? incl * t : incl;
    }
    float T = __shfl_up(incl, 1);
    T = lane == 0 ? 1.f : T;
    const float w = alpha * T;
    const float g = ls.gd[ray];
    double dep, vr, tmp;
    float cc[3];
    if constexpr (COMPOSITE) {
        cc[0] = wave_sum(w * rw[0]); cc[1] = wave_sum(w * rw[1]); cc[2] = wave_sum(w * rw[2]);
        dep = wave_sum((double)w * zk);
        const double d0 = zk - dep;
        vr = wave_sum(((double)w * d0) * d0);
        tmp = fabs((double)g - dep) / sqrt(vr + 1e-10);
        if (lane == 0 && rvalid) {
            depth[ray] = dep;
            var[ray] = vr;
            rgb[ray * 3 + 0] = cc[0]; rgb[ray * 3 + 1] = cc[1]; rgb[ray * 3 + 2] = cc[2];
            ts.tmp[ray] = tmp;
        }
    } else {
        dep = depth[ray]; vr = var[ray]; tmp = ts.tmp[ray];
        cc[0] = rgb[ray * 3 + 0]; cc[1] = rgb[ray * 3 + 1]; cc[2] = rgb[ray * 3 + 2];
    }
    const bool in = ts.inside == nullptr || ts.inside[ray] != 0;
    const bool on = rvalid && in && (!ts.dynamic || tmp < 10.0 * med) && g > 0.f;
    const double isd = 1.0 / sqrt(vr + 1e-10);
    double term = 0.0;
    if (on && lane == 0) {
        term = tmp;
        if (ls.gc != nullptr)
            term += (double)ls.w * (double)((fabsf(ls.gc[ray * 3] - cc[0]) + fabsf(ls.gc[ray * 3 + 1] - cc[1])) + fabsf(ls.gc[ray * 3 + 2] - cc[2]));
    }
    if (ls.d_raw_unit != nullptr) {                      // (uniform over the launch)
        const double diff = (double)g - dep;
        const double gD = on ? (diff > 0.0 ? -isd : (diff < 0.0 ? isd : 0.0)) : 0.0;
        float gcl[3] = {0.f, 0.f, 0.f};
        if (ls.gc != nullptr && on) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float d = ls.gc[ray * 3 + a] - cc[a];
                gcl[a] = -ls.w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            }
        }
        float gw = (float)(gD * zk);
        gw += gcl[0] * rw[0] + gcl[1] * rw[1] + gcl[2] * rw[2];
        gw = valid ? gw : 0.f;
        const float gww = gw * w;
        float suf = gww;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float tt = __shfl_down(suf, off);
            suf = lane + off < 64 ? suf + tt : suf;
        }
        suf -= gww;
        const float ga = gw * T - suf / m;
        const float gocc = ga * (1.f - alpha) * alpha * 10.f;
        const f32x4 dr = f32x4{gcl[0] * w, gcl[1] * w, gcl[2] * w, gocc};
        if (valid && rvalid) *reinterpret_cast<f32x4*>(ls.d_raw_unit + sidx * 4) = dr;
        if (wk.tiles != nullptr)
            append_active_tiles_wg(wk.tiles, wk.count, ray, S / 16, rvalid,
                                   valid && (dr[0] != 0.f || dr[1] != 0.f || dr[2] != 0.f || dr[3] != 0.f));
    }
    if (lane == 0) red[wave] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        atomicAdd(ls.loss, t);
    }
}

template <int STAGE>
int launch_stage(int ntl, int64_t n_units, const float* ro, const float* rd, const double* z, const double* pts,
                 int64_t n_points, int apply_mask, int tpr, float* act_ws, int wl, const DevScene& sc, double* depth, double* var, float* rgb, float* raw,
                 hipStream_t st) {
    if (n_units <= 0) return 0;
    const dim3 grid((unsigned)n_units), block(64);
    switch (ntl) {
        case 1: render_fwd_kernel<STAGE, 1><<<grid, block, 0, st>>>(n_units, ro, rd, z, pts, n_points, apply_mask, tpr, sc, depth, var, rgb, raw, act_ws, wl); break;
        case 2: render_fwd_kernel<STAGE, 2><<<grid, block, 0, st>>>(n_units, ro, rd, z, pts, n_points, apply_mask, tpr, sc, depth, var, rgb, raw, act_ws, wl); break;
        case 3: render_fwd_kernel<STAGE, 3><<<grid, block, 0, st>>>(n_units, ro, rd, z, pts, n_points, apply_mask, tpr, sc, depth, var, rgb, raw, act_ws, wl); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

int ens_launch_composite_fwd(int n_rays, int S, const float* raw, const double* z, double* depth, double* var,
                             float* rgb, float* weights, hipStream_t st, const LossSpec* ls, const WorkList* wl) {
    if (n_rays <= 0) return 0;
    LossSpec l{nullptr, nullptr, 0.f, nullptr, nullptr, nullptr};
    if (ls != nullptr) l = *ls;
    WorkList wk{nullptr, nullptr};
    if (wl != nullptr && l.d_raw_unit != nullptr) wk = *wl;
    if (l.gd != nullptr) composite_fwd_kernel<<<dim3((n_rays + 15) / 16), dim3(1024), 0, st>>>(n_rays, S, raw, z, depth, var, rgb, weights, l, wk);
    else composite_fwd_kernel<<<dim3(n_rays), dim3(64), 0, st>>>(n_rays, S, raw, z, depth, var, rgb, weights, l, wk);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_tracker_tail(int n_rays, int S, const float* raw, const double* z, double* depth, double* var, float* rgb,
                            const LossSpec& ls, const TrackerSpec& ts, const WorkList* wl, hipStream_t st) {
    if (n_rays <= 0) return 0;
    if (n_rays > ENS_TRACKER_TAIL_MAX_RAYS || S < 1 || S > 64 || !ls.gd || !ls.loss || !ts.tmp) return -1;
    WorkList wk{nullptr, nullptr};
    if (wl != nullptr && ls.d_raw_unit != nullptr) wk = *wl;
    if (wk.tiles != nullptr && S % 16 != 0) return -1;
    const dim3 grid((n_rays + 15) / 16), block(1024);
    if (ts.dynamic) {
        tracker_composite_kernel<<<grid, block, 0, st>>>(n_rays, S, raw, z, depth, var, rgb, ls.gd, ts);
        tracker_loss_kernel<false><<<grid, block, 0, st>>>(n_rays, S, raw, z, depth, var, rgb, ls, ts, wk);
    } else {
        tracker_loss_kernel<true><<<grid, block, 0, st>>>(n_rays, S, raw, z, depth, var, rgb, ls, ts, wk);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_render_fwd(int stage, int ntl, int64_t n_units, const float* ro, const float* rd, const double* z,
                          const double* pts, int64_t n_points, int apply_mask, const DevScene& sc, double* depth,
                          double* var, float* rgb, float* raw, float* act_ws, int act_light, hipStream_t st,
                          const LossSpec* ls, const WorkList* wl, const TrackerSpec* ts) {
    // Ray mode with raw requested: tile-per-wave decoders through the LDS weight ring + separate compositing.  (Round 1 sent
    // batches above 32768 rays to the one-wave-per-ray kernel; since the ring kernel lost its LDS bank conflicts it is the
    // faster one at every size: render_img 680 x 1200 55.9 -> 50.8 ms.  ENSLAM_TILE_MODE_MAX_RAYS restores a limit.)
    int tpr = 0;
    static const int64_t tile_mode_max = [] {                 // tuning aid: ENSLAM_TILE_MODE_MAX_RAYS
        const char* e = getenv("ENSLAM_TILE_MODE_MAX_RAYS");
        return e ? (int64_t)atoll(e) : ((int64_t)1 << 40);
    }();
    if (pts == nullptr && raw != nullptr && n_units <= tile_mode_max) {
        tpr = ntl;
        n_units *= ntl;
        ntl = 1;
    }
    // explicit points (eval_points): the same ring kernel over ceil(n_points / 16) tiles, no compositing
    const bool pts_ring = pts != nullptr && raw != nullptr && stage >= 1 && stage <= 3 && n_points <= tile_mode_max * 48;
    if (pts_ring) { tpr = 1; n_units = (n_points + 15) / 16; }
    if (ls != nullptr && tpr == 0) return -1;                 // the fused loss rides in the separate compositing launch
    int rc;
    if (tpr > 0 && stage >= 1 && stage <= 3) {
        static bool attr_done[ENS_MAX_DEVICES] = {};
        bool& attr_set = attr_done[ens_device_ordinal()];
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(render_fwd_ring_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, fwd_ring_lds_bytes(1)) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(render_fwd_ring_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, fwd_ring_lds_bytes(2)) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(render_fwd_ring_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, fwd_ring_lds_bytes(3)) != hipSuccess)
                return -2;
            attr_set = true;
        }
        // large forward-only colour-stage batches: the weight-stationary kernel (ENSLAM_FWD_RES=0 / 1 forces the ring / resident form)
        static const int res_env = [] { const char* e = getenv("ENSLAM_FWD_RES"); return e ? atoi(e) : -1; }();
        if (stage == 3 && act_ws == nullptr && raw != nullptr && (res_env >= 0 ? res_env != 0 : n_units >= FWD_RES_MIN_TILES)) {
            static bool res_done[ENS_MAX_DEVICES] = {};
            bool& res_attr = res_done[ens_device_ordinal()];
            if (!res_attr) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(render_fwd_res_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        fwd_res_lds_bytes()) != hipSuccess)
                    return -2;
                res_attr = true;
            }
            const int cus_r = ens_device_cus();
            if (cus_r < 2) return -1;                                // (the two roles need a workgroup each)
            // occupancy decoders : colour decoder = 588 : 254 MFMAs per tile (70 : 30), but the colour role also carries a grid gather,
            // an embedding and the output stores per tile: two thirds of the workgroups for role 0 measured best (eval_points, 16 M
            // points: 160 / 170 / 179 / 188 / 196 of 256 -> 887 / 946 / 935 / 834 / 740 M points/s)
            static const int occ_env = [] { const char* e = getenv("ENSLAM_FWD_RES_OCC"); return e ? atoi(e) : 0; }();      // A/B aid
            int n_occ = (occ_env > 0 && occ_env < cus_r) ? occ_env : (cus_r * 2 + 1) / 3;
            n_occ = n_occ < 1 ? 1 : (n_occ > cus_r - 1 ? cus_r - 1 : n_occ);       // both roles get at least one workgroup
            render_fwd_res_kernel<<<dim3(cus_r), dim3(64 * FWD_RES_WAVES), fwd_res_lds_bytes(), st>>>(n_units, tpr, ro, rd, z, sc, raw, pts,
                                                                                                     n_points, apply_mask, n_occ);
            if (hipGetLastError() != hipSuccess) return -2;
            if (pts_ring) return 0;
            if (ts != nullptr && ls != nullptr)
                return ens_launch_tracker_tail((int)(n_units / tpr), 16 * tpr, raw, z, depth, var, rgb, *ls, *ts, wl, st);
            return ens_launch_composite_fwd((int)(n_units / tpr), 16 * tpr, raw, z, depth, var, rgb, nullptr, st, ls, wl);
        }
        const int64_t groups = (n_units + 3) / 4;
        // colour stage: two roles per group of 4 tiles, interleaved in runs of 8 blocks (render_fwd_ring_kernel)
        const dim3 grid((unsigned)(fwd_split_roles(stage) ? ((groups + 7) / 8) * 16 : groups)), block(256);
        const int cus = ens_device_cus();
        const int stag = (ENS_FWD_STAGGER > 0 && !fwd_split_roles(stage) && grid.x <= (unsigned)(3 * cus)) ? cus : 0;
        if (stage == 1) render_fwd_ring_kernel<1><<<grid, block, fwd_ring_lds_bytes(1), st>>>(n_units, tpr, ro, rd, z, sc, raw, act_ws, act_light, stag, pts, n_points, apply_mask);
        else if (stage == 2) render_fwd_ring_kernel<2><<<grid, block, fwd_ring_lds_bytes(2), st>>>(n_units, tpr, ro, rd, z, sc, raw, act_ws, act_light, stag, pts, n_points, apply_mask);
        else render_fwd_ring_kernel<3><<<grid, block, fwd_ring_lds_bytes(3), st>>>(n_units, tpr, ro, rd, z, sc, raw, act_ws, act_light, stag, pts, n_points, apply_mask);
        if (hipGetLastError() != hipSuccess) return -2;
        if (pts_ring) return 0;
        if (ts != nullptr && ls != nullptr)             // the tracker's loss (median mask included) in place of the compositing launch
            return ens_launch_tracker_tail((int)(n_units / tpr), 16 * tpr, raw, z, depth, var, rgb, *ls, *ts, wl, st);
        return ens_launch_composite_fwd((int)(n_units / tpr), 16 * tpr, raw, z, depth, var, rgb, nullptr, st, ls, wl);
    }
    if (ts != nullptr) return -1;
    switch (stage) {
        case 0: rc = launch_stage<0>(ntl, n_units, ro, rd, z, pts, n_points, apply_mask, tpr, act_ws, act_light, sc, depth, var, rgb, raw, st); break;
        case 1: rc = launch_stage<1>(ntl, n_units, ro, rd, z, pts, n_points, apply_mask, tpr, act_ws, act_light, sc, depth, var, rgb, raw, st); break;
        case 2: rc = launch_stage<2>(ntl, n_units, ro, rd, z, pts, n_points, apply_mask, tpr, act_ws, act_light, sc, depth, var, rgb, raw, st); break;
        case 3: rc = launch_stage<3>(ntl, n_units, ro, rd, z, pts, n_points, apply_mask, tpr, act_ws, act_light, sc, depth, var, rgb, raw, st); break;
        default: return -1;
    }
    if (rc != 0 || tpr == 0) return rc;
    return ens_launch_composite_fwd((int)(n_units / tpr), 16 * tpr, raw, z, depth, var, rgb, nullptr, st, ls, wl);
}
