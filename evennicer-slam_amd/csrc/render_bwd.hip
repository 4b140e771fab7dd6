// Hand-derived backward of the rendering hot path for gfx950.
//
//   composite_bwd_kernel : one wave per ray; d(depth,var,rgb) -> d_raw[sample] = (dr,dg,db,docc)
//                          (backward of common.py:284-296, occupancy branch; suffix sums by shuffles)
//   decoder_bwd_kernel   : persistent workgroups (one per CU, 4 waves), each workgroup specialised on ONE
//                          decoder of the stage.  A wave takes a 16-sample tile at a time:
//                            recompute gather + embedding + forward chain in registers (nothing is saved by
//                            the forward but raw/z) -> backward chain on MFMA (dX = W^T dY) ->
//                            weight gradients as MFMA outer products over the tile's 16 samples, operands
//                            transposed through wave-private LDS, accumulated with LDS atomics into ONE
//                            workgroup-wide accumulator that is flushed once at the end ->
//                            feature gradient scattered to the voxel-major grid gradient with 256-byte
//                            contiguous float atomics; coordinate gradient reduced per ray.
//
// Gradient semantics follow the reference's autograd: no gradient through the out-of-bound overwrite
// (Renderer.py:58: sigmoid'(10*100) is exactly 0), none to grid_middle / the position through the fine
// decoder's concatenated middle feature (decoder.py:184-186), none to z_vals, and the colour decoder's
// 4th output is unused (decoder.py:338-341).
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"
#include "lds_util.hpp"
#include "bwd_shared.hpp"

#define IC(n) std::integral_constant<int, n>{}

// Diagnostic build only (-DENS_STAMPS): per-segment cycle counts of each wave (s_memtime), written to a
// buffer of their own; never compiled into the shipped library.
#include "stamps.hpp"

namespace {


// ------------------------------------------------------------------------------------------------
// composite backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void composite_bwd_kernel(int n_rays, int S, const float* __restrict__ raw,
                                                           const double* __restrict__ z_vals,
                                                           const double* __restrict__ depth,
                                                           const double* __restrict__ g_depth,
                                                           const double* __restrict__ g_var,
                                                           const float* __restrict__ g_rgb,
                                                           float* __restrict__ d_raw, LossSpec ls,
                                                           const float* __restrict__ rgb, WorkList wk) {
    const int lane = threadIdx.x & 63;               // one wave per ray, blockDim.x / 64 rays per workgroup
    const int64_t ray_raw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool rvalid = ray_raw < n_rays;
    const int64_t ray = rvalid ? ray_raw : n_rays - 1, sidx = ray * S + lane;
    const bool valid = lane < S;
    f32x4 rw = valid ? *reinterpret_cast<const f32x4*>(raw + sidx * 4) : splat4(0.f);
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
    const double dep = depth[ray];
    double gD = g_depth ? g_depth[ray] : 0.0;
    const double gV = g_var ? g_var[ray] : 0.0;
    float gc[3] = {0.f, 0.f, 0.f};
    if (g_rgb) { gc[0] = g_rgb[ray * 3]; gc[1] = g_rgb[ray * 3 + 1]; gc[2] = g_rgb[ray * 3 + 2]; }
    if (ls.gd != nullptr) {                          // gradients of the fused mapper loss instead of incoming ones
        const double g = ls.g_loss[0];
        const float t = ls.gd[ray];
        const double diff = (double)t - dep;
        gD = t > 0.f ? (diff > 0.0 ? -g : (diff < 0.0 ? g : 0.0)) : 0.0;
        if (ls.gc != nullptr) {
            const float gw = (float)g * ls.w;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float d = ls.gc[ray * 3 + a] - rgb[ray * 3 + a];
                gc[a] = -gw * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            }
        }
    }
    const double tmp = zk - dep;
    // var depends on depth through tmp:  d var / d depth = -2 sum_k w_k tmp_k
    const double gDt = gD - 2.0 * gV * wave_sum((double)w * tmp);
    float gw = (float)(gDt * zk + gV * tmp * tmp);
    gw += gc[0] * rw[0] + gc[1] * rw[1] + gc[2] * rw[2];
    gw = valid ? gw : 0.f;
    // exclusive suffix sum of gw*w  (cumprod backward: d m_j = sum_{k>j} gw_k w_k / m_j)
    const float gww = gw * w;
    float suf = gww;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_down(suf, off);
        suf = lane + off < 64 ? suf + t : suf;
    }
    suf -= gww;
    const float ga = gw * T - suf / m;
    const float gocc = ga * (1.f - alpha) * alpha * 10.f;
    const f32x4 dr = f32x4{gc[0] * w, gc[1] * w, gc[2] * w, gocc};
    if (valid && rvalid) *reinterpret_cast<f32x4*>(d_raw + sidx * 4) = dr;
    if (wk.tiles != nullptr)                         // (uniform over the launch)
        append_active_tiles_wg(wk.tiles, wk.count, ray, S / 16, rvalid,
                               valid && (dr[0] != 0.f || dr[1] != 0.f || dr[2] != 0.f || dr[3] != 0.f));
}

// ------------------------------------------------------------------------------------------------
// LDS helpers (wave-private transposition scratch + workgroup-wide gradient accumulator)
// ------------------------------------------------------------------------------------------------
// A "deposit" stores register tiles so that they can be read back with the SAMPLE index in the MFMA K slot:
//   float address(tile T, feature i in 0..15, sample pt in 0..15) = T*256 + (pt>>2)*64 + i*4 + (pt&3)
// so the fragment of tile T is the lane-linear 16-byte read at T*256 + lane*4, whose component s belongs to
// (feature lane&15, sample 4*(lane>>4)+s): A and B operands of a 16x16x4 MFMA that sums over samples.
ENS_DEV void deposit(float* dep, int T, const f32x4& x, int p, int q) {
    float* d = dep + T * 256 + (p >> 2) * 64 + (p & 3) + 16 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[4 * r] = x[r];
}
ENS_DEV f32x4 frag(const float* dep, int T, int lane) { return *reinterpret_cast<const f32x4*>(dep + T * 256 + lane * 4); }

// ---- owner-computes weight gradients ----------------------------------------------------------
// The 4 waves of a workgroup work in lockstep on 4 sample tiles.  Each wave deposits its tile's operands
// in its own LDS slot; after a barrier every wave accumulates the dW output tiles IT OWNS (tile index
// t = wave + 4*j of the matrix' row-major tile grid) over all 4 slots, i.e. over 64 samples, directly into
// persistent MFMA accumulators.  No LDS atomics; one staged flush at the end of the kernel.
//   acc[j] += sum_{slot, sample} Y[slot][sample][16*rt + i] * X[slot][sample][16*ct + j']
template <int NJ>
ENS_DEV void own_outer(f32x4 (&acc)[NJ], const float* slots, int slot_stride, int ytile0, int xtile0, int nc,
                       int ntiles, int wave, int lane) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        const float* base = slots + sl * slot_stride;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int t = wave + 4 * j;
            if (t < ntiles) {                                   // wave-uniform
                const int rt = t / nc, ct = t - rt * nc;
                const f32x4 a = frag(base, ytile0 + rt, lane), b = frag(base, xtile0 + ct, lane);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[j] = MFMA16(a[s], b[s], acc[j]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the scheduler from hoisting all slots' fragment reads
    }
}
ENS_DEV void own_outer_1(f32x4& acc, const float* slots, int slot_stride, int ytile0, int xtile0, int nc, int ntiles,
                         int wave, int lane) {
    f32x4 t[1] = {acc};
    own_outer<1>(t, slots, slot_stride, ytile0, xtile0, nc, ntiles, wave, lane);
    acc = t[0];
}
// bias gradient: acc += sum_{slot, sample} Y[slot][sample][16*rt + i]   (every column of the tile identical)
ENS_DEV void own_bias(f32x4& acc, const float* slots, int slot_stride, int ytile, int lane) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        const f32x4 a = frag(slots + sl * slot_stride, ytile, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = MFMA16(a[s], 1.f, acc);
    }
}
// acc[rt] += M[(16rt+p)*ld + 16t + 4q ..] * x[t]   (NR output row tiles, KT input tiles), single sample tile
template <int NR, int KT>
ENS_DEV void linear_n(f32x4 (&acc)[NR], const float* __restrict__ M, int ld, const f32x4 (&x)[KT], int p, int q) {
    f32x4 a[KT][NR];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) a[t][rt] = ldw(M, (16 * rt + p) * ld + 16 * t + 4 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int rt = 0; rt < NR; ++rt) acc[rt] = MFMA16(a[t][rt][r], x[t][r], acc[rt]);
        }
    }
}

// the same on a swizzled image (lds_util.hpp) read straight from global memory
template <int NR, int KT>
ENS_DEV void linear_n_swz(f32x4 (&acc)[NR], const float* __restrict__ M, int ld, const f32x4 (&x)[KT], int p, int q) {
    f32x4 a[KT][NR];
    const int s = p >> 1;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) a[t][rt] = ldw(M, (16 * rt + p) * ld + 4 * ((4 * t + q) ^ s));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int rt = 0; rt < NR; ++rt) acc[rt] = MFMA16(a[t][rt][r], x[t][r], acc[rt]);
        }
    }
}

ENS_DEV unsigned pos_bits(const f32x4& v) {
    return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
}

// ------------------------------------------------------------------------------------------------
// Per-tile geometry shared by the decoder roles
// ------------------------------------------------------------------------------------------------


// Scatter the tile's feature gradient (deposited as [sample][32] floats in `dep`) into the voxel-major grid
// gradient: 4 wave instructions per flush, each 2 x-adjacent corners x 32 channels = 256 contiguous bytes.
// Consecutive samples of a ray that fall into the same voxel (all 16 near-surface samples share 1-3 voxels)
// are combined in registers first, so they cost one flush instead of one each.
ENS_DEV void scatter_tile(const float* dep, const Vox& v, const DevGrid& gg, int lane) {
    const int ch = lane & 31, dxb = lane >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int cx = -1, cy = -1, cz = -1;
    auto flush = [&]() {
        const int x = cx + dxb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = cy + (k & 1), z = cz + (k >> 1);
            const bool ok = (x < gg.W) && (y < gg.H) && (z < gg.D);
            if (ok && acc[k] != 0.f) atomicAdd(gg.data + (((int64_t)z * gg.H + y) * gg.W + x) * 32 + ch, acc[k]);
            acc[k] = 0.f;
        }
    };
#pragma unroll
    for (int pt = 0; pt < 16; ++pt) {
        const float val = dep[pt * 32 + ch];
        if (!__any(val != 0.f)) continue;                 // e.g. masked samples of an occupancy decoder
        const int ix = __builtin_amdgcn_readlane(v.ix, pt), iy = __builtin_amdgcn_readlane(v.iy, pt),
                  iz = __builtin_amdgcn_readlane(v.iz, pt);
        const float fx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.fx), pt));
        const float fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.fy), pt));
        const float fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.fz), pt));
        if (ix != cx || iy != cy || iz != cz) {          // scalar comparison: new voxel
            if (cx >= 0) flush();
            cx = ix; cy = iy; cz = iz;
        }
        const float wx = dxb ? fx : (1.f - fx);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float w = (wx * ((k & 1) ? fy : (1.f - fy))) * ((k >> 1) ? fz : (1.f - fz));
            acc[k] = fmaf(w, val, acc[k]);
        }
    }
    if (cx >= 0) flush();
}

// ------------------------------------------------------------------------------------------------
// MLP (middle / fine / color) backward for one workgroup role
// ------------------------------------------------------------------------------------------------
template <int CT>
struct XyzSlots {                      // tile offsets inside one wave's LDS deposit slot (1 tile = 256 floats)
    static constexpr int EMB = 0;      // 6: embedding (later d_arg); followed by h2 so that W3's input [emb|h2] is contiguous
    static constexpr int HX2 = 6, HX0 = 8, HX1 = 10, HX3 = 12;            // 2 each: hidden activations h_i (dW operands)
    static constexpr int C = 14;       // CT: grid features
    static constexpr int H0 = 14 + CT, H1 = 16 + CT;     // dh_i, double buffered by layer parity
    static constexpr int P0 = 18 + CT, P1 = 20 + CT;     // dpre_i
    static constexpr int Q = 22 + CT;                    // sample coordinates (dB^T operand)
    static constexpr int TILES = 23 + CT;
};
// weight ring: two buffers, each holds one layer's chunk (forward: W|b|Wc|bc ; backward: W^T|Wc^T)
constexpr int ring_floats(int ct) { return cmax(32 * 128 + 32 + 32 * 16 * ct + 32, 128 * 32 + 1024); }


__shared__ int ens_vote[2][4];      // per-round activity flags of the 4 waves (double buffered by round parity)

// ---- elastic hand-offs of decoder_bwd_split_kernel ------------------------------------------------------------------
// Inside a round the chain waves (0..3) and the dW waves (4..7) of a workgroup no longer meet at workgroup barriers: they
// hand tiles over through monotonic counters in LDS (one ds_add per wave and hand-off, polled with ds_read + s_sleep).
// A chain wave therefore never waits for the dW waves' MFMAs of the layer it has just deposited, the dW waves run up to
// two layers behind and fill the MFMA pipe while the chain wave is in its vector / scalar stretches (feature-gradient
// scatter, embedding tail, deposits), and no wait includes vmcnt(0) -- a __syncthreads() made every chain wave wait for
// the acknowledges of the float atomics it had just issued.  One workgroup barrier per round (the activity vote) remains.
//   SY_DEP+i   chain waves have deposited dh_i / dpre_i of layer i                    (+4 per executed round)
//   SY_DW+i    dW waves have read everything layer i's owned products need          (+4 per executed round)
//   SY_DARG    chain waves have deposited d_arg (tiles HX2..HX1 of their slot)      (+4 per executed round)
//   SY_FILLA   the first part of the dW waves' slot fills has landed: h3 and the grid features, all layer 4 needs (+4 per executed round)
//   SY_FILL    ... and the rest (embedding, h0..h2, coordinates)                     (+4 per executed round)
//   SY_RING    a streamed W^T chunk has landed: layer 3's, then layer 0's (layers 4, 2, 1 are resident; the chain waves
//              request the chunks themselves, a quarter each)                          (+4 per chunk, 2 chunks per executed round)
//   SY_RDONE   chain waves have read the streamed chunk (the buffer may take the next) (+4 per chunk)
//   SY_STG     chain waves have staged dC of their tile for the scatter              (+4 per executed round)
//   SY_STGDONE dW waves have taken the staged dC of the previous round into registers (+4 per executed round after the first)
// The feature-gradient scatter (8 k of the chain wave's 43 k cycles per round, branchy scalar code with float atomics) runs
// on the dW waves: dW wave v scatters chain wave v's previous tile in four 4-sample pieces between its owned products.
// LDS operations of one wave execute in issue order, so a counter increment issued after a wave's writes (reads) is seen
// only after them.  Every wait is bounded (ENS_SPIN_LIMIT polls): a protocol error ends in wrong numbers, which the
// parity tests catch, never in a hung GPU.
enum { SY_DEP = 0, SY_DW = 5, SY_DARG = 10, SY_FILL = 11, SY_RING = 12, SY_RDONE = 13, SY_STG = 14, SY_STGDONE = 15, SY_FILLA = 16, SY_N = 17 };
__shared__ int ens_sync[SY_N];
constexpr int ENS_SPIN_LIMIT = 1 << 21;
ENS_DEV void sy_signal(int idx, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&ens_sync[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
ENS_DEV void sy_wait(int idx, int target) {
    for (int it = 0; it < ENS_SPIN_LIMIT; ++it) {
        const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ens_sync[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
// owned outer products with integer addressing: fb[sl] = slot sl base + frag_lane_off(lane)
template <int NJ>
ENS_DEV void own_outer_a(f32x4 (&acc)[NJ], const unsigned (&fb)[4], int ytile0, int xtile0, int nc, int ntiles, int wave) {
    // tiles t >= ntiles (only the 6-tile dB^T matrix has them) read tile 0 and are discarded by the caller's
    // staging (it never stages t >= ntiles): no divergent control flow around the MFMAs
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        f32x4 a[NJ], b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int t = wave + 4 * j;
            t = t < ntiles ? t : 0;
            const int rt = t / nc, ct = t - rt * nc;
            a[j] = lds4(fb[sl] + (ytile0 + rt) * 1024);
            b[j] = lds4(fb[sl] + (xtile0 + ct) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = MFMA16(a[j][s], b[j][s], acc[j]);
        }
    }
}
// One layer's owned weight-gradient tiles in one pass over the slots: dWc (NA tiles), dW (NB tiles) and the bias row
// (per-lane partial sums on the VALU -- a 16-MFMA tile per layer for one row was 20 % of the dW waves' MFMA time;
// stage_bias_lane reduces them over q once, at the flush).
// Per slot all operand fragments are read first, then the MFMAs of the NA+NB+1 independent accumulators interleave,
// so no MFMA waits on its own predecessor and one LDS round trip feeds 4*(NA+NB+1) MFMAs (three separate passes with
// one or two accumulators each ran at about half the MFMA rate).
template <int NA, int NB>
ENS_DEV void own_layer_a(f32x4 (&accA)[NA], int yA, int xA, int ncA, int ntA, f32x4 (&accB)[NB], int yB, int xB, int ncB,
                         int ntB, float& accBias, int ybias, const unsigned (&fb)[4], int wave) {
    // operand fragments of the next slot are requested before the current slot's MFMAs are issued (two register sets),
    // so the LDS round trip of slot s+1 runs under the MFMAs of slot s
#ifdef ENS_EXP_NOPIPE            // A/B aid: single fragment set (fewer registers, LDS latency of every slot exposed)
    constexpr bool PIPE = false;
#else
    constexpr bool PIPE = (NA + NB) <= 4;
#endif
    f32x4 aA[2][NA], bA[2][NA], aB[2][NB], bB[2][NB], ab[2];
    auto load = [&](int sl, int set) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int t = wave + 4 * j;
            t = t < ntA ? t : 0;
            const int rt = t / ncA, ct = t - rt * ncA;
            aA[set][j] = lds4(fb[sl] + (yA + rt) * 1024);
            bA[set][j] = lds4(fb[sl] + (xA + ct) * 1024);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int t = wave + 4 * j;
            t = t < ntB ? t : 0;
            const int rt = t / ncB, ct = t - rt * ncB;
            aB[set][j] = lds4(fb[sl] + (yB + rt) * 1024);
            bB[set][j] = lds4(fb[sl] + (xB + ct) * 1024);
        }
        ab[set] = lds4(fb[sl] + ybias * 1024);
    };
    if constexpr (PIPE) load(0, 0);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        const int set = PIPE ? (sl & 1) : 0;
        if constexpr (PIPE) {
            if (sl + 1 < 4) load(sl + 1, set ^ 1);
        } else {
            load(sl, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < NA; ++j) accA[j] = MFMA16(aA[set][j][s], bA[set][j][s], accA[j]);
#pragma unroll
            for (int j = 0; j < NB; ++j) accB[j] = MFMA16(aB[set][j][s], bB[set][j][s], accB[j]);
        }
        accBias += (ab[set][0] + ab[set][1]) + (ab[set][2] + ab[set][3]);     // bias row: feature p, samples 4q..4q+3 (VALU)
    }
}
ENS_DEV void own_bias_a(f32x4& acc, const unsigned (&fb)[4], int ytile) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        const f32x4 a = lds4(fb[sl] + ytile * 1024);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = MFMA16(a[s], 1.f, acc);
    }
}

template <int CT, int NOUT>
ENS_DEV void xyz_role(const BwdArgs& A, int kind, int wg, int n_wg, float* smem) {
    constexpr XyzLay L{CT * 16};
    constexpr int CD = CT * 16;
    constexpr int GF = L.fwd_floats();
    constexpr int NE = NOUT == 4 ? 3 : 1;                          // outputs that carry gradient
    using SL = XyzSlots<CT>;
    constexpr int SLOT = SL::TILES * 256;                          // floats
    constexpr int RB = ring_floats(CT);                            // floats per ring buffer
    constexpr int RING_BYTES = 2 * RB * 4;                         // LDS: [ring | 4 deposit slots]; flush image aliases the slots
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = lane & 15, q = lane >> 4;
    const float* __restrict__ pk = A.sc.packed[kind];
    const DevGrid grid = A.sc.grid[kind];
    const DevGrid ggrid = A.ggrid[kind];
    float* gpk = A.gpacked[kind];
    const bool want_w = gpk != nullptr, want_g = ggrid.data != nullptr, want_r = A.g_ro != nullptr;
    const bool want_c = want_g || want_r;
    float* ring = smem;
    float* slots = smem + 2 * RB;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)smem;   // LDS byte address of the dynamic region

    // chunk c of a round: 0..4 forward layers, 5..9 backward layers 4..0, 10 = layer 0 of the next round
    auto prefetch = [&](auto ic) {
        constexpr int c = decltype(ic)::value;
        float* dst = ring + ((c & 1) ? RB : 0);
        if constexpr (c < 5 || c == 10) {
            constexpr int i = c == 10 ? 0 : c;
            ring_load(dst, pk + L.oW(i), (L.oW(i + 1) - L.oW(i)) / 4, wave, lane);
        } else {
            constexpr int i = 9 - c;
            ring_load(dst, pk + L.oWT(i), 8 * L.K(i), wave, lane);
            ring_load(dst + 32 * L.K(i), pk + L.oWcT(i), 256, wave, lane);
        }
    };

    // owned weight-gradient accumulators (tile t = wave + 4*j of each matrix), persistent over all rounds
    f32x4 aWc[5][CT / 2], aW0[3], aW1[1], aW2[1], aW3[4], aW4[1], aB[5], aBT[2];
    float aWo[NE][8], aBo[NE];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        aB[i] = splat4(0.f);
#pragma unroll
        for (int j = 0; j < CT / 2; ++j) aWc[i][j] = splat4(0.f);
    }
    aW0[0] = aW0[1] = aW0[2] = aW1[0] = aW2[0] = aW4[0] = splat4(0.f);
    aW3[0] = aW3[1] = aW3[2] = aW3[3] = splat4(0.f);
    aBT[0] = aBT[1] = splat4(0.f);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        aBo[j] = 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f) aWo[j][f] = 0.f;
    }

    const int64_t n_tiles = (int64_t)A.n_rays * A.ntl;
    const int S = 16 * A.ntl;
    const int64_t stride = (int64_t)n_wg * 4;
    STAMP_DECL
    STAMP_START
    prefetch(IC(0));
    unsigned round_no = 0;
    for (int64_t base = (int64_t)wg * 4; base < n_tiles; base += stride) {
        const int64_t tile_raw = base + wave;
        const bool tvalid = tile_raw < n_tiles;
        const int tile = __builtin_amdgcn_readfirstlane((int)(tvalid ? tile_raw : n_tiles - 1));
        const TileGeo G = tile_geo(tile, A.ntl, S, A.ro, A.rd, A.z, p);
        f32x4 draw = *reinterpret_cast<const f32x4*>(A.d_raw + G.sidx * 4) * draw_scale_of(A);
        if (!tvalid) draw = splat4(0.f);
        float dj[NE];                                               // d(loss)/d(output j) of this lane's sample
        if constexpr (NOUT == 4) { dj[0] = draw[0]; dj[1] = draw[1]; dj[2] = draw[2]; } else { dj[0] = draw[3]; }
        bool nz = false;
#pragma unroll
        for (int j = 0; j < NE; ++j) nz = nz || dj[j] != 0.f;
        // nothing flows into any of the 4 tiles: skip the round (the waves of a workgroup stay in lockstep).
        // One barrier: each wave posts its flag in a word of the round's parity set.
        {
            const int par = (int)(round_no & 1);
            if (lane == 0) ens_vote[par][wave] = __any(nz) ? 1 : 0;
            ++round_no;
            __syncthreads();
            const int any4 = ens_vote[par][0] | ens_vote[par][1] | ens_vote[par][2] | ens_vote[par][3];
            if (!any4) continue;
        }
        STAMP(0)        // tile geometry + d_raw load + vote barrier

        // per-lane LDS bases of this round (opaque: see above)
        unsigned wt = lds0 + frag_off(p, q) * 4, wq = lds0 + q * 16;            // forward images: tile-major (lds_util.hpp)
        unsigned dep[4];
        dep_bases(dep, lds0 + RING_BYTES + wave * SLOT * 4, p, q);
        unsigned fb[4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) { fb[sl] = lds0 + RING_BYTES + sl * SLOT * 4 + frag_lane_off(lane); opaque(fb[sl]); }
        unsigned w32s = swz_base_even(lds0, 32, p, q);               // transposed (swizzled) images: lds_util.hpp
        const unsigned swd = swz_odd_delta(p);
        opaque(wt); opaque(wq); opaque(w32s);

        // ---- recompute the forward chain (weights of layer i from ring buffer i&1, next chunk in flight)
        const float pc = q == 0 ? (float)G.pw[0] : (q == 1 ? (float)G.pw[1] : (q == 2 ? (float)G.pw[2] : 0.f));
        const Vox v = make_vox(G.pw, A.sc.lo, A.sc.hi, grid);
        f32x4 c[CT];
        gather8(v, grid, q, c[0], c[1]);
        if constexpr (CT == 4) {
            const Vox vm = make_vox(G.pw, A.sc.lo, A.sc.hi, A.sc.grid[1]);
            gather8(vm, A.sc.grid[1], q, c[2], c[3]);
        }
        STAMP(1)        // voxel setup + gather
        f32x4 emb[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const float a = pk[L.oBT() + (16 * t + p) * 4 + q];
            const f32x4 arg = MFMA16(a, pc, splat4(0.f));
#pragma unroll
            for (int r = 0; r < 4; ++r) emb[t][r] = ens_sinf(arg[r]);
        }
        STAMP(2)        // embedding
        f32x4 h[5][2];
        unsigned mbits[5];
        auto fwd_layer = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int RO = (i & 1) ? RB * 4 : 0;                 // byte offset of this layer's ring buffer
            constexpr int K = L.K(i);
            constexpr int OB = RO + 32 * K * 4, OC = OB + 32 * 4, OBC = OC + 32 * CD * 4;
            prefetch(IC(i + 1));
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
            mbits[i] = pos_bits(acc[0]) | (pos_bits(acc[1]) << 4);
            acc[0] = relu4(acc[0]) + lds4(wq + OBC);
            acc[1] = relu4(acc[1]) + lds4(wq + OBC + 64);
            lin_lds_tm<2, CT, CD, OC>(acc, wt, c);
            h[i][0] = acc[0];
            h[i][1] = acc[1];
            if constexpr (i == 4) {
                if (want_w) {       // dW operands that exist now: park them in LDS, free the registers
                    dep_tile<SL::EMB + 0>(dep, emb[0]); dep_tile<SL::EMB + 1>(dep, emb[1]); dep_tile<SL::EMB + 2>(dep, emb[2]);
                    dep_tile<SL::EMB + 3>(dep, emb[3]); dep_tile<SL::EMB + 4>(dep, emb[4]); dep_tile<SL::EMB + 5>(dep, emb[5]);
                    dep_tile<SL::C + 0>(dep, c[0]); dep_tile<SL::C + 1>(dep, c[1]);
                    if constexpr (CT == 4) { dep_tile<SL::C + 2>(dep, c[2]); dep_tile<SL::C + 3>(dep, c[3]); }
                    dep_tile<SL::HX0>(dep, h[0][0]); dep_tile<SL::HX0 + 1>(dep, h[0][1]);
                    dep_tile<SL::HX1>(dep, h[1][0]); dep_tile<SL::HX1 + 1>(dep, h[1][1]);
                    dep_tile<SL::HX2>(dep, h[2][0]); dep_tile<SL::HX2 + 1>(dep, h[2][1]);
                    dep_tile<SL::HX3>(dep, h[3][0]); dep_tile<SL::HX3 + 1>(dep, h[3][1]);
                }
            }
            __syncthreads();            // next chunk has landed; everybody is done with this buffer
        };
        fwd_layer(IC(0)); fwd_layer(IC(1)); fwd_layer(IC(2)); fwd_layer(IC(3)); fwd_layer(IC(4));
        STAMP(3)        // forward chain

        // ---- output layer: dWo, dbo on the VALU (n_out <= 4 rows: not worth an MFMA tile); dh4 = Wo^T d_out
        if (want_w) {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) aWo[j][4 * t + r] = fmaf(dj[j], h[4][t][r], aWo[j][4 * t + r]);
                }
                if (q == 0) aBo[j] += dj[j];
            }
        }
        f32x4 dh[2] = {splat4(0.f), splat4(0.f)};
        {   // K = 4: one MFMA step per row tile; k-slot q carries output q
            const float dq = NOUT == 4 ? (q == 0 ? draw[0] : (q == 1 ? draw[1] : (q == 2 ? draw[2] : 0.f)))
                                       : (q == 0 ? draw[3] : 0.f);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) dh[rt] = MFMA16(pk[L.oWoT() + (16 * rt + p) * 4 + q], dq, dh[rt]);
        }
        f32x4 dc[2] = {splat4(0.f), splat4(0.f)};
        f32x4 demb[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) demb[t] = splat4(0.f);
        auto bwd_layer = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int cidx = 9 - i;
            constexpr int RO = (cidx & 1) ? RB * 4 : 0;              // W_i^T [K][32] | Wc_i^T [32][32]
            constexpr int OCT = RO + 32 * L.K(i) * 4;
            constexpr int TH = (i & 1) ? SL::H1 : SL::H0, TP = (i & 1) ? SL::P1 : SL::P0;
            constexpr int TX = i == 4 ? SL::HX3 : (i == 2 ? SL::HX1 : SL::HX0);      // input h_{i-1} of layers 4, 2, 1
            f32x4 dpre[2] = {mask4(dh[0], mbits[i], 0), mask4(dh[1], mbits[i], 4)};
            if (want_w) {
                dep_tile<TH>(dep, dh[0]); dep_tile<TH + 1>(dep, dh[1]);
                dep_tile<TP>(dep, dpre[0]); dep_tile<TP + 1>(dep, dpre[1]);
            }
            STAMP(6)    // layer deposits (+ tail of previous dX)
            __syncthreads();            // deposits visible; this layer's W^T chunk has landed
            STAMP(5)    // barrier wait
            prefetch(IC(cidx + 1));
            if (want_w) {
                own_outer_a<CT / 2>(aWc[i], fb, TH, SL::C, CT, 2 * CT, wave);                     // dWc_i
                if constexpr (i == 0) {
                    own_outer_a<3>(aW0, fb, TP, SL::EMB, 6, 12, wave);
                } else if constexpr (i == 3) {
                    own_outer_a<4>(aW3, fb, TP, SL::EMB, 8, 16, wave);                           // [emb | h2] contiguous
                } else if constexpr (i == 1) {
                    own_outer_a<1>(aW1, fb, TP, TX, 2, 4, wave);
                } else if constexpr (i == 2) {
                    own_outer_a<1>(aW2, fb, TP, TX, 2, 4, wave);
                } else {
                    own_outer_a<1>(aW4, fb, TP, TX, 2, 4, wave);
                }
                own_bias_a(aB[i], fb, (wave < 2 ? TP : TH) + (wave & 1));                        // db_i | dbc_i
                STAMP(7)    // owned dW MFMAs
            }
            if (want_c) lin_lds_swz<2, 2, 32, OCT>(dc, w32s, swd, dh);                    // dC += Wc_i^T dh_i
            if constexpr (i == 0) {
                if (want_r || want_w) lin_lds_swz<6, 2, 32, RO>(demb, w32s, swd, dpre);
            } else if constexpr (i == 3) {
                if (want_r || want_w) lin_lds_swz<6, 2, 32, RO>(demb, w32s, swd, dpre);   // rows 0..95 of W3^T: embedding part
                dh[0] = dh[1] = splat4(0.f);
                lin_lds_swz<2, 2, 32, RO + 96 * 32 * 4>(dh, w32s, swd, dpre);
            } else {
                dh[0] = dh[1] = splat4(0.f);
                lin_lds_swz<2, 2, 32, RO>(dh, w32s, swd, dpre);
            }
        };
        bwd_layer(IC(4)); bwd_layer(IC(3)); bwd_layer(IC(2)); bwd_layer(IC(1)); bwd_layer(IC(0));
        STAMP(8)        // dX chain of the last layer
        // ---- embedding: d_arg = d_emb * cos(arg);  dB^T += d_arg (x) p ;  dp += B d_arg
        float dpx = 0.f, dpy = 0.f, dpz = 0.f;
        if (want_r || want_w) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {                               // cos(arg) recomputed: cheaper than carrying it
                const f32x4 arg = MFMA16(pk[L.oBT() + (16 * t + p) * 4 + q], pc, splat4(0.f));
#pragma unroll
                for (int r = 0; r < 4; ++r) demb[t][r] *= ens_cosf(arg[r]);
            }
            if (want_w) {
                __syncthreads();                                     // every wave is done reading EMB
                dep_tile<SL::EMB + 0>(dep, demb[0]); dep_tile<SL::EMB + 1>(dep, demb[1]); dep_tile<SL::EMB + 2>(dep, demb[2]);
                dep_tile<SL::EMB + 3>(dep, demb[3]); dep_tile<SL::EMB + 4>(dep, demb[4]); dep_tile<SL::EMB + 5>(dep, demb[5]);
                f32x4 pt4 = splat4(0.f);
                if (q == 0) pt4 = f32x4{(float)G.pw[0], (float)G.pw[1], (float)G.pw[2], 0.f};
                dep_tile<SL::Q>(dep, pt4);
                __syncthreads();
                own_outer_a<2>(aBT, fb, SL::EMB, SL::Q, 1, 6, wave);                              // dB^T
            }
            if (want_r) {
                f32x4 dpe[1] = {splat4(0.f)};
                linear_n_swz<1, 6>(dpe, pk + L.oBp(), 96, demb, p, q);
                dpx = dpe[0][0]; dpy = dpe[0][1]; dpz = dpe[0][2];      // valid on q == 0 lanes
            }
        }
        STAMP(9)        // embedding tail (cos recompute, dB^T, dp)
        // ---- grid: coordinate gradient and feature-gradient scatter
        if (want_c) {
            if (want_r) {
                float gx, gy, gz;
                coord_grad_partial(v, grid, q, dc[0], dc[1], gx, gy, gz);
                gx += __shfl_xor(gx, 16); gx += __shfl_xor(gx, 32);
                gy += __shfl_xor(gy, 16); gy += __shfl_xor(gy, 32);
                gz += __shfl_xor(gz, 16); gz += __shfl_xor(gz, 32);
                dpx += gx * v.gx; dpy += gy * v.gy; dpz += gz * v.gz;
            }
            if (want_g && tvalid) {
                // stage dC as [sample][32 channels] in this wave's H1 tiles (layer 1 was their last reader, two
                // barriers ago; the next writer is the next round's layer 3, behind further barriers)
                float* stg = slots + wave * SLOT + SL::H1 * 256;
                *reinterpret_cast<f32x4*>(stg + p * 32 + 4 * q) = dc[0];
                *reinterpret_cast<f32x4*>(stg + p * 32 + 16 + 4 * q) = dc[1];
                wave_lds_fence();
                scatter_tile(stg, v, ggrid, lane);
                wave_lds_fence();
            }
        }
        if (want_r && tvalid) {
            if (q != 0) { dpx = dpy = dpz = 0.f; }
            add_ray_grad(dpx, dpy, dpz, G.zf, G.ray, A.g_ro, A.g_rd, lane);
        }
        STAMP(10)       // coordinate gradient + scatter + ray grads
    }
    STAMP_FLUSH

    // ---- flush: stage the owned tiles into a packed-layout LDS image, then coalesced global atomics
    if (want_w) {
        float* sacc = slots;
        __syncthreads();
        for (int e = threadIdx.x; e < GF; e += 256) sacc[e] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int j = 0; j < CT / 2; ++j) stage_tile(sacc + L.oWc(i), CT * 16, 0, CT, wave + 4 * j, aWc[i][j], 32, CT * 16, p, q);
            stage_bias(sacc + (wave < 2 ? L.ob(i) : L.obc(i)), wave & 1, aB[i], 32, p, q);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) stage_tile(sacc + L.oW(0), 96, 0, 6, wave + 4 * j, aW0[j], 32, 96, p, q);
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_tile(sacc + L.oW(3), 128, 0, 8, wave + 4 * j, aW3[j], 32, 128, p, q);
        stage_tile(sacc + L.oW(1), 32, 0, 2, wave, aW1[0], 32, 32, p, q);
        stage_tile(sacc + L.oW(2), 32, 0, 2, wave, aW2[0], 32, 32, p, q);
        stage_tile(sacc + L.oW(4), 32, 0, 2, wave, aW4[0], 32, 32, p, q);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (wave + 4 * j < 6) stage_tile(sacc + L.oBT(), 4, 0, 1, wave + 4 * j, aBT[j], 93, 3, p, q);
        // output layer: per-lane partial sums -> reduce over the 16 sample lanes; the 4 waves add in LDS
#pragma unroll
        for (int j = 0; j < NE; ++j) {
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                float a = aWo[j][f];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o);
                if (p == 0) lds_add(sacc + L.oWo() + j * 32 + 16 * (f >> 2) + 4 * q + (f & 3), a);
            }
            float bsum = wave_sum(aBo[j]);
            if (lane == 0) lds_add(sacc + L.obo() + j, bsum);
        }
        __syncthreads();
        flush_image(sacc, GF, gpk, A.gpart[kind], wg, n_wg, 256);
    }
}

// Variant that takes the forward activations from the workspace written by the forward kernel instead of
// recomputing them.  The feature-gradient scatter uses the cell records the forward saved (no geometry here); the
// ray gradients (fp64 geometry, corner re-gather, coordinate gradient, ray reduction) are left to grid_bwd_kernel.  Per round and wave: d_raw -> vote -> async fill
// of the LDS slot from the workspace -> output layer (VALU) -> 5 backward layers (W^T chunks from the ring, owned
// dW tiles) -> embedding tail -> dC and the embedding's position gradient handed to grid_bwd_kernel.
template <int CT>
struct XyzSlotsS {                     // slot = workspace block order, then the in-kernel deposits
    static constexpr int EMB = 0, HX2 = 6, HX0 = 8, HX1 = 10, HX3 = 12, C = 14, Q = 14 + CT;
    static constexpr int FILL = 15 + CT;                             // tiles copied from the workspace
    static constexpr int H0 = 15 + CT, H1 = 17 + CT, P0 = 19 + CT, P1 = 21 + CT;
    static constexpr int TILES = 23 + CT;
};
constexpr int RB_SAVED = 96 * 32 + 1024 + 16 * 96 + 96 * 4;           // layer-0 chunk: W0^T | Wc0^T | B (padded) | B^T
// Weight area of the saved-activation kernels (floats).  The W^T | Wc^T chunks of layers 4, 2 and 1 (2048 floats each) are
// RESIDENT: loaded once per workgroup, read by every round without any hand-off.  Layers 3 (5120 floats) and 0 (6016) share
// ONE streaming buffer: layer 0's chunk is requested once every chain wave has read layer 3's (two layers ahead of its use),
// layer 3's at the start of a round.  Same LDS as the former 2-buffer ring (+512 bytes), 2 chunk hand-offs per round
// instead of 5, 44 KB instead of 69 KB of L2 -> LDS traffic per round.
constexpr int WRES_SAVED = 3 * 2048;
constexpr int WAREA_SAVED = WRES_SAVED + RB_SAVED;
constexpr int wres_off(int i) { return i == 4 ? 0 : (i == 2 ? 2048 : (i == 1 ? 4096 : WRES_SAVED)); }   // layer -> float offset

// WW = false: no decoder-parameter gradients (tracker iterations; mapper stages that optimise no decoder): no
// deposits, no owned dW tiles, no accumulators, no slot fill -- under 256 registers, two workgroups per CU.
// SPLIT = true: this is the chain half of decoder_bwd_split_kernel (waves 0..3 of an 8-wave workgroup): it deposits the
// operands but leaves every owned weight-gradient tile to the dW waves (xyz_dw_loop), which run one barrier phase
// behind on the same SIMDs.
// DF = true: the deferred-scatter instantiations (ens_launch_decoder_bwd: the launcher has taken the scatter away, ggrid.data null): dC
// always leaves through the hand-off.  A compile-time switch, and kernels of their own, because the default instantiation's
// register allocation does not survive the run-time form (16 -> 124 spill instructions in the chain waves: 134 -> 146 us).
template <int CT, int NOUT, bool WW, bool SPLIT = false, bool DF = false>
ENS_DEV void xyz_role_saved(const BwdArgs& A, int kind, int wg, int n_wg, float* smem) {
    constexpr XyzLay L{CT * 16};
    constexpr int GF = L.fwd_floats();
    constexpr int NE = NOUT == 4 ? 3 : 1;                          // outputs that carry gradient
    using SL = XyzSlotsS<CT>;
    constexpr int SLOT = WW ? SL::TILES * 256 : 512;               // floats (light variant: the 2-tile scatter staging only)
    constexpr int STG = WW ? SL::H1 * 256 : 0;
    static_assert(RB_SAVED >= 128 * 32 + 1024, "the streaming buffer must hold the largest W^T|Wc^T chunk");
    constexpr int RING_BYTES = WAREA_SAVED * 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = lane & 15, q = lane >> 4;
    const float* __restrict__ pk = A.sc.packed[kind];
    float* gpk = A.gpacked[kind];
    const bool want_w = WW && gpk != nullptr;       // (kept a run-time flag in the WW variant: folding it costs 30 spills)
    const bool want_g = A.ggrid[kind].data != nullptr, want_r = A.g_ro != nullptr;
    const bool want_c = want_g || want_r || DF;
    float* ring = smem;
    float* slots = smem + WAREA_SAVED;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)smem;
    const int slot_idx = kind - 1;                                  // decoder slot in the workspaces

    // backward layer i's chunk (W_i^T | Wc_i^T [| B | B^T for layer 0]) into its place (wres_off)
    auto prefetch = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        float* dst = ring + wres_off(i);
        ring_load(dst, pk + L.oWT(i), 8 * L.K(i), wave, lane);
        ring_load(dst + 32 * L.K(i), pk + L.oWcT(i), 256, wave, lane);
        if constexpr (i == 0) {
            ring_load(dst + 96 * 32 + 1024, pk + L.oBp(), 16 * 96 / 4, wave, lane);
            ring_load(dst + 96 * 32 + 1024 + 16 * 96, pk + L.oBT(), 96, wave, lane);
        }
    };

    f32x4 aWc[5][CT / 2], aW0[3], aW1[1], aW2[1], aW3[4], aW4[1], aBT[2];
    float aB[5];
    float aWo[NE][8], aBo[NE];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        aB[i] = 0.f;
#pragma unroll
        for (int j = 0; j < CT / 2; ++j) aWc[i][j] = splat4(0.f);
    }
    aW0[0] = aW0[1] = aW0[2] = aW1[0] = aW2[0] = aW4[0] = splat4(0.f);
    aW3[0] = aW3[1] = aW3[2] = aW3[3] = splat4(0.f);
    aBT[0] = aBT[1] = splat4(0.f);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        aBo[j] = 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f) aWo[j][f] = 0.f;
    }

    // Wo^T fragments of the output layer's backward: the same two values every round.  Left in the loop they were two global
    // loads per round, each followed by a vmcnt(0) (which also drains the chunk prefetch and the next round's d_raw) in
    // front of the round's first MFMAs.
    float woT[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) woT[rt] = pk[L.oWoT() + (16 * rt + p) * 4 + q];
    const int64_t n_tiles = work_count(A);                          // (work items; all tiles without a work list)
    const int64_t stride = (int64_t)n_wg * 4;
    STAMP_DECL
    STAMP_START
    prefetch(IC(4)); prefetch(IC(2)); prefetch(IC(1));              // resident chunks (SPLIT: the dW waves fill the slots)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // landed before the first round's vote barrier lets anyone read them
    TL(1)
    unsigned round_no = 0;
    unsigned r_exec = 0;                                            // executed (not skipped) rounds so far: targets of the hand-off counters
    // The feature-gradient scatter of a tile is deferred into the next executed round (after its first barrier): the
    // ~30 atomics of a tile then drain under that round's MFMAs instead of in front of its loads (vmcnt is in order).
    float* const stg = slots + wave * SLOT + STG;                   // dC as [sample][32]; H1 is idle until layer 3
    f32x4 rec_prev = splat4(0.f);
    bool pend = false;
    auto scatter_pending = [&]() {
        if (pend) {
            scatter_tile_rec(stg, rec_prev, A.ggrid[kind], lane);
            wave_lds_fence();
            pend = false;
        }
    };
    // d_raw and the ReLU mask words of a tile are fetched one round ahead: their latency (the vote and the first
    // layer wait on them) hides under the previous round
    // workspace layout: full, or (light variant only) the light one
    const bool wl = !WW && A.act_light != 0;
    const int WSS = wl ? ACTL_STRIDE : ACT_STRIDE, WSM = wl ? ACTL_MASK : ACT_MASK, WSV = wl ? ACTL_VOX : ACT_VOX,
              WSQ = wl ? ACTL_Q : SL::Q * 256;
    auto tile_of = [&](int64_t b) {
        const int64_t tr = b + wave;
        return __builtin_amdgcn_readfirstlane((int)work_tile(A, tr < n_tiles ? tr : n_tiles - 1));
    };
    const float dscale = draw_scale_of(A);
    f32x4 draw_n = splat4(0.f);
    uint2 mw_n = make_uint2(0u, 0u);
    if ((int64_t)wg * 4 < n_tiles) {
        const int t0 = tile_of((int64_t)wg * 4);
        draw_n = *reinterpret_cast<const f32x4*>(A.d_raw + ((int64_t)t0 * 16 + p) * 4);
        mw_n = *reinterpret_cast<const uint2*>(A.act_ws + ((int64_t)t0 * ACT_SLOTS + slot_idx) * WSS + WSM + lane * 2);
    }
    for (int64_t base = (int64_t)wg * 4; base < n_tiles; base += stride) {
        const int64_t tile_raw = base + wave;
        const bool tvalid = tile_raw < n_tiles;
        const int tile = tile_of(base);
        f32x4 draw = draw_n * dscale;
        const uint2 mw = mw_n;
        if (base + stride < n_tiles) {
            const int t1 = tile_of(base + stride);
            draw_n = *reinterpret_cast<const f32x4*>(A.d_raw + ((int64_t)t1 * 16 + p) * 4);
            mw_n = *reinterpret_cast<const uint2*>(A.act_ws + ((int64_t)t1 * ACT_SLOTS + slot_idx) * WSS + WSM + lane * 2);
        }
        if (!tvalid) draw = splat4(0.f);
        float dj[NE];                                               // d(loss)/d(output j) of this lane's sample
        if constexpr (NOUT == 4) { dj[0] = draw[0]; dj[1] = draw[1]; dj[2] = draw[2]; } else { dj[0] = draw[3]; }
        bool nz = false;
#pragma unroll
        for (int j = 0; j < NE; ++j) nz = nz || dj[j] != 0.f;
        // ray gradients: handed off per tile to the ray-gradient launch (dgrid_ws given), or -- dgrid_ws NULL -- computed here at
        // the end of the round from the registers that would have been handed off (no 3 KB store + load per tile and decoder,
        // no ray-gradient role in the finish launch)
        const bool handoff = (DF || want_r) && A.dgrid_ws != nullptr;
        float* dgw = handoff ? A.dgrid_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * DG_STRIDE : nullptr;
        {   // skip the round when nothing flows into any of its 4 tiles (one barrier; waves stay in lockstep)
            const int par = (int)(round_no & 1);
            if (lane == 0) ens_vote[par][wave] = __any(nz) ? 1 : 0;
            ++round_no;
            if constexpr (SPLIT) wg_barrier_lds(); else __syncthreads();
            const int any4 = ens_vote[par][0] | ens_vote[par][1] | ens_vote[par][2] | ens_vote[par][3];
            if (!any4) {
                if (handoff && tvalid) {                            // grid_bwd_kernel reads the hand-off of every tile
                    *reinterpret_cast<f32x4*>(dgw + lane * 4) = splat4(0.f);
                    *reinterpret_cast<f32x4*>(dgw + 256 + lane * 4) = splat4(0.f);
                    *reinterpret_cast<f32x4*>(dgw + DG_DPE + lane * 4) = splat4(0.f);
                }
                scatter_pending();
                continue;
            }
        }
        STAMP(0)        // d_raw load + vote barrier
        // layer 3's chunk into the streaming buffer: every wave is past the previous round's tail (its last reader) -- the
        // vote barrier says so -- and layer 4 runs on resident weights while it lands
        prefetch(IC(3));

        unsigned dep[4];
        dep_bases(dep, lds0 + RING_BYTES + wave * SLOT * 4, p, q);
        unsigned fb[4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) { fb[sl] = lds0 + RING_BYTES + sl * SLOT * 4 + frag_lane_off(lane); opaque(fb[sl]); }

        // ---- forward activations from the workspace: deposit tiles straight into this wave's LDS slot (async),
        //      h4 and the ReLU masks into registers
        const float* __restrict__ wsb = A.act_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * WSS;
        f32x4 h4[2] = {splat4(0.f), splat4(0.f)};
        if constexpr (WW) {
            if constexpr (!SPLIT) {
                float* myslot = slots + wave * SLOT;
#pragma unroll
                for (int t = 0; t < SL::FILL; ++t)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsb + t * 256 + lane * 4),
                                                     (__attribute__((address_space(3))) void*)(myslot + t * 256), 16, 0, 0);
            }
            h4[0] = ld4(wsb + ACT_H4 + lane * 4);
            h4[1] = ld4(wsb + ACT_H4 + 256 + lane * 4);
        }
        f32x4 rec = splat4(0.f);
        if constexpr (!SPLIT) { if (want_g) rec = ld4(wsb + WSV + p * 4); }
        unsigned mbits[5] = {mw.x & 255u, (mw.x >> 8) & 255u, (mw.x >> 16) & 255u, (mw.x >> 24) & 255u, mw.y & 255u};
        unsigned wsw = swz_base_even(lds0, 32, p, q);                  // swizzled-image lane base, ring buffer 0
        const unsigned swd = swz_odd_delta(p);
        opaque(wsw);
        STAMP(3)

        // ---- output layer: dh4 = Wo^T d_out (dWo, dbo follow after the first barrier, when h4 has arrived)
        f32x4 dh[2] = {splat4(0.f), splat4(0.f)};
        {   // K = 4: one MFMA step per row tile; k-slot q carries output q
            const float dq = NOUT == 4 ? (q == 0 ? draw[0] : (q == 1 ? draw[1] : (q == 2 ? draw[2] : 0.f)))
                                       : (q == 0 ? draw[3] : 0.f);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) dh[rt] = MFMA16(woT[rt], dq, dh[rt]);
        }
        f32x4 dc[2] = {splat4(0.f), splat4(0.f)};
        f32x4 demb[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) demb[t] = splat4(0.f);
        unsigned ring0 = 0;                                         // byte address of the layer-0 ring buffer (tail)
        auto bwd_layer = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const unsigned wb = wsw + wres_off(i) * 4;
            constexpr int OCT = 32 * L.K(i) * 4;                      // W_i^T [K][32] | Wc_i^T [32][32]
            constexpr int TH = (i & 1) ? SL::H1 : SL::H0, TP = (i & 1) ? SL::P1 : SL::P0;
            constexpr int TX = i == 4 ? SL::HX3 : (i == 2 ? SL::HX1 : SL::HX0);      // input h_{i-1} of layers 4, 2, 1
            f32x4 dpre[2] = {mask4(dh[0], mbits[i], 0), mask4(dh[1], mbits[i], 4)};
            STAMP(2)    // (stamps build: the previous layer's dX results have arrived)
            if constexpr (SPLIT) {
                // the dh / dpre tiles of this parity still hold layer i+2's operands: its owned products must have read them
                if constexpr (i <= 2) sy_wait(SY_DW + i + 2, 4 * (int)(r_exec + 1));
                // ... and the H1 tiles the previous tile's dC was staged in: the dW waves must have taken it
                if constexpr (i == 3) { if (want_g) sy_wait(SY_STGDONE, 4 * (int)r_exec); }
            }
            STAMP(4)    // (stamps build: waits for the dW waves' reads)
            if (want_w) {
                dep_tile<TH>(dep, dh[0]); dep_tile<TH + 1>(dep, dh[1]);
                dep_tile<TP>(dep, dpre[0]); dep_tile<TP + 1>(dep, dpre[1]);
            }
            STAMP(6)    // layer deposits (+ tail of previous dX)
            if constexpr (SPLIT) {
                sy_signal(SY_DEP + i, lane);                                  // layer i's operands are in place
                // streamed chunks only (two per executed round): all four quarters have landed
                if constexpr (i == 3) sy_wait(SY_RING, 4 * (int)(2 * r_exec + 1));
                if constexpr (i == 0) sy_wait(SY_RING, 4 * (int)(2 * r_exec + 2));
            } else {
                __syncthreads();        // deposits + slot fill visible; a streamed chunk requested before the previous barrier has landed
                if constexpr (i == 2) prefetch(IC(0));   // every wave is past layer 3: its chunk gives way to layer 0's
            }
            STAMP(5)    // barrier wait
            if constexpr (i == 4 && !SPLIT) scatter_pending();       // previous tile's atomics, behind this round's loads
            if constexpr (i == 4) { STAMP(1) }      // deferred scatter
            if constexpr (i == 4) {
                if (want_w) {                       // dWo, dbo on the VALU (n_out <= 4 rows: not worth an MFMA tile)
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) aWo[j][4 * t + r] = fmaf(dj[j], h4[t][r], aWo[j][4 * t + r]);
                        }
                        if (q == 0) aBo[j] += dj[j];
                    }
                }
            }
            if (want_w) {
                const int ybias = (wave < 2 ? TP : TH) + (wave & 1);                              // db_i | dbc_i
                if constexpr (SPLIT) {
                    // the dW waves do it
                } else if constexpr (i == 0) {                                                    // dWc_i, dW_i, bias
                    own_layer_a<CT / 2, 3>(aWc[i], TH, SL::C, CT, 2 * CT, aW0, TP, SL::EMB, 6, 12, aB[i], ybias, fb, wave);
                } else if constexpr (i == 3) {
                    own_layer_a<CT / 2, 4>(aWc[i], TH, SL::C, CT, 2 * CT, aW3, TP, SL::EMB, 8, 16, aB[i], ybias, fb, wave);   // [emb | h2] contiguous
                } else if constexpr (i == 1) {
                    own_layer_a<CT / 2, 1>(aWc[i], TH, SL::C, CT, 2 * CT, aW1, TP, TX, 2, 4, aB[i], ybias, fb, wave);
                } else if constexpr (i == 2) {
                    own_layer_a<CT / 2, 1>(aWc[i], TH, SL::C, CT, 2 * CT, aW2, TP, TX, 2, 4, aB[i], ybias, fb, wave);
                } else {
                    own_layer_a<CT / 2, 1>(aWc[i], TH, SL::C, CT, 2 * CT, aW4, TP, TX, 2, 4, aB[i], ybias, fb, wave);
                }
                STAMP(7)    // owned dW MFMAs
            }
            if (want_c) lin_lds_swz<2, 2, 32, OCT>(dc, wb, swd, dh);                      // dC += Wc_i^T dh_i
            if constexpr (i == 0) {
                ring0 = lds0 + WRES_SAVED * 4;
                if (want_r || want_w) lin_lds_swz<6, 2, 32, 0>(demb, wb, swd, dpre);
            } else if constexpr (i == 3) {
                if (want_r || want_w) lin_lds_swz<6, 2, 32, 0>(demb, wb, swd, dpre);  // rows 0..95 of W3^T: embedding part
                dh[0] = dh[1] = splat4(0.f);
                lin_lds_swz<2, 2, 32, 96 * 32 * 4>(dh, wb, swd, dpre);
            } else {
                dh[0] = dh[1] = splat4(0.f);
                lin_lds_swz<2, 2, 32, 0>(dh, wb, swd, dpre);
            }
            if constexpr (SPLIT) {
                if constexpr (i == 4 || i == 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's quarter of the streamed chunk requested a layer ago has landed
                    sy_signal(SY_RING, lane);
                }
                if constexpr (i == 3) sy_signal(SY_RDONE, lane);     // layer 3's chunk read (lgkmcnt(0) inside)
                if constexpr (i == 2) {                              // ... by every chain wave: layer 0's chunk may replace it
                    sy_wait(SY_RDONE, 4 * (int)(2 * r_exec + 1));
                    prefetch(IC(0));
                }
            }
        };
        bwd_layer(IC(4)); bwd_layer(IC(3)); bwd_layer(IC(2)); bwd_layer(IC(1)); bwd_layer(IC(0));
        STAMP(8)        // dX chain of the last layer
        // ---- embedding: d_arg = d_emb * cos(arg);  dB^T += d_arg (x) p ;  dp += B d_arg.  B and B^T ride in the
        //      layer-0 ring chunk; the sample coordinates come from the XYZ tile of the slot.
        f32x4 dpe[1] = {splat4(0.f)};
        if (want_r || want_w) {
            constexpr int OBP = (96 * 32 + 1024) * 4, OBT = OBP + 16 * 96 * 4;
            // coordinate q of sample p from the XYZ tile (feature i = q)
            float pc = 0.f;
            if constexpr (SPLIT) sy_wait(SY_FILL, 4 * (int)(r_exec + 1));       // (the slot fill landed long ago)
            if constexpr (WW) {
                if (q < 3) pc = *reinterpret_cast<const lds_float*>(static_cast<uintptr_t>(
                                    lds0 + RING_BYTES + (wave * SLOT + SL::Q * 256 + (p >> 2) * 64 + (q ^ (p >> 2)) * 4 + (p & 3)) * 4));
            } else {
                if (q < 3) pc = wsb[WSQ + (p >> 2) * 64 + (q ^ (p >> 2)) * 4 + (p & 3)];  // same tile (swizzled chunk), straight from the workspace
            }
#pragma unroll
            for (int t = 0; t < 6; ++t) {                               // cos(arg) recomputed: cheaper than carrying it
                const float a = *reinterpret_cast<const lds_float*>(static_cast<uintptr_t>(ring0 + OBT + ((16 * t + p) * 4 + q) * 4));
                const f32x4 arg = MFMA16(a, pc, splat4(0.f));
#pragma unroll
                for (int r = 0; r < 4; ++r) demb[t][r] *= ens_cosf(arg[r]);
            }
            if (want_w) {
                if constexpr (SPLIT) {
                    // d_arg goes into the six tiles of h2 | h0 | h1 (contiguous, tiles 6..11): their last readers are the
                    // owned products of layers 3, 1 and 2 -- no need to wait for layer 0's, which still read EMB.  The same
                    // wait frees the H1 tiles, where this wave stages dC for the deferred scatter below.
                    sy_wait(SY_DW + 1, 4 * (int)(r_exec + 1));
                    dep_tile<SL::HX2 + 0>(dep, demb[0]); dep_tile<SL::HX2 + 1>(dep, demb[1]); dep_tile<SL::HX2 + 2>(dep, demb[2]);
                    dep_tile<SL::HX2 + 3>(dep, demb[3]); dep_tile<SL::HX2 + 4>(dep, demb[4]); dep_tile<SL::HX2 + 5>(dep, demb[5]);
                    sy_signal(SY_DARG, lane);
                } else {
                    __syncthreads();                                 // every wave is done reading EMB
                    dep_tile<SL::EMB + 0>(dep, demb[0]); dep_tile<SL::EMB + 1>(dep, demb[1]); dep_tile<SL::EMB + 2>(dep, demb[2]);
                    dep_tile<SL::EMB + 3>(dep, demb[3]); dep_tile<SL::EMB + 4>(dep, demb[4]); dep_tile<SL::EMB + 5>(dep, demb[5]);
                    __syncthreads();
                    own_outer_a<2>(aBT, fb, SL::EMB, SL::Q, 1, 6, wave);                             // dB^T
                }
            }
            if (want_r) lin_lds_swz<1, 6, 96, OBP>(dpe, swz_base_even(ring0, 96, p, q), swd, demb);   // rows 0..2: dp (q == 0 lanes)
        }
        if constexpr (SPLIT) sy_signal(SY_RDONE, lane);             // the layer-0 chunk (B, B^T ride in it) is no longer needed
        STAMP(9)        // embedding tail (cos recompute, dB^T, dp)
        if (handoff && tvalid) {        // hand-off to grid_bwd_kernel: dC (register layout) + embedding's position gradient
            *reinterpret_cast<f32x4*>(dgw + lane * 4) = dc[0];
            *reinterpret_cast<f32x4*>(dgw + 256 + lane * 4) = dc[1];
            *reinterpret_cast<f32x4*>(dgw + DG_DPE + lane * 4) = dpe[0];
        }
        if constexpr (SPLIT) {
            if (want_g) {               // dC of this tile (zeros for a padding tile) -> [sample][32] in this wave's H1 tiles, for dW wave `wave`
                *reinterpret_cast<f32x4*>(stg + p * 32 + 4 * q) = dc[0];
                *reinterpret_cast<f32x4*>(stg + p * 32 + 16 + 4 * q) = dc[1];
                sy_signal(SY_STG, lane);
            }
        } else if (want_g && tvalid) {
            // dC of this tile -> [sample][32] in this wave's H1 tiles (last read before the layer-0 barrier); the
            // 256-byte atomics per cell corner row are issued by scatter_pending() in the next round
            *reinterpret_cast<f32x4*>(stg + p * 32 + 4 * q) = dc[0];
            *reinterpret_cast<f32x4*>(stg + p * 32 + 16 + 4 * q) = dc[1];
            wave_lds_fence();
            rec_prev = rec;
            pend = true;
        }
        if (want_r && !handoff && tvalid) {                         // ray_grad_unit (raygrad.hpp) on this tile, in place
            const TileGeo G = tile_geo((int64_t)tile, A.ntl, 16 * A.ntl, A.ro, A.rd, A.z, p);
            const Vox v = make_vox(G.pw, A.sc.lo, A.sc.hi, A.sc.grid[kind]);
            float gx, gy, gz;
            coord_grad_partial(v, A.sc.grid[kind], q, dc[0], dc[1], gx, gy, gz);
            gx += __shfl_xor(gx, 16); gx += __shfl_xor(gx, 32);
            gy += __shfl_xor(gy, 16); gy += __shfl_xor(gy, 32);
            gz += __shfl_xor(gz, 16); gz += __shfl_xor(gz, 32);
            float dpx = gx * v.gx + dpe[0][0], dpy = gy * v.gy + dpe[0][1], dpz = gz * v.gz + dpe[0][2];
            if (q != 0) { dpx = dpy = dpz = 0.f; }
            add_ray_grad(dpx, dpy, dpz, G.zf, G.ray, A.g_ro, A.g_rd, lane);
        }
        TL(2 + (r_exec < 7 ? r_exec : 7))
        ++r_exec;
        STAMP(10)
    }
    if constexpr (!SPLIT) scatter_pending();
    STAMP_FLUSH
    TL(10)

    // ---- flush: stage the owned tiles into a packed-layout LDS image, then coalesced global atomics
    if (want_w) {
        float* sacc = slots;
        const int nthr = SPLIT ? 512 : 256;
        // The three barriers of the flush order LDS traffic only (wg_barrier_lds): a __syncthreads() here made every wave wait
        // for the acknowledges of the float atomics of its last scatter (~3 us, tools/stamps_bwd.py budget).  With partial
        // images only the output layer's rows are summed in LDS (lds_add) and need clearing; the padding of the image is never
        // read by the finish launch.
        const bool part = A.gpart[kind] != nullptr;
        wg_barrier_lds();
        if (part) { for (int e = L.oWo() + threadIdx.x; e < GF; e += nthr) sacc[e] = 0.f; }
        else { for (int e = threadIdx.x; e < GF; e += nthr) sacc[e] = 0.f; }
        wg_barrier_lds();
        if constexpr (!SPLIT) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#pragma unroll
                for (int j = 0; j < CT / 2; ++j) stage_tile(sacc + L.oWc(i), CT * 16, 0, CT, wave + 4 * j, aWc[i][j], 32, CT * 16, p, q);
                stage_bias_lane(sacc + (wave < 2 ? L.ob(i) : L.obc(i)), wave & 1, aB[i], 32, p, q);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) stage_tile(sacc + L.oW(0), 96, 0, 6, wave + 4 * j, aW0[j], 32, 96, p, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) stage_tile(sacc + L.oW(3), 128, 0, 8, wave + 4 * j, aW3[j], 32, 128, p, q);
            stage_tile(sacc + L.oW(1), 32, 0, 2, wave, aW1[0], 32, 32, p, q);
            stage_tile(sacc + L.oW(2), 32, 0, 2, wave, aW2[0], 32, 32, p, q);
            stage_tile(sacc + L.oW(4), 32, 0, 2, wave, aW4[0], 32, 32, p, q);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (wave + 4 * j < 6) stage_tile(sacc + L.oBT(), 4, 0, 1, wave + 4 * j, aBT[j], 93, 3, p, q);
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                float a = aWo[j][f];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o);
                if (p == 0) lds_add(sacc + L.oWo() + j * 32 + 16 * (f >> 2) + 4 * q + (f & 3), a);
            }
            float bsum = wave_sum(aBo[j]);
            if (lane == 0) lds_add(sacc + L.obo() + j, bsum);
        }
        wg_barrier_lds();
        flush_image(sacc, GF, gpk, A.gpart[kind], wg, n_wg, nthr);
    }
    TL(12)
}

// The dW half of decoder_bwd_split_kernel (waves 4..7): the owned weight-gradient tiles of xyz_role_saved, one barrier
// phase behind the chain waves that deposit their operands -- its MFMAs fill the gaps of the chain waves' latencies on
// the same SIMDs.  Executes exactly the barriers of the chain half: vote, 5 layers, 2 in the tail, 3 in the flush.
template <int CT, int NOUT>
ENS_DEV void xyz_dw_loop(const BwdArgs& A, int kind, int wg, int n_wg, float* smem) {
    constexpr XyzLay L{CT * 16};
    constexpr int GF = L.fwd_floats();
    using SL = XyzSlotsS<CT>;
    constexpr int SLOT = SL::TILES * 256;
    constexpr int RING_BYTES = WAREA_SAVED * 4;
    const int lane = threadIdx.x & 63, ow = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) - 4, p = lane & 15, q = lane >> 4;
    float* gpk = A.gpacked[kind];
    float* slots = smem + WAREA_SAVED;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)smem;
    f32x4 aWc[5][CT / 2], aW0[3], aW1[1], aW2[1], aW3[4], aW4[1], aBT[2];
    float aB[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        aB[i] = 0.f;
#pragma unroll
        for (int j = 0; j < CT / 2; ++j) aWc[i][j] = splat4(0.f);
    }
    aW0[0] = aW0[1] = aW0[2] = aW1[0] = aW2[0] = aW4[0] = splat4(0.f);
    aW3[0] = aW3[1] = aW3[2] = aW3[3] = splat4(0.f);
    aBT[0] = aBT[1] = splat4(0.f);
    unsigned fb[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) { fb[sl] = lds0 + RING_BYTES + sl * SLOT * 4 + frag_lane_off(lane); opaque(fb[sl]); }
    const int64_t n_tiles = work_count(A);
    const int64_t stride = (int64_t)n_wg * 4;
    unsigned round_no = 0;
    // These waves also fill the slots from the activation workspace (the chain waves stream the W^T ring themselves) and run
    // the feature-gradient scatter of the chain waves' tiles.
    unsigned r_exec = 0;
    // ---- the feature-gradient scatter of chain wave `ow`'s tiles runs here (see the hand-off counters): state across pieces
    const DevGrid ggrid = A.ggrid[kind];
    const bool want_g = ggrid.data != nullptr;
    float sval[16];
    int sri = 0, srx = 0, sry = 0, srz = 0;
    ScatterSt sst;
    sst.acc[0] = sst.acc[1] = sst.acc[2] = sst.acc[3] = 0.f; sst.cur = 0u; sst.open = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) sval[i] = 0.f;
    int64_t tile_prev = 0;
    const unsigned stg_rd = lds0 + RING_BYTES + (ow * SLOT + SL::H1 * 256 + (lane & 31)) * 4;     // dC[pt][ch]: + pt * 128 bytes
    // take the dC chain wave `ow` staged for its tile of the previous executed round (and that tile's cell records)
    auto take_staged = [&](int rounds_done) {
        sy_wait(SY_STG, 4 * rounds_done);
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) sval[pt] = *reinterpret_cast<const lds_float*>(static_cast<uintptr_t>(stg_rd + pt * 128));
        const f32x4 rec = ld4(A.act_ws + (tile_prev * ACT_SLOTS + (kind - 1)) * ACT_STRIDE + ACT_VOX + (lane & 15) * 4);
        const float r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
        sri = __builtin_bit_cast(int, r0); srx = __builtin_bit_cast(int, r1); sry = __builtin_bit_cast(int, r2); srz = __builtin_bit_cast(int, r3);
        sst.acc[0] = sst.acc[1] = sst.acc[2] = sst.acc[3] = 0.f; sst.cur = 0u; sst.open = false;
    };
    STAMP_DECL
    STAMP_START
    TL(1)
    auto own = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int TH = (i & 1) ? SL::H1 : SL::H0, TP = (i & 1) ? SL::P1 : SL::P0;
        constexpr int TX = i == 4 ? SL::HX3 : (i == 2 ? SL::HX1 : SL::HX0);
        sy_wait(SY_DEP + i, 4 * (int)(r_exec + 1));                 // layer i's deposits are in place
        STAMP(2)        // (stamps build) wait for the chain waves' deposits
        if constexpr (i == 4) {
            // layer 4 reads h3 and the grid features only: they were requested first, and vmcnt counts in issue order -- wait
            // until just the 13 later tiles (embedding, h0..h2, coordinates) are still in flight
            asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            sy_signal(SY_FILLA, lane);
            sy_wait(SY_FILLA, 4 * (int)(r_exec + 1));               // ... in all four slots
            STAMP(3)    // slot fill landed
        }
        if constexpr (i == 3) sy_wait(SY_FILL, 4 * (int)(r_exec + 1));    // the rest of all four slots
#ifdef ENS_EXP_NO_DW
        sy_signal(SY_DW + i, lane);
        return;                             // timing experiment (wrong results): no owned-tile MFMAs
#endif
        const int ybias = (ow < 2 ? TP : TH) + (ow & 1);
        if constexpr (i == 0) {
            own_layer_a<CT / 2, 3>(aWc[i], TH, SL::C, CT, 2 * CT, aW0, TP, SL::EMB, 6, 12, aB[i], ybias, fb, ow);
        } else if constexpr (i == 3) {
            own_layer_a<CT / 2, 4>(aWc[i], TH, SL::C, CT, 2 * CT, aW3, TP, SL::EMB, 8, 16, aB[i], ybias, fb, ow);
        } else if constexpr (i == 1) {
            own_layer_a<CT / 2, 1>(aWc[i], TH, SL::C, CT, 2 * CT, aW1, TP, TX, 2, 4, aB[i], ybias, fb, ow);
        } else if constexpr (i == 2) {
            own_layer_a<CT / 2, 1>(aWc[i], TH, SL::C, CT, 2 * CT, aW2, TP, TX, 2, 4, aB[i], ybias, fb, ow);
        } else {
            own_layer_a<CT / 2, 1>(aWc[i], TH, SL::C, CT, 2 * CT, aW4, TP, TX, 2, 4, aB[i], ybias, fb, ow);
        }
        sy_signal(SY_DW + i, lane);                                  // (the operand reads have returned: lgkmcnt(0) inside)
        if constexpr (i == 4) {              // the rest of this wave's fill has had a layer's time to land; nothing younger is in flight yet
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sy_signal(SY_FILL, lane);
        }
        STAMP(4)        // owned products
        // one 4-sample piece of the previous tile's scatter behind each of layers 4..1: the chain waves are not waiting for it
        if (want_g && r_exec > 0) {
            if constexpr (i == 4) {
                take_staged((int)r_exec);
                sy_signal(SY_STGDONE, lane);                         // the H1 tiles may take layer 3's deposits
                scatter_piece<0>(sst, sval, sri, srx, sry, srz, ggrid, lane);
            } else if constexpr (i == 3) {
                scatter_piece<4>(sst, sval, sri, srx, sry, srz, ggrid, lane);
            } else if constexpr (i == 2) {
                scatter_piece<8>(sst, sval, sri, srx, sry, srz, ggrid, lane);
            } else if constexpr (i == 1) {
                scatter_piece<12>(sst, sval, sri, srx, sry, srz, ggrid, lane);
                if (sst.open) scatter_flush(sst, ggrid, lane, 15, 0);
            }
        }
        STAMP(5)        // scatter piece
    };
    for (int64_t base = (int64_t)wg * 4; base < n_tiles; base += stride) {
        const int par = (int)(round_no & 1);
        ++round_no;
        wg_barrier_lds();                   // vote (the one workgroup barrier of a round)
        const int any4 = ens_vote[par][0] | ens_vote[par][1] | ens_vote[par][2] | ens_vote[par][3];
        STAMP(0)        // vote barrier
        if (!any4) continue;
        const int64_t tr = base + ow;
        const int64_t tile = work_tile(A, tr < n_tiles ? tr : n_tiles - 1);
        {   // slot `ow` <- the operands the forward parked for chain wave ow's tile of this round
            const float* __restrict__ wsb = A.act_ws + (tile * ACT_SLOTS + (kind - 1)) * ACT_STRIDE;
            float* myslot = slots + ow * SLOT;
            static_assert(SL::HX3 == 12 && SL::C == 14 && SL::FILL - (2 + CT) == 13, "fill order / vmcnt(13) in own(4)");
            auto fill_tile = [&](int t) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsb + t * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void*)(myslot + t * 256), 16, 0, 0);
            };
#pragma unroll
            for (int t = SL::HX3; t < SL::C + CT; ++t) fill_tile(t);        // h3 | grid features: layer 4's operands first
#pragma unroll
            for (int t = 0; t < SL::HX3; ++t) fill_tile(t);                 // embedding, h2, h0, h1
            fill_tile(SL::Q);                                               // coordinates
        }
        STAMP(1)        // fill issue
        own(IC(4)); own(IC(3)); own(IC(2)); own(IC(1)); own(IC(0));
        sy_wait(SY_DARG, 4 * (int)(r_exec + 1));                     // the chain waves have deposited d_arg (tiles HX2..HX1)
        STAMP(6)        // wait for d_arg
        own_outer_a<2>(aBT, fb, SL::HX2, SL::Q, 1, 6, ow);                                         // dB^T
        tile_prev = tile;
        TL(2 + (r_exec < 7 ? r_exec : 7))
        ++r_exec;
        STAMP(7)        // dB^T
    }
    TL(10)
#ifdef ENS_STAMPS
    if (g_stamp_buf && lane == 0) {         // dW waves: second half of the stamp buffer (s_memrealtime span in the last slot)
        unsigned long long rt1_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_)::"memory");
        st_acc[ENS_NSEG - 1] = rt1_ - st_rt0;
        for (int k_ = 0; k_ < ENS_NSEG; ++k_) g_stamp_buf[((size_t)(gridDim.x + blockIdx.x) * 4 + ow) * ENS_NSEG + k_] = st_acc[k_];
    }
#endif
    if (want_g && r_exec > 0) {                                       // the last executed round's tile
        take_staged((int)r_exec);
        scatter_piece<0>(sst, sval, sri, srx, sry, srz, ggrid, lane);
        scatter_piece<4>(sst, sval, sri, srx, sry, srz, ggrid, lane);
        scatter_piece<8>(sst, sval, sri, srx, sry, srz, ggrid, lane);
        scatter_piece<12>(sst, sval, sri, srx, sry, srz, ggrid, lane);
        if (sst.open) scatter_flush(sst, ggrid, lane, 15, 0);
    }
    TL(11)
    // (the barriers of the flush order LDS traffic only: the atomics just issued drain behind them, see xyz_role_saved)
    float* sacc = slots;
    const bool part = A.gpart[kind] != nullptr;
    wg_barrier_lds();
    if (part) { for (int e = L.oWo() + threadIdx.x; e < GF; e += 512) sacc[e] = 0.f; }
    else { for (int e = threadIdx.x; e < GF; e += 512) sacc[e] = 0.f; }
    wg_barrier_lds();
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = 0; j < CT / 2; ++j) stage_tile(sacc + L.oWc(i), CT * 16, 0, CT, ow + 4 * j, aWc[i][j], 32, CT * 16, p, q);
        stage_bias_lane(sacc + (ow < 2 ? L.ob(i) : L.obc(i)), ow & 1, aB[i], 32, p, q);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) stage_tile(sacc + L.oW(0), 96, 0, 6, ow + 4 * j, aW0[j], 32, 96, p, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_tile(sacc + L.oW(3), 128, 0, 8, ow + 4 * j, aW3[j], 32, 128, p, q);
    stage_tile(sacc + L.oW(1), 32, 0, 2, ow, aW1[0], 32, 32, p, q);
    stage_tile(sacc + L.oW(2), 32, 0, 2, ow, aW2[0], 32, 32, p, q);
    stage_tile(sacc + L.oW(4), 32, 0, 2, ow, aW4[0], 32, 32, p, q);
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (ow + 4 * j < 6) stage_tile(sacc + L.oBT(), 4, 0, 1, ow + 4 * j, aBT[j], 93, 3, p, q);
    wg_barrier_lds();
    flush_image(sacc, GF, gpk, A.gpart[kind], wg, n_wg, 512);
    TL(12)
}

// Ray-gradient side of the backward for the saved-activation path: one wave per (16-sample tile, decoder slot)
// (ray_grad_unit, raygrad.hpp).  Normally this work rides in the finish launch (step_kernel); the stand-alone kernel
// serves enslam_ray_grad_bwd.
__global__ __launch_bounds__(64) void grid_bwd_kernel(RayGradArgs A) { ray_grad_unit(A, (int64_t)blockIdx.x, (int)threadIdx.x); }

// ------------------------------------------------------------------------------------------------
// MLP_no_xyz (coarse) backward role
// ------------------------------------------------------------------------------------------------
struct FeatSlots {
    static constexpr int C = 0;        // 2: grid features; followed by X1 so that W3's input [c|h2] is contiguous
    static constexpr int X1 = 2;       // 2: h2 / h0 / h4
    static constexpr int X0 = 4;       // 2: h3 / h1
    static constexpr int P0 = 6, P1 = 8;
    static constexpr int Q = 10;
    static constexpr int TILES = 11;
};

ENS_DEV void feat_role(const BwdArgs& A, int wg, int n_wg, float* smem) {
    constexpr FeatLay L{};
    constexpr int GF = L.fwd_floats();
    using SL = FeatSlots;
    constexpr int SLOT = SL::TILES * 256;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = lane & 15, q = lane >> 4;
    const float* __restrict__ pk = A.sc.packed[0];
    const DevGrid grid = A.sc.grid[0];
    const DevGrid ggrid = A.ggrid[0];
    float* gpk = A.gpacked[0];
    const bool want_w = gpk != nullptr, want_g = ggrid.data != nullptr, want_r = A.g_ro != nullptr;
    float* my = smem + wave * SLOT;
    f32x4 aW[5][2], aB[5], aWo[1], aBo;
#pragma unroll
    for (int i = 0; i < 5; ++i) { aW[i][0] = aW[i][1] = aB[i] = splat4(0.f); }
    aWo[0] = aBo = splat4(0.f);

    const int64_t n_tiles = (int64_t)A.n_rays * A.ntl;
    const int S = 16 * A.ntl;
    const int64_t stride = (int64_t)n_wg * 4;
    for (int64_t base = (int64_t)wg * 4; base < n_tiles; base += stride) {
        const int64_t tile_raw = base + wave;
        const bool tvalid = tile_raw < n_tiles;
        const int tile = __builtin_amdgcn_readfirstlane((int)(tvalid ? tile_raw : n_tiles - 1));
        const TileGeo G = tile_geo(tile, A.ntl, S, A.ro, A.rd, A.z, p);
        f32x4 draw = *reinterpret_cast<const f32x4*>(A.d_raw + G.sidx * 4) * draw_scale_of(A);
        if (!tvalid) draw = splat4(0.f);
        f32x4 dout = splat4(0.f);
        if (q == 0) dout[0] = draw[3];
        const int active = __any(dout[0] != 0.f) ? 1 : 0;
        if (want_w ? !__syncthreads_or(active) : !active) continue;
        const Vox v = make_vox(G.pw, A.sc.clo, A.sc.chi, grid);
        f32x4 c[1][2];
        gather8(v, grid, q, c[0][0], c[0][1]);
        f32x4 h[5][1][2];
        unsigned mbits[5];
        auto fwd_layer = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            f32x4 acc[1][2];
            acc[0][0] = ld4(pk + L.ob(i) + 4 * q);
            acc[0][1] = ld4(pk + L.ob(i) + 16 + 4 * q);
            if constexpr (i == 0) {
                linear32<2, 1, 2>(acc, pk + L.oW(0), 32, c, 0, p, q);
            } else if constexpr (i == 3) {
                linear32<2, 1, 2>(acc, pk + L.oW(3), 64, c, 0, p, q);
                linear32<2, 1, 2>(acc, pk + L.oW(3) + 32, 64, h[2], 0, p, q);
            } else {
                linear32<2, 1, 2>(acc, pk + L.oW(i), 32, h[i - 1], 0, p, q);
            }
            mbits[i] = pos_bits(acc[0][0]) | (pos_bits(acc[0][1]) << 4);
            h[i][0][0] = relu4(acc[0][0]);
            h[i][0][1] = relu4(acc[0][1]);
        };
        fwd_layer(IC(0)); fwd_layer(IC(1)); fwd_layer(IC(2)); fwd_layer(IC(3)); fwd_layer(IC(4));
        if (want_w) {
            deposit(my, SL::C, c[0][0], p, q); deposit(my, SL::C + 1, c[0][1], p, q);
            deposit(my, SL::Q, dout, p, q);
            deposit(my, SL::X1, h[4][0][0], p, q); deposit(my, SL::X1 + 1, h[4][0][1], p, q);
            __syncthreads();
            own_outer<1>(aWo, smem, SLOT, SL::Q, SL::X1, 2, 2, wave, lane);
            if (wave == 2) own_bias(aBo, smem, SLOT, SL::Q, lane);
        }
        f32x4 dh[2] = {splat4(0.f), splat4(0.f)};
        {
            const float dq = q == 0 ? draw[3] : 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) dh[rt] = MFMA16(pk[L.oWoT() + (16 * rt + p) * 4 + q], dq, dh[rt]);
        }
        f32x4 dc[2] = {splat4(0.f), splat4(0.f)};
        auto bwd_layer = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int TP = (i & 1) ? SL::P1 : SL::P0, TX = (i & 1) ? SL::X1 : SL::X0;
            f32x4 dpre[2] = {mask4(dh[0], mbits[i], 0), mask4(dh[1], mbits[i], 4)};
            if (want_w) {
                deposit(my, TP, dpre[0], p, q); deposit(my, TP + 1, dpre[1], p, q);
                if constexpr (i >= 1) {
                    constexpr int j = i == 3 ? 2 : i - 1;
                    deposit(my, TX, h[j][0][0], p, q); deposit(my, TX + 1, h[j][0][1], p, q);
                }
                __syncthreads();
                if constexpr (i == 0) {
                    own_outer_1(aW[0][0], smem, SLOT, TP, SL::C, 2, 4, wave, lane);
                } else if constexpr (i == 3) {
                    own_outer<2>(aW[3], smem, SLOT, TP, SL::C, 4, 8, wave, lane);                  // [c | h2] contiguous
                } else {
                    own_outer_1(aW[i][0], smem, SLOT, TP, TX, 2, 4, wave, lane);
                }
                if (wave < 2) own_bias(aB[i], smem, SLOT, TP + wave, lane);
            }
            if constexpr (i == 0) {
                linear_n<2, 2>(dc, pk + L.oWT(0), 32, dpre, p, q);
            } else if constexpr (i == 3) {
                linear_n<2, 2>(dc, pk + L.oWT(3), 32, dpre, p, q);
                dh[0] = dh[1] = splat4(0.f);
                linear_n<2, 2>(dh, pk + L.oWT(3) + 32 * 32, 32, dpre, p, q);
            } else {
                dh[0] = dh[1] = splat4(0.f);
                linear_n<2, 2>(dh, pk + L.oWT(i), 32, dpre, p, q);
            }
        };
        bwd_layer(IC(4)); bwd_layer(IC(3)); bwd_layer(IC(2)); bwd_layer(IC(1)); bwd_layer(IC(0));
        if (want_r) {
            float gx, gy, gz;
            coord_grad_partial(v, grid, q, dc[0], dc[1], gx, gy, gz);
            gx += __shfl_xor(gx, 16); gx += __shfl_xor(gx, 32);
            gy += __shfl_xor(gy, 16); gy += __shfl_xor(gy, 32);
            gz += __shfl_xor(gz, 16); gz += __shfl_xor(gz, 32);
            float dpx = gx * v.gx, dpy = gy * v.gy, dpz = gz * v.gz;
            if (q != 0) { dpx = dpy = dpz = 0.f; }
            if (tvalid) add_ray_grad(dpx, dpy, dpz, G.zf, G.ray, A.g_ro, A.g_rd, lane);
        }
        if (want_g && tvalid) {
            if (want_w) __syncthreads();                 // all waves are done reading this round's P tiles
            float* stg = my + SL::P0 * 256;
            *reinterpret_cast<f32x4*>(stg + p * 32 + 4 * q) = dc[0];
            *reinterpret_cast<f32x4*>(stg + p * 32 + 16 + 4 * q) = dc[1];
            wave_lds_fence();
            scatter_tile(stg, v, ggrid, lane);
            wave_lds_fence();
        }
    }
    if (want_w) {
        float* sacc = smem;
        __syncthreads();
        for (int e = threadIdx.x; e < GF; e += 256) sacc[e] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i == 3) {
#pragma unroll
                for (int j = 0; j < 2; ++j) stage_tile(sacc + L.oW(3), 64, 0, 4, wave + 4 * j, aW[3][j], 32, 64, p, q);
            } else {
                stage_tile(sacc + L.oW(i), 32, 0, 2, wave, aW[i][0], 32, 32, p, q);
            }
            if (wave < 2) stage_bias(sacc + L.ob(i), wave, aB[i], 32, p, q);
        }
        if (wave < 2) stage_tile(sacc + L.oWo(), 32, 0, 2, wave, aWo[0], 1, 32, p, q);
        if (wave == 2) stage_bias(sacc + L.obo(), 0, aBo, 1, p, q);
        __syncthreads();
        flush_image(sacc, GF, gpk, A.gpart[0], wg, n_wg, 256);
    }
}

extern __shared__ __attribute__((aligned(16))) float ens_smem[];

template <bool SAVED>
__global__ __launch_bounds__(256, 1) void decoder_bwd_kernel(BwdArgs A) {
    int role = 0;
#pragma unroll
    for (int r = 1; r < 4; ++r) role = (r < A.n_roles && (int)blockIdx.x >= A.role_begin[r]) ? r : role;
    const int wg = blockIdx.x - A.role_begin[role], n_wg = A.role_begin[role + 1] - A.role_begin[role];
    const int kind = A.role_kind[role];
    switch (kind) {
        case 0: feat_role(A, wg, n_wg, ens_smem); break;
        case 1: if constexpr (SAVED) xyz_role_saved<2, 1, true>(A, 1, wg, n_wg, ens_smem); else xyz_role<2, 1>(A, 1, wg, n_wg, ens_smem); break;
        case 2: if constexpr (SAVED) xyz_role_saved<4, 1, true>(A, 2, wg, n_wg, ens_smem); else xyz_role<4, 1>(A, 2, wg, n_wg, ens_smem); break;
        case 3: if constexpr (SAVED) xyz_role_saved<2, 4, true>(A, 3, wg, n_wg, ens_smem); else xyz_role<2, 4>(A, 3, wg, n_wg, ens_smem); break;
        default: break;
    }
}

// The saved-activation backward with the two MFMA streams of a decoder on separate waves: 8 waves per workgroup, two
// per SIMD.  Waves 0..3 run the dX chain of one tile each (xyz_role_saved<.., SPLIT>), waves 4..7 accumulate the owned
// weight-gradient tiles (xyz_dw_loop) from the operands the chain waves deposit, one barrier phase behind.
template <bool DF>
ENS_DEV void split_kernel_body(const BwdArgs& A) {
    TL(0)           // (stamps build) kernel entry
    if (threadIdx.x < SY_N) ens_sync[threadIdx.x] = 0;
    __syncthreads();
    int role = 0;
#pragma unroll
    for (int r = 1; r < 4; ++r) role = (r < A.n_roles && (int)blockIdx.x >= A.role_begin[r]) ? r : role;
    const int wg = blockIdx.x - A.role_begin[role], n_wg = A.role_begin[role + 1] - A.role_begin[role];
    const bool chain = threadIdx.x < 256;
    switch (A.role_kind[role]) {
        case 1: if (chain) xyz_role_saved<2, 1, true, true, DF>(A, 1, wg, n_wg, ens_smem); else xyz_dw_loop<2, 1>(A, 1, wg, n_wg, ens_smem); break;
        case 2: if (chain) xyz_role_saved<4, 1, true, true, DF>(A, 2, wg, n_wg, ens_smem); else xyz_dw_loop<4, 1>(A, 2, wg, n_wg, ens_smem); break;
        case 3: if (chain) xyz_role_saved<2, 4, true, true, DF>(A, 3, wg, n_wg, ens_smem); else xyz_dw_loop<2, 4>(A, 3, wg, n_wg, ens_smem); break;
        default: break;
    }
}
__global__ __launch_bounds__(512, 1) void decoder_bwd_split_kernel(BwdArgs A) { split_kernel_body<false>(A); }
__global__ __launch_bounds__(512, 1) void decoder_bwd_split_defer_kernel(BwdArgs A) { split_kernel_body<true>(A); }

// Saved-activation backward of decoders whose parameters get no gradient: the dX chain, the feature-gradient scatter
// and the ray-gradient hand-off only.  No accumulators -> two workgroups per CU overlap each other's latencies.
template <bool DF>
ENS_DEV void light_kernel_body(const BwdArgs& A) {
    int role = 0;
#pragma unroll
    for (int r = 1; r < 4; ++r) role = (r < A.n_roles && (int)blockIdx.x >= A.role_begin[r]) ? r : role;
    const int wg = blockIdx.x - A.role_begin[role], n_wg = A.role_begin[role + 1] - A.role_begin[role];
    switch (A.role_kind[role]) {
        case 1: xyz_role_saved<2, 1, false, false, DF>(A, 1, wg, n_wg, ens_smem); break;
        case 2: xyz_role_saved<4, 1, false, false, DF>(A, 2, wg, n_wg, ens_smem); break;
        case 3: xyz_role_saved<2, 4, false, false, DF>(A, 3, wg, n_wg, ens_smem); break;
        default: break;
    }
}
__global__ __launch_bounds__(256, 2) void decoder_bwd_light_kernel(BwdArgs A) { light_kernel_body<false>(A); }
__global__ __launch_bounds__(256, 2) void decoder_bwd_light_defer_kernel(BwdArgs A) { light_kernel_body<true>(A); }
constexpr int lds_bytes_light() { return (WAREA_SAVED + 4 * 512) * 4; }

// deposit slots of the 4 waves; the packed-layout flush image aliases them at the end of the kernel
constexpr int lds_bytes_xyz(int ct) { return (cmax(XyzLay{ct * 16}.fwd_floats(), 4 * (23 + ct) * 256) + 2 * ring_floats(ct)) * 4; }
constexpr int lds_bytes_xyz_saved(int ct) { return (cmax(XyzLay{ct * 16}.fwd_floats(), 4 * (23 + ct) * 256) + WAREA_SAVED) * 4; }
constexpr int lds_bytes_feat() { return cmax(FeatLay{}.fwd_floats(), 4 * 11 * 256) * 4; }

}  // namespace

int ens_bwd_max_workgroups() { return device_cus(); }

#ifdef ENS_STAMPS
extern "C" int enslam_debug_set_stamp_buffer(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &v, sizeof(v)) == hipSuccess ? 0 : -2;
}
#endif

int ens_launch_composite_bwd(int n_rays, int S, const float* raw, const double* z, const double* depth,
                             const double* g_depth, const double* g_var, const float* g_rgb, float* d_raw,
                             hipStream_t st, const LossSpec* ls, const float* rgb, const WorkList* wl) {
    if (n_rays <= 0) return 0;
    LossSpec l{nullptr, nullptr, 0.f, nullptr, nullptr, nullptr};
    if (ls != nullptr) l = *ls;
    WorkList wk{nullptr, nullptr};
    if (wl != nullptr) wk = *wl;
    if (wk.tiles != nullptr) composite_bwd_kernel<<<dim3((n_rays + 15) / 16), dim3(1024), 0, st>>>(n_rays, S, raw, z, depth, g_depth, g_var, g_rgb, d_raw, l, rgb, wk);
    else composite_bwd_kernel<<<dim3(n_rays), dim3(64), 0, st>>>(n_rays, S, raw, z, depth, g_depth, g_var, g_rgb, d_raw, l, rgb, wk);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_decoder_bwd(int stage, int ntl, int n_rays, const float* ro, const float* rd, const double* z,
                           const DevScene& sc, const float* d_raw, const float* act_ws, int act_light, float* dgrid_ws,
                           const DevGrid* grad_grids, float* const* grad_packed, float* g_ro, float* g_rd,
                           hipStream_t st, const double* draw_scale, const WorkList* wl, float* const* grad_partial) {
    if (n_rays <= 0) return 0;
    BwdArgs A;
    A.act_light = act_light;
    A.act_ws = stage == 0 ? nullptr : act_ws;
    A.dgrid_ws = dgrid_ws;
    // the dh region of the two-kernel form follows the activation blocks of a full workspace (enslam_activation_floats)
    A.dh_ws = (A.act_ws != nullptr && !act_light) ? const_cast<float*>(act_ws) + (size_t)n_rays * ntl * ACT_SLOTS * ACT_STRIDE : nullptr;
    A.n_rays = n_rays; A.ntl = ntl; A.ro = ro; A.rd = rd; A.z = z; A.d_raw = d_raw; A.draw_scale = draw_scale; A.sc = sc;
    const bool listed = wl != nullptr && wl->tiles != nullptr && act_ws != nullptr && stage != 0;   // (the recompute and coarse roles walk every tile)
    A.work = listed ? wl->tiles : nullptr; A.n_work = listed ? wl->count : nullptr;
    A.g_ro = (g_ro && g_rd) ? g_ro : nullptr;
    A.g_rd = (g_ro && g_rd) ? g_rd : nullptr;
    // roles and their relative cost (MFMA count per tile: middle/color 270, fine 350)
    int kinds[3], nk = 0;
    float cost[3];
    switch (stage) {
        case 0: kinds[nk] = 0; cost[nk++] = 1.f; break;
        case 1: kinds[nk] = 1; cost[nk++] = 1.f; break;
        case 2: kinds[nk] = 1; cost[nk++] = 0.44f; kinds[nk] = 2; cost[nk++] = 0.56f; break;
        case 3: kinds[nk] = 1; cost[nk++] = 0.315f; kinds[nk] = 2; cost[nk++] = 0.365f; kinds[nk] = 3; cost[nk++] = 0.32f; break;
    }
    if (const char* e = getenv("ENS_ROLE_COST")) {          // tuning aid: "a,b,c" relative cost of middle, fine, color
        float x[3];
        if (stage == 3 && sscanf(e, "%f,%f,%f", &x[0], &x[1], &x[2]) == 3) { cost[0] = x[0]; cost[1] = x[1]; cost[2] = x[2]; }
    }
    switch (stage) {
        case 0: case 1: case 2: case 3: break;
        default: return -1;
    }
    for (int k = 0; k < 4; ++k) { A.ggrid[k] = DevGrid{nullptr, 0, 0, 0}; A.gpacked[k] = nullptr; A.gpart[k] = nullptr; }
    int lds = 0, lds_saved = 0;
    // drop roles with nothing to produce
    int kk[3], n2 = 0;
    float cc[3], csum = 0.f;
    for (int i = 0; i < nk; ++i) {
        const int k = kinds[i];
        const bool any = grad_grids[k].data != nullptr || grad_packed[k] != nullptr || A.g_ro != nullptr;
        if (!any) continue;
        A.ggrid[k] = grad_grids[k];
        A.gpacked[k] = grad_packed[k];
        A.gpart[k] = (grad_partial != nullptr && grad_packed[k] != nullptr) ? grad_partial[k] : nullptr;
        kk[n2] = k; cc[n2] = cost[i]; csum += cost[i]; ++n2;
        const int need = k == 0 ? lds_bytes_feat() : lds_bytes_xyz(k == 2 ? 4 : 2);
        lds = need > lds ? need : lds;
        const int need_s = k == 0 ? lds_bytes_feat() : lds_bytes_xyz_saved(k == 2 ? 4 : 2);
        lds_saved = need_s > lds_saved ? need_s : lds_saved;
    }
    if (n2 == 0) return 0;
    // Deferred feature-gradient scatter (grid_scatter.hip; opt-in): with a hand-off workspace the saved-activation kernels leave dC there
    // and a launch of its own forms the sums that share a voxel row on chip before they reach the gradient (6.6 MB of float atomics per
    // 1000-ray step instead of 50 MB; 50 us).  ENSLAM_DEFER_SCATTER=1: whenever a hand-off workspace is given; =2: when a LIGHT role -- a
    // decoder without parameter gradients, e.g. the fixed occupancy decoders of the reference's mapper -- has a grid gradient (the light
    // kernel has no second wave kind to hide the atomics behind: 75.7 us with the scatter, 19 without; once the launch exists the heavy
    // roles hand off too).  Default off: on a RANDOM-INIT map that policy takes the mapper's backward from 151 to 117 us, on a map with
    // surfaces most feature gradients are exact zeros, the in-kernel scatter skips them, and the extra launch loses (mapper iteration
    // 307 -> 327 us); with heavy roles only the persistent kernel hides most of it (134 us against 100 + 50).
    static const int defer_mode = [] { const char* e = getenv("ENSLAM_DEFER_SCATTER"); return e == nullptr ? 0 : (e[0] == '1' ? 1 : (e[0] == '2' ? 2 : 0)); }();
    static const bool split_on = [] { const char* e = getenv("ENS_SPLIT"); return e == nullptr || e[0] != '0'; }();   // (the 4-wave A/B kernel has no deferred form)
    A.defer_mask = 0;
    if (defer_mode != 0 && split_on && A.act_ws != nullptr && stage != 0 && dgrid_ws != nullptr) {
        bool light_grid = false;
        for (int i = 0; i < n2; ++i) light_grid = light_grid || (A.ggrid[kk[i]].data != nullptr && A.gpacked[kk[i]] == nullptr);
        if (defer_mode == 1 || light_grid)
            for (int i = 0; i < n2; ++i)
                if (A.ggrid[kk[i]].data != nullptr) A.defer_mask |= 1 << kk[i];
    }
    const int64_t n_tiles = (int64_t)n_rays * ntl;
    static bool attr_done[ENS_MAX_DEVICES] = {};
    bool& attr_set = attr_done[ens_device_ordinal()];
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_bwd_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_xyz(4)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_bwd_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_xyz(4)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_bwd_split_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_xyz(4)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_bwd_split_defer_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_xyz(4)) != hipSuccess)
            return -2;
        attr_set = true;
    }
    // Roles whose decoder gets no parameter gradient run in the light kernel (saved-activation path only); the others
    // in the persistent one.  Each launch splits its workgroups over its roles by integer rounds: a workgroup takes 4
    // tiles per round, so a role with n workgroups needs ceil(groups / n) rounds of relative cost cc -- pick the split
    // whose slowest role finishes first (the proportional split wastes up to one round of the slowest role: 4-11 % at
    // 1000 rays).  Searched once per (shape, roles).
    const int groups = (int)((n_tiles + 3) / 4);
    // With a work list the kernels walk the ACTIVE tiles only, and their number is on the device.  ENS_ACTIVE_FRACTION=f (A/B aid) lets the
    // split search run on f x the tiles: at 0.67 the random-init bench scene gains 4.5 us (134.9 -> 130.4, five alternating pairs) and
    // the fitted scene loses 14 (114 -> 128): the roles' relative costs move with the scene (zero feature gradients skip their atomics),
    // so the quantisation that suits one does not suit the other.  Default: all tiles, as before.
    static const float act_frac = [] { const char* e = getenv("ENS_ACTIVE_FRACTION"); const float f = e ? (float)atof(e) : 1.f; return f > 0.05f && f <= 1.f ? f : 1.f; }();
    const int groups_eff = listed ? (int)(groups * act_frac + 0.5f) > 0 ? (int)(groups * act_frac + 0.5f) : 1 : groups;
    // chain waves + dW waves (decoder_bwd_split_kernel) unless ENS_SPLIT=0 selects the 4-wave kernel (A/B aid)
    static const bool use_split = [] { const char* e = getenv("ENS_SPLIT"); return e == nullptr || e[0] != '0'; }();
    // ENS_BWD2=1: the two-kernel form (render_bwd2.hip: dX-chain kernel + split-K weight-gradient kernel).  Built and parity-
    // tested in round 4, slower than the persistent kernel (the feature-gradient scatter's atomics need a
    // kernel that lasts as long as they take to drain), so it stays an A/B aid.
    static const bool use_bwd2 = [] { const char* e = getenv("ENS_BWD2"); return e != nullptr && e[0] == '1'; }();
    auto launch_subset = [&](const int* ks, const float* cs, int n, bool light) -> int {
        if (n == 0) return 0;
        int total = device_cus() * (light ? 2 : 1);
        const int64_t max_useful = (int64_t)groups * n;
        if (total > max_useful) total = (int)max_useful;
        if (total < n) total = n;
        int split[3] = {0, 0, 0};
        static thread_local int c_key[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}}, c_split[2][3];
        int mask = 0;
        for (int i = 0; i < n; ++i) mask |= 1 << ks[i];
        int* key = c_key[light ? 1 : 0];
        int* cached = c_split[light ? 1 : 0];
        if (key[0] == groups && key[1] == total && key[2] == mask && key[3] == stage) {
            for (int i = 0; i < 3; ++i) split[i] = cached[i];
        } else {
            auto rounds = [&](int m) { return (groups_eff + m - 1) / m; };
            if (n == 1) {
                split[0] = total;
            } else if (n == 2) {
                float best = 1e30f;
                for (int a = 1; a < total; ++a) {
                    const float t = fmaxf(rounds(a) * cs[0], rounds(total - a) * cs[1]);
                    if (t < best) { best = t; split[0] = a; split[1] = total - a; }
                }
            } else {
                float best = 1e30f;
                for (int a = 1; a < total - 1; ++a) {
                    const float ta = rounds(a) * cs[0];
                    if (ta >= best) continue;
                    for (int b2 = 1; b2 < total - a; ++b2) {
                        const float t = fmaxf(ta, fmaxf(rounds(b2) * cs[1], rounds(total - a - b2) * cs[2]));
                        if (t < best) { best = t; split[0] = a; split[1] = b2; split[2] = total - a - b2; }
                    }
                }
            }
            key[0] = groups; key[1] = total; key[2] = mask; key[3] = stage;
            for (int i = 0; i < 3; ++i) cached[i] = split[i];
        }
        BwdArgs B = A;
        for (int k = 0; k < 4; ++k) { B.ggrid[k] = DevGrid{nullptr, 0, 0, 0}; B.gpacked[k] = nullptr; B.gpart[k] = nullptr; }
        B.n_roles = n;
        int begin = 0;
        for (int i = 0; i < n; ++i) {
            B.role_kind[i] = ks[i];
            B.role_begin[i] = begin;
            B.ggrid[ks[i]] = A.ggrid[ks[i]];
            if ((A.defer_mask >> ks[i]) & 1) B.ggrid[ks[i]].data = nullptr;       // (dims kept; the kernels test .data)
            B.gpacked[ks[i]] = A.gpacked[ks[i]];
            B.gpart[ks[i]] = A.gpart[ks[i]];
            begin += split[i];
        }
        B.role_begin[n] = total;
        for (int i = n; i < 4; ++i) B.role_kind[i] = -1;
        const bool df = A.defer_mask != 0;
        if (light && df) decoder_bwd_light_defer_kernel<<<dim3(total), dim3(256), lds_bytes_light(), st>>>(B);
        else if (light) decoder_bwd_light_kernel<<<dim3(total), dim3(256), lds_bytes_light(), st>>>(B);
        else if (A.act_ws != nullptr && use_split && ks[0] != 0 && df)
            decoder_bwd_split_defer_kernel<<<dim3(total), dim3(512), lds_saved, st>>>(B);
        else if (A.act_ws != nullptr && use_split && ks[0] != 0)
            decoder_bwd_split_kernel<<<dim3(total), dim3(512), lds_saved, st>>>(B);
        else if (A.act_ws != nullptr) decoder_bwd_kernel<true><<<dim3(total), dim3(256), lds_saved, st>>>(B);   // ray gradients: ens_launch_ray_grad_bwd
        else decoder_bwd_kernel<false><<<dim3(total), dim3(256), lds, st>>>(B);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    };
    int hk[3], lk[3], nh = 0, nl = 0;
    float hc[3], lc[3];
    for (int i = 0; i < n2; ++i) {
        const bool light = A.act_ws != nullptr && kk[i] != 0 && A.gpacked[kk[i]] == nullptr;
        if (light) { lk[nl] = kk[i]; lc[nl++] = cc[i]; } else { hk[nh] = kk[i]; hc[nh++] = cc[i]; }
    }
    const int rl = launch_subset(lk, lc, nl, true);
    if (rl != 0) return rl;
    bool partials = false;                                          // (partial images: the persistent kernel's flush only)
    for (int i = 0; i < nh; ++i) partials = partials || A.gpart[hk[i]] != nullptr;
    if (nh > 0 && use_bwd2 && A.act_ws != nullptr && A.dh_ws != nullptr && hk[0] != 0 && !partials) {
        BwdArgs A2 = A;                                             // (the two-kernel form keeps its own scatter)
        A2.defer_mask = 0;
        const int r2 = ens_launch_decoder_bwd2(A2, hk, hc, nh, stage, n_tiles, st);
        if (r2 != 0) return r2;
        int lmask = 0;
        for (int i = 0; i < nl; ++i) lmask |= 1 << lk[i];
        A.defer_mask &= lmask;
    } else {
        const int rh = launch_subset(hk, hc, nh, false);
        if (rh != 0) return rh;
    }
    if (A.defer_mask != 0) {
        DevGrid dg[4];
        for (int k = 0; k < 4; ++k) { dg[k] = A.ggrid[k]; if (!((A.defer_mask >> k) & 1)) dg[k].data = nullptr; }
        return ens_launch_grid_scatter(stage, ntl, n_rays, ro, rd, z, sc, dgrid_ws, listed ? d_raw : nullptr, dg, st);
    }
    return 0;
}

// Second kernel of the saved-activation backward: ray gradients from the decoder kernel's hand-off buffer.
bool ens_ray_grad_args(int stage, int ntl, int n_rays, const float* ro, const float* rd, const double* z, const DevScene& sc,
                       float* dgrid_ws, float* g_ro, float* g_rd, RayGradArgs& A, const WorkList* wl) {
    if (n_rays <= 0 || stage == 0 || !dgrid_ws || !g_ro || !g_rd) return false;
    A.n_rays = n_rays; A.ntl = ntl; A.n_slots = stage;       // middle | middle+fine | middle+fine+color
    A.ro = ro; A.rd = rd; A.z = z; A.dgrid_ws = dgrid_ws; A.g_ro = g_ro; A.g_rd = g_rd;
    for (int a = 0; a < 3; ++a) { A.lo[a] = sc.lo[a]; A.hi[a] = sc.hi[a]; }
    for (int k = 0; k < 4; ++k) A.grid[k] = sc.grid[k];
    A.work = (wl != nullptr) ? wl->tiles : nullptr; A.n_work = (wl != nullptr) ? wl->count : nullptr;
    return true;
}
int ens_launch_ray_grad_bwd(int stage, int ntl, int n_rays, const float* ro, const float* rd, const double* z,
                            const DevScene& sc, float* dgrid_ws, float* g_ro, float* g_rd, hipStream_t st) {
    if (n_rays <= 0 || stage == 0) return 0;
    RayGradArgs A;
    if (!ens_ray_grad_args(stage, ntl, n_rays, ro, rd, z, sc, dgrid_ws, g_ro, g_rd, A)) return -1;
    grid_bwd_kernel<<<dim3((unsigned)((int64_t)n_rays * ntl * A.n_slots)), dim3(64), 0, st>>>(A);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
