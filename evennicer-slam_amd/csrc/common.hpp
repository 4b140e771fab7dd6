// Shared device helpers for the gfx950 rendering kernels.
//
// Register-tile convention used everywhere ("D layout" of v_mfma_f32_16x16x4_f32):
//   lane = 16*q + p  (p = lane & 15, q = lane >> 4)
//   an activation tile  f32x4 x[t]  holds, for sample point p of the tile,
//   features 16*t + 4*q + r  in component r.
// This is what the MFMA writes as C/D (col = lane&15, row = 4*(lane>>4)+reg), and it is also a valid
// B operand of the next layer when the K order is permuted to (t, r): at step (t,r) k-slot q carries
// feature 16t+4q+r.  The matching A operand (weights, row i = lane&15 of the output tile) is then the
// 4 consecutive floats  W[16*rt + i][16*t + 4*q .. +3]  of a row-major weight matrix, i.e. one 16-byte
// load per lane per 4 MFMAs, with no packing beyond zero-padding K to a multiple of 16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ENS_DEV __device__ __forceinline__
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

ENS_DEV f32x4 ld4(const float* __restrict__ p) { return *reinterpret_cast<const f32x4*>(p); }
#ifdef ENS_EXP_L1WEIGHTS      // timing experiment only (wrong results): all weight fragments from one 4 KB window
#define ldw(base, off) ld4((base) + ((off) & 1020))
#else
#define ldw(base, off) ld4((base) + (off))
#endif
ENS_DEV f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
ENS_DEV f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }

// ----------------------------------------------------------------------------------------------
// Packed decoder layouts (float offsets).  Forward section first (the gradient accumulator has the
// same layout and size), then the transposed copies the backward chain reads.
// ----------------------------------------------------------------------------------------------
struct XyzLay {                 // MLP (middle / fine / color): decoder.py:91-203
    int CD;                     // fc_c input width: 32, fine 64
    constexpr int K(int i) const { return i == 0 ? 96 : (i == 3 ? 128 : 32); }   // padded pts_linears input width
    constexpr int oBT() const { return 0; }                                        // [96][4]  B^T (rows >=93, col 3 zero)
    constexpr int oW(int i) const {
        int o = 384;
        for (int j = 0; j < i; ++j) o += 32 * K(j) + 32 + 32 * CD + 32;
        return o;
    }
    constexpr int ob(int i) const { return oW(i) + 32 * K(i); }
    constexpr int oWc(int i) const { return ob(i) + 32; }
    constexpr int obc(int i) const { return oWc(i) + 32 * CD; }
    constexpr int oWo() const { return oW(5); }                                    // [16][32], rows >= n_out zero
    constexpr int obo() const { return oWo() + 512; }                              // [16]
    constexpr int fwd_floats() const { return obo() + 16; }
    constexpr int oWT(int i) const {                                               // [K(i)][32]
        int o = fwd_floats();
        for (int j = 0; j < i; ++j) o += 32 * K(j);
        return o;
    }
    constexpr int oWcT(int i) const { return oWT(5) + i * 1024; }                  // [32][32] grid channels only
    constexpr int oWoT() const { return oWcT(5); }                                 // [32][4]
    constexpr int oBp() const { return oWoT() + 128; }                             // [16][96] rows 0..2 = B
    constexpr int total() const { return oBp() + 16 * 96; }
};

struct FeatLay {                // MLP_no_xyz (coarse): decoder.py:206-274
    constexpr int K(int i) const { return i == 3 ? 64 : 32; }
    constexpr int oW(int i) const {
        int o = 0;
        for (int j = 0; j < i; ++j) o += 32 * K(j) + 32;
        return o;
    }
    constexpr int ob(int i) const { return oW(i) + 32 * K(i); }
    constexpr int oWo() const { return oW(5); }
    constexpr int obo() const { return oWo() + 512; }
    constexpr int fwd_floats() const { return obo() + 16; }
    constexpr int oWT(int i) const {
        int o = fwd_floats();
        for (int j = 0; j < i; ++j) o += 32 * K(j);
        return o;
    }
    constexpr int oWoT() const { return oWT(5); }
    constexpr int total() const { return oWoT() + 128; }
};

// ----------------------------------------------------------------------------------------------
// Activation workspace written by the forward for the backward (one block per 16-sample tile and decoder slot):
//   tiles in the backward's LDS "deposit" layout, in the order of its slot
//   [EMB 6 | h2 2 | h0 2 | h1 2 | h3 2 | C ct | XYZ 1 (sample coordinates as features 0..2)]
//   (address of (tile T, feature i, sample pt) = T*256 + (pt>>2)*64 + i*4 + (pt&3) floats), then h4 in register
//   layout (2 tiles, lane-linear) and the ReLU masks (2 words per lane).
// ----------------------------------------------------------------------------------------------
constexpr int ACT_DEP_TILES_MAX = 19;                       // 14 + ct + 1, ct <= 4
constexpr int ACT_H4 = ACT_DEP_TILES_MAX * 256;             // float offset of h4 (2 tiles)
constexpr int ACT_MASK = ACT_H4 + 2 * 256;                  // float offset of the mask words (128 words)
constexpr int ACT_VOX = ACT_MASK + 128;                     // float offset of the 16 cell records (vox_record)
constexpr int ACT_STRIDE = ACT_VOX + 64;                    // floats per (tile, decoder slot)
// light workspace (no decoder-parameter gradients will be asked for): coordinates tile | mask words | cell records
constexpr int ACTL_Q = 0, ACTL_MASK = 256, ACTL_VOX = 384, ACTL_STRIDE = 448;
constexpr int64_t ACT_MAX_VOXELS = (int64_t)1 << 29;        // a cell record keeps the linear voxel index in 29 bits
constexpr int ACT_SLOTS = 3;                                // decoder slots per tile: middle, fine, color
// Hand-off from decoder_bwd_kernel to grid_bwd_kernel, per (tile, decoder slot): dC in register layout (2 tiles,
// lane-linear) and the embedding's position gradient (one float4 per lane, meaningful on q == 0 lanes).
constexpr int DG_DPE = 2 * 256;
constexpr int DG_STRIDE = 3 * 256;
// Two-kernel backward (render_bwd2.hip): dh_i of the five layers, chain kernel -> weight-gradient kernel, per (tile, decoder
// slot): 10 tiles in register layout (lane-linear), layer i's row tile rt at (2 i + rt) * 256, then dC as [sample][32]
// (2 tiles) for the scatter.  The region follows the activation blocks in the full workspace (enslam_activation_floats
// counts it).
constexpr int DH_STRIDE = 12 * 256;

// ----------------------------------------------------------------------------------------------
// Scene description passed by value to kernels
// ----------------------------------------------------------------------------------------------
struct DevGrid {
    float* data;          // [D*H*W][32]
    int D, H, W;
};
struct DevScene {
    double lo[3], hi[3];      // Renderer.bound
    double clo[3], chi[3];    // coarse decoder bound
    float gs[3];              // (float)(2 / (hi - lo)): the gradient scale of the normalisation, divided once on the host
    DevGrid grid[4];
    const float* packed[4];
};

// Position of one sample inside one grid: base corner, fractions, clip-gradient flags.
struct Vox {
    int ix, iy, iz;
    float fx, fy, fz;
    float gx, gy, gz;     // d(unnormalised, clipped coord)/d(world coord); 0 where clipped
};

// normalize_3d_coordinate (common.py:354-356) in float64, cast to float32, then ATen
// grid_sampler_unnormalize (align_corners) / clip_coordinates / floor in float32.
ENS_DEV void axis_coord(double pw, double lo, double hi, int size, int& i0, float& fr, float& gmul) {
    const double pn64 = ((pw - lo) / (hi - lo)) * 2.0 - 1.0;
    const float pn = (float)pn64;
    float c = ((pn + 1.f) / 2.f) * (float)(size - 1);
    const float mx = (float)(size - 1);
    float g = (float)(size - 1) / 2.f;            // unnormalize grad
    if (c <= 0.f) { c = 0.f; g = 0.f; }
    else if (c >= mx) { c = mx; g = 0.f; }
    const float fl = floorf(c);
    i0 = (int)fl;
    fr = c - fl;
    gmul = g * (float)(2.0 / (hi - lo));          // chain through the float64 normalisation
}

// The two halves of axis_coord for callers that look one point up in several grids over the SAME bound (middle, fine
// and colour grids): the float64 normalisation (a division) once, the per-grid unnormalise / clip / floor per grid.
ENS_DEV float axis_norm(double pw, double lo, double hi) { return (float)(((pw - lo) / (hi - lo)) * 2.0 - 1.0); }
ENS_DEV int axis_cell(float pn, int size) {
    float c = ((pn + 1.f) / 2.f) * (float)(size - 1);
    const float mx = (float)(size - 1);
    c = c <= 0.f ? 0.f : (c >= mx ? mx : c);
    return (int)floorf(c);
}

ENS_DEV Vox make_vox(const double pw[3], const double* lo, const double* hi, const DevGrid& g) {
    Vox v;
    axis_coord(pw[0], lo[0], hi[0], g.W, v.ix, v.fx, v.gx);
    axis_coord(pw[1], lo[1], hi[1], g.H, v.iy, v.fy, v.gy);
    axis_coord(pw[2], lo[2], hi[2], g.D, v.iz, v.fz, v.gz);
    return v;
}

// make_vox for callers that look ONE point up in several grids over the same bound (the forward: middle, fine, colour):
// the float64 normalisation and the float64 gradient scale -- six float64 divisions -- once per point, the per-grid
// unnormalise / clip / floor in float32 per grid.  Term for term the arithmetic of axis_coord (bit-identical results).
struct VoxNorm {
    float pn[3];          // normalised coordinate in [-1, 1], float32 of the float64 value
    float gs[3];          // (float)(2 / (hi - lo))
};
ENS_DEV VoxNorm vox_norm(const double pw[3], const double* lo, const double* hi, const float* gs) {
    VoxNorm n;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        n.pn[a] = (float)(((pw[a] - lo[a]) / (hi[a] - lo[a])) * 2.0 - 1.0);
        n.gs[a] = gs[a];
    }
    return n;
}
ENS_DEV void axis_coord_n(float pn, float gs, int size, int& i0, float& fr, float& gmul) {
    float c = ((pn + 1.f) / 2.f) * (float)(size - 1);
    const float mx = (float)(size - 1);
    float g = (float)(size - 1) / 2.f;
    if (c <= 0.f) { c = 0.f; g = 0.f; }
    else if (c >= mx) { c = mx; g = 0.f; }
    const float fl = floorf(c);
    i0 = (int)fl;
    fr = c - fl;
    gmul = g * gs;
}
ENS_DEV Vox make_vox_n(const VoxNorm& n, const DevGrid& g) {
    Vox v;
    axis_coord_n(n.pn[0], n.gs[0], g.W, v.ix, v.fx, v.gx);
    axis_coord_n(n.pn[1], n.gs[1], g.H, v.iy, v.fy, v.gy);
    axis_coord_n(n.pn[2], n.gs[2], g.D, v.iz, v.fz, v.gz);
    return v;
}

// corner k = 4*dz + 2*dy + dx (ATen order tnw,tne,tsw,tse,bnw,bne,bsw,bse); weight = (wx*wy)*wz.
// A corner beyond the last voxel has weight exactly 0 (fraction is 0 there); its index is clamped so
// the load stays in bounds, and the weight is forced to 0 as ATen skips it.
ENS_DEV void corner(const Vox& v, const DevGrid& g, int k, int64_t& idx, float& w) {
    const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
    int x = v.ix + dx, y = v.iy + dy, z = v.iz + dz;
    const bool ok = (x < g.W) && (y < g.H) && (z < g.D);
    x = min(x, g.W - 1); y = min(y, g.H - 1); z = min(z, g.D - 1);
    const float wx = dx ? v.fx : (1.f - v.fx);
    const float wy = dy ? v.fy : (1.f - v.fy);
    const float wz = dz ? v.fz : (1.f - v.fz);
    w = ok ? (wx * wy) * wz : 0.f;
    idx = ((int64_t)z * g.H + y) * g.W + x;
}

// The trilinear cell of a sample as the backward's scatter wants it: linear index of corner (0,0,0) in bits 0..28,
// "the +1 neighbour exists" flags of x, y, z in bits 29..31, then the three fractions.
ENS_DEV f32x4 vox_record(const Vox& v, const DevGrid& g) {
    unsigned lin = (unsigned)((v.iz * g.H + v.iy) * g.W + v.ix);
    lin |= (v.ix + 1 < g.W ? 1u << 29 : 0u) | (v.iy + 1 < g.H ? 1u << 30 : 0u) | (v.iz + 1 < g.D ? 1u << 31 : 0u);
    return f32x4{__builtin_bit_cast(float, lin), v.fx, v.fy, v.fz};
}

// Trilinear gather of the lane's 8 channels (16t+4q+r, t=0,1) of one sample.
ENS_DEV void gather8(const Vox& v, const DevGrid& g, int q, f32x4& c0, f32x4& c1) {
    c0 = splat4(0.f);
    c1 = splat4(0.f);
    f32x4 a[8], b[8];
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                      // all 16 loads in flight before the first use
        int64_t idx;
        corner(v, g, k, idx, w[k]);
        const float* src = g.data + idx * 32 + 4 * q;
        a[k] = ld4(src);
        b[k] = ld4(src + 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {                      // ATen corner order tnw, tne, tsw, tse, bnw, bne, bsw, bse
#pragma unroll
        for (int r = 0; r < 4; ++r) { c0[r] = fmaf(a[k][r], w[k], c0[r]); c1[r] = fmaf(b[k][r], w[k], c1[r]); }
    }
}

// Work-list append (see WorkList, kernels.hpp): one wave holds the d_raw of one ray, lane = sample; the waves of a workgroup
// (blockDim.x / 64 rays) append together with ONE atomic -- one atomic per ray, all on one address, cost 11 us per 1000
// rays.  Tiles whose 16 lanes carry any non-zero component are appended in ray order within the workgroup.  Every thread
// of the workgroup must call this (two barriers); rays beyond the batch pass ray_valid = false.
ENS_DEV void append_active_tiles_wg(int* tiles, int* count, int64_t ray, int ntl, bool ray_valid, bool lane_nonzero) {
    __shared__ int wk[16];
    __shared__ int wbase;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long b = __ballot(lane_nonzero);
    int k = 0;
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) k += (ray_valid && tl < ntl && ((b >> (16 * tl)) & 0xFFFFull)) ? 1 : 0;
    if (lane == 0) wk[wave] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < nw; ++w) { const int c = wk[w]; wk[w] = tot; tot += c; }       // exclusive offsets
        wbase = tot > 0 ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    if (lane == 0 && k > 0) {
        int j = wbase + wk[wave];
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
            if (tl < ntl && ((b >> (16 * tl)) & 0xFFFFull)) tiles[j++] = (int)(ray * ntl + tl);
    }
}

// ----------------------------------------------------------------------------------------------
// MFMA building blocks
// ----------------------------------------------------------------------------------------------
// acc[tl][rt] += W[32 x 16*KT] (row-major, leading dim ld) * x[tl][0..KT)   for NTL point tiles.
// All weight-fragment loads of the call are issued first (independent, distinct registers) and pinned there;
// the MFMAs then alternate accumulators so that consecutive issues never wait on their own result.
// TM: W is one of the tile-major fragment images of the xyz decoders (lds_util.hpp: unit (rt, t) at rt*16*ld + t*256,
// lane chunk at frag_off); the coarse decoder's images are plain row-major.
template <int KT, int NTL, int XS, bool TM = false>
ENS_DEV void linear32(f32x4 (&acc)[NTL][2], const float* __restrict__ W, int ld, const f32x4 (&x)[NTL][XS], int xoff,
                      int p, int q) {
    f32x4 a[KT][2];
    const int v = (p >> 1) & 3;
    const int fo = TM ? 64 * v + 16 * ((p & 1) + 2 * (p >> 3)) + 4 * (q ^ v) : p * ld + 4 * q;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) a[t][rt] = ldw(W, fo + 16 * rt * ld + (TM ? 256 : 16) * t);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int tl = 0; tl < NTL; ++tl) acc[tl][rt] = MFMA16(a[t][rt][r], x[tl][xoff + t][r], acc[tl][rt]);
            }
        }
    }
}

// out[tl] (one 16-row tile, rows >= n_out are zero padding) = Wo[16 x 32] * h + bo
template <int NTL>
ENS_DEV void out_layer(f32x4 (&o)[NTL], const float* __restrict__ Wo, const float* __restrict__ bo,
                       const f32x4 (&h)[NTL][2], int p, int q) {
    const f32x4 b = ld4(bo + 4 * q);
    const f32x4 a0 = ld4(Wo + p * 32 + 4 * q), a1 = ld4(Wo + p * 32 + 16 + 4 * q);
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) o[tl] = b;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) o[tl] = MFMA16(a0[r], h[tl][0][r], o[tl]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) o[tl] = MFMA16(a1[r], h[tl][1][r], o[tl]);
    }
}

// sin / cos of the Fourier argument p@B (|x| up to a few thousand: B ~ 25*randn, decoder.py:21-22), float32 only.
// Range reduction: k = rint(x * 2/pi), r = x - k*pi/2 with the Cody-Waite split of pi/2 into three float32 constants
// (twice Cephes' sinf DP1..DP3): C1 has 8 significant bits, so k*C1 is exact for |k| < 2^16 and the first fused
// multiply-add returns x - k*C1 exactly; the other two round once each.  |error of r| < 1.5e-7 for |x| < 1e5 (the argument
// itself carries 1.2e-4 of float32 rounding at |x| = 2000), then the classic float32 minimax polynomials on
// [-pi/4, pi/4].  No float64 instructions (they issue at half rate), no calls, no tables.
// tests/test_hip_forward.py::test_fourier_embedding_large_arguments pins |x| up to 4000 against float64.
ENS_DEV float ens_reduce_pio2(float x, int& n) {
    const float k = rintf(x * 0.63661977236758134308f);                 // 2/pi
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    n = (int)k;
    return r;
}
ENS_DEV void ens_sincosf(float x, float& s, float& c) {
    int n;
    const float r = ens_reduce_pio2(x, n);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                          fmaf(-0.5f, z, 1.f));
    const float ss = (n & 1) ? pc : ps;
    const float cc = (n & 1) ? ps : pc;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
// sin alone: the same reduction and the same two polynomials, but only the one the quadrant selects is evaluated
// (coefficients chosen by n & 1) -- bit-identical to the s of ens_sincosf.  The FORWARD's sine stays this polynomial form
// (9e-8 of float64): its values decide the ReLU masks of layer 0 and 3, and an implementation that strays 4x further from the
// reference's torch.sin flips more pre-activations that sit within rounding of zero -- the outputs do not notice, but the
// gradient of that sample jumps (measured in round 4 with the hardware sine in the forward: recording4 fixture, ONE ray of
// 1000 with a ray-gradient error of 1.9e-2 of the maximum, every rendered output still within 2.5e-6).
ENS_DEV float ens_sinf(float x) {
    int n;
    const float r = ens_reduce_pio2(x, n);
    const bool odd = n & 1;
    const float z = r * r;
    const float k3 = odd ? 2.443315711809948e-5f : -1.9515295891e-4f;
    const float k2 = odd ? -1.388731625493765e-3f : 8.3321608736e-3f;
    const float k1 = odd ? 4.166664568298827e-2f : -1.6666654611e-1f;
    const float t = fmaf(fmaf(k3, z, k2), z, k1) * z;
    const float v = fmaf(t, odd ? z : r, odd ? fmaf(-0.5f, z, 1.f) : r);
    return (n & 2) ? -v : v;
}
#ifndef ENS_POLY_SINCOS
// cos alone (the BACKWARD's d sin(x)/dx), default since round 4: the hardware's v_cos_f32 (argument in revolutions) behind a
// Cody-Waite reduction by 2*pi (6.28125 has 8 significant bits: k * 6.28125 is exact for |k| < 2^16).  7 vector instructions
// instead of ~20; 3.5e-7 maximum absolute error for |x| <= 4000 against 9e-8 of the polynomial form (measured on gfx950).  The
// factor enters d_arg = d_emb * cos(arg) smoothly (no mask depends on it): gradients move by ~3e-7 relative.
// -DENS_POLY_SINCOS restores the polynomial (A/B aid; profiles/r04_sincos_errors.txt holds both).
ENS_DEV float ens_red_rev(float x) {
    const float k = rintf(x * 0.15915494309189535f);
    float r = fmaf(-k, 6.28125f, x);
    r = fmaf(-k, 1.9350051879882812e-3f, r);
    r = fmaf(-k, 3.0199159819567529e-7f, r);
    return r * 0.15915494309189535f;
}
ENS_DEV float ens_cosf(float x) { return __builtin_amdgcn_cosf(ens_red_rev(x)); }
#else
// cos alone, same scheme as ens_sinf
ENS_DEV float ens_cosf(float x) {
    int n;
    const float r = ens_reduce_pio2(x, n);
    const bool odd = n & 1;                                             // odd quadrant: |cos| = sin polynomial
    const float z = r * r;
    const float k3 = odd ? -1.9515295891e-4f : 2.443315711809948e-5f;
    const float k2 = odd ? 8.3321608736e-3f : -1.388731625493765e-3f;
    const float k1 = odd ? -1.6666654611e-1f : 4.166664568298827e-2f;
    const float t = fmaf(fmaf(k3, z, k2), z, k1) * z;
    const float v = fmaf(t, odd ? r : z, odd ? r : fmaf(-0.5f, z, 1.f));
    return ((n + 1) & 2) ? -v : v;
}
#endif

// wave-wide helpers (64 lanes)
ENS_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
ENS_DEV double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
