// Two-kernel form of the saved-activation decoder backward for gfx950 (round 4).
//
// The persistent chain + dW kernel of render_bwd.hip runs the two MFMA streams of a decoder's backward -- the serial dX
// chain of a tile and the weight-gradient outer products over all samples -- in one workgroup, coupled through LDS
// hand-offs that leave both halves waiting for each other.  Here they are two launches:
//
//   decoder_chain_kernel : dX only.  A workgroup holds ALL transposed matrices of ONE decoder in LDS (69 KB, loaded
//                          once), every wave walks its own 16-sample tiles with no barrier and no hand-off: d_raw ->
//                          five backward layers on MFMA (W^T from LDS) -> embedding tail -> feature-gradient scatter
//                          (float atomics, deferred by one tile so they drain under the next tile's MFMAs) and the
//                          ray-gradient hand-off.  When decoder parameters want gradients it also stores dh_i of every
//                          layer (register layout, 10 KB per tile and decoder) for the second kernel and keeps the three
//                          small products that need nothing else: dB^T (MFMA over a wave-private LDS transposition),
//                          dWo / dbo (VALU).
//   decoder_dw_kernel    : the weight gradients as a split-K GEMM over the stored operands.  Per tile and decoder a
//                          workgroup streams the forward's activation tiles (already in the sample-in-K "deposit"
//                          layout) into a 3-stage LDS ring by LDS-DMA, turns dh_i into deposit tiles of dh_i and
//                          dpre_i = relu'(.) dh_i on the way, and its 8 waves accumulate the output tiles they own
//                          (dW_i = dpre_i^T x_i, dWc_i = dh_i^T c, bias rows) in registers over all of its tiles -- no
//                          dependency chain, one barrier per tile; one flush per workgroup at the end.
//
// Gradient semantics are those of render_bwd.hip (reference autograd of src/conv_onet/models/decoder.py:177-203,
// Mapper.py:573-575).
#include <type_traits>
#include <utility>
#include "common.hpp"
#include "kernels.hpp"
#include "lds_util.hpp"
#include "bwd_shared.hpp"

#define IC(n) std::integral_constant<int, n>{}

// Diagnostic build only (-DENS_STAMPS, tools/stamps_bwd2.py): per-wave s_memtime totals per code segment, written to a buffer of
// their own ([workgroup][16 waves][S2_NSEG], last slot = s_memrealtime span); never compiled into the shipped library.
#ifdef ENS_STAMPS
#define S2_NSEG 12
static __device__ unsigned long long* g_stamp_buf2 = nullptr;
#define S2_DECL unsigned long long s2_acc[S2_NSEG] = {}; unsigned long long s2_prev = 0, s2_rt0 = 0;
#define S2_START { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s2_rt0)::"memory"); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s2_prev)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#define S2(k) { unsigned long long n_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_)::"memory"); __builtin_amdgcn_sched_barrier(0); s2_acc[k] += n_ - s2_prev; s2_prev = n_; }
#define S2_FLUSH(sel_, wave_, lane_) { unsigned long long r1_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1_)::"memory"); s2_acc[S2_NSEG - 1] = r1_ - s2_rt0; if (g_stamp_buf2 && (lane_) == 0) { for (int k_ = 0; k_ < S2_NSEG; ++k_) g_stamp_buf2[(((size_t)(sel_) * 256 + blockIdx.x) * 16 + (wave_)) * S2_NSEG + k_] = s2_acc[k_]; } }
#else
#define S2_DECL
#define S2_START
#define S2(k)
#define S2_FLUSH(sel_, wave_, lane_)
#endif

namespace {

// cooperative async copy global -> LDS of n4 float4 by NWAVES waves (1 KB per wave instruction)
template <int NWAVES>
ENS_DEV void lds_fill(float* dst_lds, const float* __restrict__ src, int n4, int wave, int lane) {
    for (int j = 0; j * NWAVES * 64 < n4; ++j) {
        const int e0 = (j * NWAVES + wave) * 64;
        if (e0 + lane < n4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * (e0 + lane)),
                                             (__attribute__((address_space(3))) void*)(dst_lds + 4 * e0), 16, 0, 0);
    }
}
ENS_DEV void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
// lin_lds_swz (lds_util.hpp) on the rows R0 .. R0 + NRP - 1 of a taller accumulator block: acc[R0 + rt] += M[16 (R0 + rt) + p][.] x
template <int R0, int NRP, int KT, int LD, int OFF, int NTOT>
ENS_DEV void lin_lds_swz_rows(f32x4 (&acc)[NTOT], unsigned base_even, unsigned odd_delta, const f32x4 (&x)[KT]) {
    f32x4 a[KT][NRP];
    const unsigned base_odd = base_even - odd_delta;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int rt = 0; rt < NRP; ++rt) a[t][rt] = lds4(((t & 1) ? base_odd : base_even) + OFF + (16 * (R0 + rt) * LD + 16 * t) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int rt = 0; rt < NRP; ++rt) acc[R0 + rt] = MFMA16(a[t][rt][r], x[t][r], acc[R0 + rt]);
        }
    }
}
ENS_DEV int64_t work_count_u(const BwdArgs& A) { return A.work != nullptr ? (int64_t)uload(A.n_work) : (int64_t)A.n_rays * A.ntl; }
ENS_DEV int work_tile_u(const BwdArgs& A, int64_t v) { return A.work != nullptr ? uload(A.work + v) : (int)v; }

template <class F, int... I>
ENS_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
ENS_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ------------------------------------------------------------------------------------------------
// Kernel 1: the dX chain
// ------------------------------------------------------------------------------------------------
// LDS weight area (floats): per layer i the swizzled images W_i^T [K_i][32] | Wc_i^T [32][32], then B (padded, [16][96]) and
// B^T [96][4] -- the whole backward section of a packed decoder, resident for the life of the workgroup.
constexpr int c2_off(int i) { return i == 4 ? 0 : (i == 3 ? 2048 : (i == 2 ? 7168 : (i == 1 ? 9216 : 11264))); }
constexpr int C2_BP = 15360, C2_BT = 16896, C2_WFLOATS = 17280;
constexpr int C2_IMG = 384;                         // gradient image of a workgroup: dB^T [96][4]
constexpr int C2_WAVES = 12;                        // three per SIMD (<= 168 registers)
constexpr int C2_WAVE_FLOATS = 6 * 256;             // per-wave scratch: 6 tiles d_arg deposit (tiles 0, 1 double as dC staging)
// mode 3 (below): 8 chain waves + 4 scatter waves; per chain wave two hand-off buffers [dC 512 | cell records 64] behind its scratch
constexpr int C2_CHAIN_WAVES_3 = 8;
constexpr int C2_HAND = 512 + 64;
constexpr int C2_WAVE_FLOATS_3 = 6 * 256 + C2_HAND;
// ... and a WINDOW of the grid gradient in LDS: the 4 x 4 x 4 voxels around the origin of the workgroup's first ray.  Every ray of
// a camera starts in the same cell, so the first samples of ALL rays add into the same few voxel rows: measured with the scatter
// as a launch of its own, the tile next to the camera costs 62 of its 82 us (atomics on one address are served one after the
// other at the memory side) and all other tiles together 14.5.  Adds into the 3 x 3 x 3 cells of the window go to LDS
// (ds_add_f32) and leave the workgroup once, at its end: a hot row then receives one add per workgroup instead of one per ray.
constexpr int C2_WIN = 64 * 32;
// Where the feature-gradient scatter runs (A/B aid):
//   2 = in a launch of its own between the two kernels (decoder_scatter_kernel: one wave per tile and decoder, values in
//       registers, fire-and-forget atomics);
//   3 = in the chain kernel by DEDICATED scatter waves: 8 chain waves + 4 scatter waves per workgroup; a chain wave hands a
//       tile's dC and cell records over in LDS (two buffers, counters), a scatter wave serves two chain waves, reads only LDS
//       and issues atomics -- it never has a load to wait for, so no atomic is ever waited for either;
//   1 = in the chain kernel, the previous tile's scatter in four pieces in front of the backward layers 4..1 of the wave's next tile;
//   0 = in the weight-gradient kernel, two samples per wave and step.
// Measured (room0, 1000 rays, same box): 1 -> chain 47 -> 103 us (the scatter is ~1800 instructions per tile, as many as the
// chain itself, and every wait for a load behind it waits for its atomics); 0 -> weight-gradient kernel 59 -> 102 us (its
// barrier per item waits for the slowest wave's atomics).
#ifndef ENS_SCATTER_WHERE
#define ENS_SCATTER_WHERE 3
#endif
constexpr int C2_SYNC = 32;                         // hand-off counters (ints): ready[8] | done[8]
constexpr int lds_bytes_chain2() {
    return ENS_SCATTER_WHERE == 3 ? (C2_WFLOATS + C2_IMG + C2_SYNC + C2_WIN + C2_CHAIN_WAVES_3 * C2_WAVE_FLOATS_3) * 4
                                  : (C2_WFLOATS + C2_IMG + C2_WAVES * C2_WAVE_FLOATS) * 4;
}
ENS_DEV int lds_poll(const int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }

// What the chain kernel leaves per (tile, decoder slot) for the weight-gradient kernel (DH_STRIDE floats): dh_i of the five
// layers in register layout (tile 2 i + rt) and dC as [sample][32 channels] (tiles 10, 11), which that kernel scatters
// into the grid gradient.
constexpr int DH_DC = 10 * 256;

// ---- the scatter state machine of bwd_shared.hpp with the LDS window
struct GradWin { float* win; int lin0, W, HW; };          // window in LDS (null: none), linear index of its voxel (0,0,0), grid strides
struct ScatterStW { float acc[4]; unsigned cur; bool open; int woff; };
ENS_DEV int win_offset(unsigned cur, const GradWin& w) {  // float offset of cell `cur`'s corner voxel in the window, or -1 (all scalar)
    if (w.win == nullptr) return -1;
    const int d = (int)(cur & 0x1fffffffu) - w.lin0;
    if (d < 0 || d >= 3 * w.HW) return -1;
    const int dz = d >= 2 * w.HW ? 2 : (d >= w.HW ? 1 : 0);
    const int r = d - dz * w.HW;
    if (r >= 3 * w.W) return -1;
    const int dy = r >= 2 * w.W ? 2 : (r >= w.W ? 1 : 0);
    const int dx = r - dy * w.W;
    if (dx >= 3) return -1;
    return ((dz * 4 + dy) * 4 + dx) * 32;
}
ENS_DEV void scatter_flush_w(ScatterStW& st, const DevGrid& gg, const GradWin& w, int lane, int kmask, int half) {
    const int ch = lane & 31, dxb = lane >> 5;
    const int rowy = gg.W * 32, rowz = gg.H * gg.W * 32;
    const unsigned cur = st.cur;
    const bool okx = !dxb || (cur >> 29 & 1u), oky = cur >> 30 & 1u, okz = cur >> 31;
    const bool mine = half == 0 || (half == 1) == (dxb == 0);
    if (st.woff >= 0) {                                   // (wave-uniform) the whole cell lies in the window: LDS float atomics
        float* base = w.win + st.woff + dxb * 32 + ch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!((kmask >> k) & 1)) continue;
            if (mine && st.acc[k] != 0.f) lds_add(base + ((k >> 1) * 16 + (k & 1) * 4) * 32, st.acc[k]);
            if (mine) st.acc[k] = 0.f;
        }
        return;
    }
    float* base = gg.data + (int64_t)(cur & 0x1fffffffu) * 32 + dxb * 32 + ch;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!((kmask >> k) & 1)) continue;
        const bool ok = mine && okx && (!(k & 1) || oky) && (!(k >> 1) || okz);
        if (ok && st.acc[k] != 0.f) atomicAdd(base + (k & 1) * rowy + (k >> 1) * rowz, st.acc[k]);
        if (mine) st.acc[k] = 0.f;
    }
}
template <int P0>
ENS_DEV void scatter_piece_w(ScatterStW& st, const float (&val)[16], int ri, int rx, int ry, int rz, const DevGrid& gg,
                             const GradWin& w, int lane) {
    const int dxb = lane >> 5;
    const int stepy = gg.W, stepz = gg.H * gg.W;
#pragma unroll
    for (int pt = P0; pt < P0 + 4; ++pt) {
        const float v = val[pt];
        if (!__any(v != 0.f)) continue;
        const unsigned lin = (unsigned)__builtin_amdgcn_readlane(ri, pt);
        const float fx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rx, pt));
        const float fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ry, pt));
        const float fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rz, pt));
        if (!st.open) {
            st.cur = lin; st.open = true; st.woff = win_offset(lin, w);
        } else if (lin != st.cur) {                        // new cell: keep the partial sums of shared corner voxels (see scatter_tile_rec)
            const unsigned cur = st.cur;
            const int d = (int)(lin & 0x1fffffffu) - (int)(cur & 0x1fffffffu);
            const int nwoff = win_offset(lin, w);
            // (partial sums are carried over only between two cells on the same side of the window's edge)
            const bool carry = (nwoff >= 0) == (st.woff >= 0);
            if (carry && d == 1 && (cur >> 29 & 1u)) {
                scatter_flush_w(st, gg, w, lane, 15, 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float o = __shfl_xor(st.acc[k], 32); st.acc[k] = dxb ? 0.f : o; }
            } else if (carry && d == -1 && (lin >> 29 & 1u)) {
                scatter_flush_w(st, gg, w, lane, 15, 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float o = __shfl_xor(st.acc[k], 32); st.acc[k] = dxb ? o : 0.f; }
            } else if (carry && d == stepy && (cur >> 30 & 1u)) {
                scatter_flush_w(st, gg, w, lane, 5, 0);
                st.acc[0] = st.acc[1]; st.acc[2] = st.acc[3]; st.acc[1] = 0.f; st.acc[3] = 0.f;
            } else if (carry && d == -stepy && (lin >> 30 & 1u)) {
                scatter_flush_w(st, gg, w, lane, 10, 0);
                st.acc[1] = st.acc[0]; st.acc[3] = st.acc[2]; st.acc[0] = 0.f; st.acc[2] = 0.f;
            } else if (carry && d == stepz && (cur >> 31)) {
                scatter_flush_w(st, gg, w, lane, 3, 0);
                st.acc[0] = st.acc[2]; st.acc[1] = st.acc[3]; st.acc[2] = 0.f; st.acc[3] = 0.f;
            } else if (carry && d == -stepz && (lin >> 31)) {
                scatter_flush_w(st, gg, w, lane, 12, 0);
                st.acc[2] = st.acc[0]; st.acc[3] = st.acc[1]; st.acc[0] = 0.f; st.acc[1] = 0.f;
            } else {
                scatter_flush_w(st, gg, w, lane, 15, 0);
            }
            st.cur = lin;
            st.woff = nwoff;
        }
        const float wx = dxb ? fx : (1.f - fx);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float wt = (wx * ((k & 1) ? fy : (1.f - fy))) * ((k >> 1) ? fz : (1.f - fz));
            st.acc[k] = fmaf(wt, v, st.acc[k]);
        }
    }
}

template <int CT, int NOUT>
ENS_DEV void chain2_role(const BwdArgs& A, int kind, int wg, int n_wg, float* smem) {
    constexpr XyzLay L{CT * 16};
    constexpr int NW = C2_WAVES;                                    // waves of the workgroup
    constexpr bool HANDS = ENS_SCATTER_WHERE == 3;
    constexpr int NCW = HANDS ? C2_CHAIN_WAVES_3 : C2_WAVES;        // ... of which run the chain
    constexpr int WF = HANDS ? C2_WAVE_FLOATS_3 : C2_WAVE_FLOATS;
    constexpr int SCR0 = C2_WFLOATS + C2_IMG + (HANDS ? C2_SYNC + C2_WIN : 0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = lane & 15, q = lane >> 4;
    const float* __restrict__ pk = A.sc.packed[kind];
    float* gpk = A.gpacked[kind];
    const bool want_g = A.ggrid[kind].data != nullptr, want_r = A.g_ro != nullptr;
    const bool want_c = want_g || want_r;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)smem;
    const int slot_idx = kind - 1;                                  // decoder slot in the workspaces
    float* img = smem + C2_WFLOATS;

    // ---- resident weights (offsets as compile-time constants: left as calls, the layout functions stay calls in the kernel)
    static_for<5>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int dst = c2_off(i), oW = L.oWT(i), oC = L.oWcT(i), K = L.K(i);
        lds_fill<NW>(smem + dst, pk + oW, 8 * K, wave, lane);
        lds_fill<NW>(smem + dst + 32 * K, pk + oC, 256, wave, lane);
    });
    {
        constexpr int oBp = L.oBp(), oBT = L.oBT();
        lds_fill<NW>(smem + C2_BP, pk + oBp, 16 * 96 / 4, wave, lane);
        lds_fill<NW>(smem + C2_BT, pk + oBT, 96, wave, lane);
    }
    for (int e = threadIdx.x; e < C2_IMG + (HANDS ? C2_SYNC + C2_WIN : 0); e += NW * 64) img[e] = 0.f;      // (counters and window follow the image)
    int* const sync = reinterpret_cast<int*>(smem + C2_WFLOATS + C2_IMG);
    float woT[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { constexpr int oWoT = L.oWoT(); woT[rt] = pk[oWoT + (16 * rt + p) * 4 + q]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int64_t n_items = work_count_u(A);
    // mode 3: a workgroup takes a CONTIGUOUS chunk of the work list (rays of one camera, mostly: the window sits at their origin),
    // its chain wave w the items c0 + w, c0 + w + 8, ...; the other modes deal the items round-robin over all waves
    const int64_t chunk = HANDS ? (n_items + n_wg - 1) / n_wg : 0;
    const int64_t c0 = HANDS ? (int64_t)wg * chunk : (int64_t)wg * NCW;
    const int64_t c1 = HANDS ? (c0 + chunk < n_items ? c0 + chunk : n_items) : n_items;
    const int64_t stride = HANDS ? NCW : (int64_t)n_wg * NCW;
    GradWin gw{nullptr, 0, A.ggrid[kind].W, A.ggrid[kind].H * A.ggrid[kind].W};
    if constexpr (HANDS) {
        if (want_g && c0 < c1) {                                    // the window: around the origin of the chunk's first ray
            const int ray = work_tile_u(A, c0) / A.ntl;
            const double o[3] = {(double)A.ro[ray * 3], (double)A.ro[ray * 3 + 1], (double)A.ro[ray * 3 + 2]};
            const DevGrid g = A.sc.grid[kind];
            const Vox v = make_vox(o, A.sc.lo, A.sc.hi, g);
            const int wx0 = min(max(__builtin_amdgcn_readfirstlane(v.ix) - 1, 0), g.W - 4);
            const int wy0 = min(max(__builtin_amdgcn_readfirstlane(v.iy) - 1, 0), g.H - 4);
            const int wz0 = min(max(__builtin_amdgcn_readfirstlane(v.iz) - 1, 0), g.D - 4);
            gw.win = smem + C2_WFLOATS + C2_IMG + C2_SYNC;
            gw.lin0 = (wz0 * g.H + wy0) * g.W + wx0;
        }
    }
    auto finish_wg = [&]() {                                        // every wave of the workgroup: dB^T image and the window leave
        wg_barrier_lds();
        for (int e = threadIdx.x; e < C2_IMG; e += NW * 64) {
            const float v = img[e];
            constexpr int oBT = L.oBT();
            if (v != 0.f) atomicAdd(gpk + oBT + e, v);
        }
        if (HANDS && gw.win != nullptr) {
            for (int e = threadIdx.x; e < C2_WIN; e += NW * 64) {
                const float v = gw.win[e];
                const int vox = e >> 5, dx = vox & 3, dy = (vox >> 2) & 3, dz = vox >> 4;
                if (v != 0.f) atomicAdd(A.ggrid[kind].data + (int64_t)(gw.lin0 + dz * gw.HW + dy * gw.W + dx) * 32 + (e & 31), v);
            }
        }
    };
    if constexpr (HANDS) {
        if (wave >= NCW) {
            // ---- scatter wave: serves chain waves 2 s and 2 s + 1, tile by tile in their order; everything it reads comes from LDS
            if (want_g) {
                const int s0 = 2 * (wave - NCW);
                const int ch = lane & 31;
                int64_t cnt[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t it0 = c0 + s0 + j;
                    cnt[j] = it0 < c1 ? (c1 - it0 + stride - 1) / stride : 0;
                }
                const int64_t kmax = cnt[0] > cnt[1] ? cnt[0] : cnt[1];
                for (int64_t k = 0; k < kmax; ++k) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (k >= cnt[j]) continue;
                        const int w = s0 + j;
                        for (int spin = 0; spin < (1 << 22) && lds_poll(sync + w) < (int)k + 1; ++spin) __builtin_amdgcn_s_sleep(4);
                        asm volatile("" ::: "memory");
                        const float* hb = smem + SCR0 + w * WF + 6 * 256;
                        float val[16];
#pragma unroll
                        for (int pt = 0; pt < 16; ++pt) val[pt] = hb[pt * 32 + ch];
                        const f32x4 rec = *reinterpret_cast<const f32x4*>(hb + 512 + p * 4);
                        const float c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3];
                        const int ri = __builtin_bit_cast(int, c0), rx = __builtin_bit_cast(int, c1), ry = __builtin_bit_cast(int, c2),
                                  rz = __builtin_bit_cast(int, c3);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_store(sync + 8 + w, (int)k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // buffer free
                        ScatterStW st;
                        st.acc[0] = st.acc[1] = st.acc[2] = st.acc[3] = 0.f; st.cur = 0u; st.open = false; st.woff = -1;
                        scatter_piece_w<0>(st, val, ri, rx, ry, rz, A.ggrid[kind], gw, lane);
                        scatter_piece_w<4>(st, val, ri, rx, ry, rz, A.ggrid[kind], gw, lane);
                        scatter_piece_w<8>(st, val, ri, rx, ry, rz, A.ggrid[kind], gw, lane);
                        scatter_piece_w<12>(st, val, ri, rx, ry, rz, A.ggrid[kind], gw, lane);
                        if (st.open) scatter_flush_w(st, A.ggrid[kind], gw, lane, 15, 0);
                    }
                }
            }
            finish_wg();
            return;
        }
    }

    // ---- per-wave LDS bases
    const unsigned scr = lds0 + (SCR0 + wave * WF) * 4;
    unsigned dep[4];
    dep_bases(dep, scr, p, q);
    unsigned fbs = scr + frag_lane_off(lane);
    opaque(fbs);
    unsigned wsw = swz_base_even(lds0, 32, p, q);
    const unsigned swd = swz_odd_delta(p);
    unsigned wbp = swz_base_even(lds0 + C2_BP * 4, 96, p, q);
    unsigned wbt = lds0 + (C2_BT + p * 4 + q) * 4;
    opaque(wsw); opaque(wbp); opaque(wbt);

    constexpr int WSQ = (14 + CT) * 256;
    const float dscale = draw_scale_of(A);
    const bool handoff = want_r && A.dgrid_ws != nullptr;
    auto item_tile = [&](int64_t i) { return work_tile_u(A, i < n_items ? i : n_items - 1); };

    // loads of a tile are issued one tile ahead
    f32x4 draw_n = splat4(0.f), rec_n = splat4(0.f), bq_n = splat4(0.f);
    uint2 mw_n = make_uint2(0u, 0u);
    float pc_n = 0.f;
    constexpr bool SCAT = ENS_SCATTER_WHERE == 1;
    auto fetch = [&](int tile) {
        const float* __restrict__ w = A.act_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * ACT_STRIDE;
        draw_n = ld4(A.d_raw + ((int64_t)tile * 16 + p) * 4);
        mw_n = *reinterpret_cast<const uint2*>(w + ACT_MASK + lane * 2);
        if (q < 3) pc_n = w[WSQ + (p >> 2) * 64 + (q ^ (p >> 2)) * 4 + (p & 3)];      // coordinate q of sample p (swizzled tile)
        bq_n = ld4(w + WSQ + (lane ^ (lane >> 4)) * 4);             // the coordinates' fragment (dB^T operand), un-swizzled by address
        if ((SCAT || HANDS) && want_g) rec_n = ld4(w + ACT_VOX + p * 4);
    };
    float* const stg = smem + SCR0 + wave * WF;                     // dC as [sample][32]: scratch tiles 0, 1
    float* const hand = smem + SCR0 + wave * WF + 6 * 256;          // mode 3: the two hand-off buffers of this wave
    int n_done = 0;                                                 // tiles of this wave so far
    f32x4 rec_prev = splat4(0.f);
    bool pend = false;
    int64_t it = c0 + wave;
    int tile_cur = 0, tile_nxt = 0;                                 // work-list entries run two tiles ahead of their use
    if (it < c1) {
        tile_cur = item_tile(it);
        tile_nxt = item_tile(it + stride);
        fetch(tile_cur);
    }
    S2_DECL
    S2_START
    for (; it < c1; it += stride) {
        const int tile = tile_cur;
        tile_cur = tile_nxt;
        tile_nxt = item_tile(it + 2 * stride);
        f32x4 draw = draw_n * dscale, rec = rec_n, bq = bq_n;
        uint2 mw = mw_n;
        float pc = pc_n;
        // The prefetched values are taken into their own registers HERE, in front of this tile's atomics (left to the compiler
        // the copies sink behind them, and the wait for the loads becomes a wait for float atomics issued a moment ago)
        asm volatile("" : "+v"(draw), "+v"(mw.x), "+v"(mw.y), "+v"(rec), "+v"(pc), "+v"(bq) :: "memory");
        S2(0)       // prefetched loads have arrived
        if (it + stride < c1) fetch(tile_cur);
        ScatterSt sst;
        sst.acc[0] = sst.acc[1] = sst.acc[2] = sst.acc[3] = 0.f; sst.cur = 0u; sst.open = false;
        const float rp0 = rec_prev[0], rp1 = rec_prev[1], rp2 = rec_prev[2], rp3 = rec_prev[3];
        const int sri = __builtin_bit_cast(int, rp0), srx = __builtin_bit_cast(int, rp1), sry = __builtin_bit_cast(int, rp2),
                  srz = __builtin_bit_cast(int, rp3);
        // one 4-sample piece of the PREVIOUS tile's scatter (values from the staging tiles): in front of layers 4, 3, 2, 1, so that
        // the last atomics are issued two layers and the tail before the tile ends
        auto piece = [&](auto ic) {
            constexpr int P0 = decltype(ic)::value;
            if constexpr (SCAT) {
                if (pend) {
                    scatter_piece_lds<P0>(sst, stg, sri, srx, sry, srz, A.ggrid[kind], lane);
                    if constexpr (P0 == 12) {
                        if (sst.open) scatter_flush(sst, A.ggrid[kind], lane, 15, 0);
                        wave_lds_fence();
                    }
                }
            }
        };
        float* const dhw = A.dh_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * DH_STRIDE;
        float* const dgw = handoff ? A.dgrid_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * DG_STRIDE : nullptr;
        const unsigned mbits[5] = {mw.x & 255u, (mw.x >> 8) & 255u, (mw.x >> 16) & 255u, (mw.x >> 24) & 255u, mw.y & 255u};

        // ---- output layer: dh4 = Wo^T d_out (K = 4: one MFMA step per row tile; k-slot q carries output q)
        f32x4 dh[2] = {splat4(0.f), splat4(0.f)};
        {
            const float dq = NOUT == 4 ? (q == 0 ? draw[0] : (q == 1 ? draw[1] : (q == 2 ? draw[2] : 0.f)))
                                       : (q == 0 ? draw[3] : 0.f);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) dh[rt] = MFMA16(woT[rt], dq, dh[rt]);
        }
        f32x4 dc[2] = {splat4(0.f), splat4(0.f)};
        f32x4 demb[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) demb[t] = splat4(0.f);
        auto bwd_layer = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int WOFF = c2_off(i) * 4;
            const unsigned wb = wsw + WOFF;
            constexpr int OCT = 32 * L.K(i) * 4;                      // W_i^T [K][32] | Wc_i^T [32][32]
            const f32x4 dpre[2] = {mask4(dh[0], mbits[i], 0), mask4(dh[1], mbits[i], 4)};
            st4(dhw + (2 * i) * 256 + lane * 4, dh[0]);               // dh_i for the weight-gradient kernel (register layout)
            st4(dhw + (2 * i + 1) * 256 + lane * 4, dh[1]);
            if (want_c) lin_lds_swz<2, 2, 32, OCT>(dc, wb, swd, dh);                      // dC += Wc_i^T dh_i
            if constexpr (i == 0 || i == 3) {                         // rows 0..95 of W_i^T: the embedding part, in two halves
                lin_lds_swz_rows<0, 3, 2, 32, 0>(demb, wb, swd, dpre);
                lin_lds_swz_rows<3, 3, 2, 32, 0>(demb, wb, swd, dpre);
            }
            if constexpr (i == 3) {
                dh[0] = dh[1] = splat4(0.f);
                lin_lds_swz<2, 2, 32, 96 * 32 * 4>(dh, wb, swd, dpre);
            } else if constexpr (i != 0) {
                dh[0] = dh[1] = splat4(0.f);
                lin_lds_swz<2, 2, 32, 0>(dh, wb, swd, dpre);
            }
        };
        S2(1)       // next tile's loads issued, output layer
        piece(IC(0)); S2(2) bwd_layer(IC(4)); S2(3) piece(IC(4)); S2(2) bwd_layer(IC(3)); S2(3) piece(IC(8)); S2(2) bwd_layer(IC(2)); S2(3)
        piece(IC(12)); S2(2)
        bwd_layer(IC(1)); bwd_layer(IC(0));
        S2(3)       // (2: scatter pieces, 3: backward layers)
        pend = false;

        // ---- embedding: d_arg = d_emb * cos(arg);  dB^T += d_arg (x) p;  dp += B d_arg
#pragma unroll
        for (int t = 0; t < 6; ++t) {                               // cos(arg) recomputed: cheaper than carrying it
            const float a = *reinterpret_cast<const lds_float*>(static_cast<uintptr_t>(wbt + 16 * t * 16));
            const f32x4 arg = MFMA16(a, pc, splat4(0.f));
#pragma unroll
            for (int r = 0; r < 4; ++r) demb[t][r] *= ens_cosf(arg[r]);
        }
        f32x4 dpe[1] = {splat4(0.f)};
        if (want_r) lin_lds_swz<1, 6, 96, 0>(dpe, wbp, swd, demb);  // rows 0..2: dp (q == 0 lanes)
        S2(4)       // cos, d_arg, dp
        {
            // dB^T of this tile: d_arg through this wave's own LDS tiles (sample index into the MFMA K slot), six 16 x 16
            // products against the coordinates, summed into the workgroup's image by LDS float atomics (12 lanes of each carry a
            // value: nothing to keep in registers across tiles)
            dep_tile<0>(dep, demb[0]); dep_tile<1>(dep, demb[1]); dep_tile<2>(dep, demb[2]);
            dep_tile<3>(dep, demb[3]); dep_tile<4>(dep, demb[4]); dep_tile<5>(dep, demb[5]);
            wave_lds_fence();
            f32x4 fa[6], o[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) { fa[t] = lds4(fbs + t * 1024); o[t] = splat4(0.f); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < 6; ++t) o[t] = MFMA16(fa[t][s], bq[s], o[t]);
            }
            if (p < 3) {
#pragma unroll
                for (int t = 0; t < 6; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * t + 4 * q + r;
                        if (row < 93) lds_add(img + row * 4 + p, o[t][r]);
                    }
                }
            }
        }
        S2(5)       // dB^T
        if (handoff) {                  // hand-off to the ray-gradient role: dC (register layout) + embedding's position gradient
            st4(dgw + lane * 4, dc[0]);
            st4(dgw + 256 + lane * 4, dc[1]);
            st4(dgw + DG_DPE + lane * 4, dpe[0]);
        }
        if (want_g) {
            if constexpr (HANDS) {      // dC as [sample][32] and the cell records into the hand-off buffer of this tile's parity
                float* hb = hand;
                // (its previous content, the tile before, has been taken)
                for (int spin = 0; spin < (1 << 22) && lds_poll(sync + 8 + wave) < n_done; ++spin) __builtin_amdgcn_s_sleep(2);
                asm volatile("" ::: "memory");
                st4(hb + p * 32 + 4 * q, dc[0]);
                st4(hb + p * 32 + 16 + 4 * q, dc[1]);
                if (q == 0) st4(hb + 512 + p * 4, rec);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(sync + wave, n_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if constexpr (SCAT) {       // dC as [sample][32] in this wave's staging tiles (the dB^T fragments above have been read)
                wave_lds_fence();
                st4(stg + p * 32 + 4 * q, dc[0]);
                st4(stg + p * 32 + 16 + 4 * q, dc[1]);
                wave_lds_fence();
                rec_prev = rec;
                pend = true;
            } else {                    // ... or in the workspace, for the scatter in the weight-gradient kernel
                st4(dhw + DH_DC + p * 32 + 4 * q, dc[0]);
                st4(dhw + DH_DC + p * 32 + 16 + 4 * q, dc[1]);
            }
        }
        if (want_r && !handoff) {                                   // ray_grad_unit (raygrad.hpp) on this tile, in place
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
        S2(6)       // hand-off, staging
        ++n_done;
    }

    if (SCAT && pend) scatter_tile_rec(stg, rec_prev, A.ggrid[kind], lane);
    S2(7)           // last tile's scatter
    S2_FLUSH(0, wave, lane)

    // ---- dB^T: one flush of the workgroup's image (and of the gradient window)
    finish_wg();
}

extern __shared__ __attribute__((aligned(16))) float ens_smem2[];

__global__ __launch_bounds__(C2_WAVES * 64, 1) void decoder_chain_kernel(BwdArgs A) {
#if ENS_SCATTER_WHERE == 3
    if (threadIdx.x < C2_CHAIN_WAVES_3 * 64) __builtin_amdgcn_s_setprio(1);     // chain waves first, scatter waves in their gaps
#endif
    int role = 0;
#pragma unroll
    for (int r = 1; r < 4; ++r) role = (r < A.n_roles && (int)blockIdx.x >= A.role_begin[r]) ? r : role;
    const int wg = blockIdx.x - A.role_begin[role], n_wg = A.role_begin[role + 1] - A.role_begin[role];
    switch (A.role_kind[role]) {
        case 1: chain2_role<2, 1>(A, 1, wg, n_wg, ens_smem2); break;
        case 2: chain2_role<4, 1>(A, 2, wg, n_wg, ens_smem2); break;
        case 3: chain2_role<2, 4>(A, 3, wg, n_wg, ens_smem2); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------------
// Kernel 2: weight gradients, split-K over the stored operands (+ the feature-gradient scatter)
// ------------------------------------------------------------------------------------------------
// One ring stage (tiles of 1 KB): the forward's activation tiles in workspace order [EMB 6 | h2 2 | h0 2 | h1 2 | h3 2 | C ct],
// then the deposit tiles this kernel makes from register-layout tiles: dh_i (2 per layer), dpre_i (2 per layer), h4 (2) and
// the output gradient (1: rows 0..n_out-1 of a 16-row tile).
template <int CT>
struct DwLay {
    static constexpr int NA = 14 + CT;
    static constexpr int DH = NA, DP = NA + 10, H4 = NA + 20, DO = NA + 22, TILES = NA + 23;
    static constexpr int STAGE = TILES * 256;                        // floats
    static constexpr int NDMA = (NA + 7) / 8;                        // activation tiles per wave and fill
    static constexpr int NFILL = NDMA + 3;                           // vector-memory operations of one fill, every wave alike
};
constexpr int DW_STAGES = 3;
constexpr int lds_bytes_dw(int ct) { return cmax(DW_STAGES * (37 + ct) * 256, XyzLay{ct * 16}.fwd_floats()) * 4; }

// The 16 x 16 output tiles of a decoder's weight gradients are dealt to the 8 waves of a workgroup: wave w owns row tile
// rt = w & 1 of every matrix and the column tiles its quarter qd = w >> 1 is given below (7-8 products per wave with 32
// grid channels, 10 with 64).  A product (layer i, wc, ct) is  dW_i[rt][ct] += dpre_i[rt]^T x_i[ct]  (wc = 0)  or
// dWc_i[rt][ct] += dh_i[rt]^T c[ct]  (wc = 1).  The output layer (dWo = d_out^T h4: one row tile) goes to the rt = 0 waves
// of quarters 2 and 3, which own one product less.
struct DwProd { int i, wc, ct; };
struct DwProds { DwProd p[10]; int n; };
constexpr DwProds dw_prods(int CT, int qd) {
    DwProds P{};
    int n = 0;
    P.p[n++] = DwProd{3, 0, 2 * qd}; P.p[n++] = DwProd{3, 0, 2 * qd + 1};
    if (qd < 2) { P.p[n++] = DwProd{0, 0, 2 * qd}; P.p[n++] = DwProd{0, 0, 2 * qd + 1}; P.p[n++] = DwProd{4, 0, qd}; }
    else { P.p[n++] = DwProd{0, 0, qd + 2}; P.p[n++] = DwProd{1, 0, qd - 2}; P.p[n++] = DwProd{2, 0, qd - 2}; }
    for (int c = 0; c < CT; ++c) P.p[n++] = DwProd{qd, 1, c};
    if (CT == 4 || qd < 2) P.p[n++] = DwProd{4, 1, qd};
    P.n = n;
    return P;
}
// bias rows (db_i = sum dpre_i, dbc_i = sum dh_i) ride with a quarter that reads that operand anyway; slot or -1
constexpr int dw_bias_slot(int qd, int i, int wc) {
    if (qd == 0) return (wc == 0 && i == 3) ? 0 : (wc == 0 && i == 0) ? 1 : (wc == 0 && i == 4) ? 2 : (wc == 1 && i == 0) ? 3 : (wc == 1 && i == 4) ? 4 : -1;
    if (qd == 1) return (wc == 1 && i == 1) ? 0 : -1;
    if (qd == 2) return (wc == 0 && i == 1) ? 0 : (wc == 0 && i == 2) ? 1 : (wc == 1 && i == 2) ? 2 : -1;
    return (wc == 1 && i == 3) ? 0 : -1;
}
constexpr int dw_xbase(int i) { return i == 1 ? 8 : (i == 2 ? 10 : (i == 4 ? 12 : 0)); }   // x_0 = emb, x_3 = [emb | h2], x_1 = h0, x_2 = h1, x_4 = h3

// Two samples (pt0, pt0 + 1) of a tile's feature-gradient scatter: the state machine of scatter_piece (bwd_shared.hpp) with
// the sample index at run time; v0 / v1 = this lane's channel (lane & 31) of the two samples' dC.
ENS_DEV void scatter_two(ScatterSt& st, float v0, float v1, int pt0, int ri, int rx, int ry, int rz, const DevGrid& gg, int lane) {
    const int dxb = lane >> 5;
    const int stepy = gg.W, stepz = gg.H * gg.W;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float v = j == 0 ? v0 : v1;
        const int pt = pt0 + j;
        if (!__any(v != 0.f)) continue;
        const unsigned lin = (unsigned)__builtin_amdgcn_readlane(ri, pt);
        const float fx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rx, pt));
        const float fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ry, pt));
        const float fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rz, pt));
        if (!st.open) {
            st.cur = lin; st.open = true;
        } else if (lin != st.cur) {                        // new cell: keep the partial sums of shared corner voxels (see scatter_tile_rec)
            const unsigned cur = st.cur;
            const int d = (int)(lin & 0x1fffffffu) - (int)(cur & 0x1fffffffu);
            if (d == 1 && (cur >> 29 & 1u)) {
                scatter_flush(st, gg, lane, 15, 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float o = __shfl_xor(st.acc[k], 32); st.acc[k] = dxb ? 0.f : o; }
            } else if (d == -1 && (lin >> 29 & 1u)) {
                scatter_flush(st, gg, lane, 15, 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float o = __shfl_xor(st.acc[k], 32); st.acc[k] = dxb ? o : 0.f; }
            } else if (d == stepy && (cur >> 30 & 1u)) {
                scatter_flush(st, gg, lane, 5, 0);
                st.acc[0] = st.acc[1]; st.acc[2] = st.acc[3]; st.acc[1] = 0.f; st.acc[3] = 0.f;
            } else if (d == -stepy && (lin >> 30 & 1u)) {
                scatter_flush(st, gg, lane, 10, 0);
                st.acc[1] = st.acc[0]; st.acc[3] = st.acc[2]; st.acc[0] = 0.f; st.acc[2] = 0.f;
            } else if (d == stepz && (cur >> 31)) {
                scatter_flush(st, gg, lane, 3, 0);
                st.acc[0] = st.acc[2]; st.acc[1] = st.acc[3]; st.acc[2] = 0.f; st.acc[3] = 0.f;
            } else if (d == -stepz && (lin >> 31)) {
                scatter_flush(st, gg, lane, 12, 0);
                st.acc[2] = st.acc[0]; st.acc[3] = st.acc[1]; st.acc[0] = 0.f; st.acc[1] = 0.f;
            } else {
                scatter_flush(st, gg, lane, 15, 0);
            }
            st.cur = lin;
        }
        const float wx = dxb ? fx : (1.f - fx);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float w = (wx * ((k & 1) ? fy : (1.f - fy))) * ((k >> 1) ? fz : (1.f - fz));
            st.acc[k] = fmaf(w, v, st.acc[k]);
        }
    }
}

template <int CT, int NOUT, int QD>
ENS_DEV void dw2_body(const BwdArgs& A, int kind, int wg, int n_wg, float* smem, int rt) {
    constexpr XyzLay L{CT * 16};
    constexpr int GF = L.fwd_floats();
    using D = DwLay<CT>;
    constexpr int NP = dw_prods(CT, QD).n;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = lane & 15, q = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)smem;
    const int slot_idx = kind - 1;
    const bool own_wo = QD >= 2 && rt == 0;                         // this wave owns dWo[:, 16 (QD-2) ..] (and, QD == 2, dbo)
    f32x4 acc[NP], accWo = splat4(0.f);
    float accB[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, accBo = 0.f;
#pragma unroll
    for (int k = 0; k < NP; ++k) acc[k] = splat4(0.f);

    const DevGrid ggrid = A.ggrid[kind];
    const bool want_g = ENS_SCATTER_WHERE == 0 && ggrid.data != nullptr;
    const float dscale = draw_scale_of(A);
    const int64_t n_items = work_count_u(A);
    const int64_t my_n = n_items > wg ? (n_items - wg + n_wg - 1) / n_wg : 0;      // items wg, wg + n_wg, ...
    if (my_n > 0) {
        // item n of this workgroup (indices past the end repeat the last item: the pipeline below issues every fill and every
        // turn unconditionally so that the number of vector-memory operations between two waits is a compile-time constant)
        auto item_tile = [&](int64_t n) {
            const int64_t m = n < my_n ? n : my_n - 1;
            return work_tile_u(A, (int64_t)wg + m * n_wg);
        };
        // Register-layout tiles this wave turns into deposit tiles, two loads per fill for every wave (the counted vmcnt below
        // relies on equal counts): dh tile `wave`, and -- waves 0, 1: dh tiles 8, 9; waves 2, 3: h4; wave 4: d_raw (the output
        // gradient); waves 5..7: the first tile again (not used)
        const int j0 = wave;
        f32x4 r0[2], r1[2];
        uint2 mwr[2];
        unsigned dep[4];
        dep_bases(dep, lds0, p, q);
        unsigned fy = lds0 + frag_lane_off(lane) + rt * 1024, fx = lds0 + frag_lane_off(lane);
        opaque(fy); opaque(fx);

        auto fill = [&](int64_t n, int tile, auto rs) {             // item n (tile `tile`) -> stage n % 3, register set rs
            constexpr int R = decltype(rs)::value;
            float* stage = smem + (int)(n % DW_STAGES) * D::STAGE;
            const float* __restrict__ wsb = A.act_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * ACT_STRIDE;
            const float* __restrict__ dhb = A.dh_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * DH_STRIDE;
#pragma unroll
            for (int k = 0; k < D::NDMA; ++k) {
                int t = wave + 8 * k;
                t = t < D::NA ? t : wave;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsb + t * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void*)(stage + t * 256), 16, 0, 0);
            }
            r0[R] = ld4(dhb + j0 * 256 + lane * 4);
            const float* src1 = wave < 2 ? dhb + (8 + wave) * 256 + lane * 4
                              : (wave < 4 ? wsb + ACT_H4 + (wave - 2) * 256 + lane * 4
                              : (wave == 4 ? A.d_raw + ((int64_t)tile * 16 + p) * 4 : dhb + j0 * 256 + lane * 4));
            r1[R] = ld4(src1);
            mwr[R] = *reinterpret_cast<const uint2*>(wsb + ACT_MASK + lane * 2);
        };
        auto turn = [&](int64_t n, auto rs) {                       // register set rs -> deposit tiles in item n's stage
            constexpr int R = decltype(rs)::value;
            const unsigned sb = (unsigned)((int)(n % DW_STAGES) * D::STAGE * 4);
            auto dh_tile = [&](int j, const f32x4& v) {             // dh tile j = 2 i + rt': dh_i and dpre_i = relu'(.) dh_i
                const int i = j >> 1;
                const unsigned bits = ((i < 4 ? mwr[R].x >> (8 * i) : mwr[R].y) & 255u) >> (4 * (j & 1));
                const f32x4 pre = mask4(v, bits, 0);
                const unsigned o = sb + (unsigned)(D::DH + j) * 1024u;
#pragma unroll
                for (int r = 0; r < 4; ++r) { lds_st1(dep[r] + o, v[r]); lds_st1(dep[r] + o + 10 * 1024, pre[r]); }
            };
            dh_tile(j0, r0[R]);
            if (wave < 2) {
                dh_tile(8 + wave, r1[R]);
            } else if (wave < 4) {
                const unsigned o = sb + (unsigned)(D::H4 + wave - 2) * 1024u;
#pragma unroll
                for (int r = 0; r < 4; ++r) lds_st1(dep[r] + o, r1[R][r]);
            } else if (wave == 4) {
                // the output gradient as a register tile: sample p, "feature" 4q + r = output row (colour: r, g, b; else occupancy)
                const f32x4 d = r1[R] * dscale;
                f32x4 x = splat4(0.f);
                if (q == 0) x = NOUT == 4 ? f32x4{d[0], d[1], d[2], 0.f} : f32x4{d[3], 0.f, 0.f, 0.f};
                const unsigned o = sb + (unsigned)D::DO * 1024u;
#pragma unroll
                for (int r = 0; r < 4; ++r) lds_st1(dep[r] + o, x[r]);
            }
        };
        auto compute = [&](int64_t n) {
            const unsigned sb = (unsigned)((int)(n % DW_STAGES) * D::STAGE * 4);
            const unsigned by = fy + sb, bx = fx + sb;
            f32x4 a[NP], b[NP], ao = splat4(0.f), bo = splat4(0.f);
            static_for<NP>([&](auto ic) {
                constexpr int k = decltype(ic)::value;
                constexpr DwProd P = dw_prods(CT, QD).p[k];
                constexpr DwProd Q = dw_prods(CT, QD).p[k > 0 ? k - 1 : 0];
                constexpr bool same = k > 0 && Q.i == P.i && Q.wc == P.wc;
                if constexpr (same) a[k] = a[k > 0 ? k - 1 : 0];
                else a[k] = lds4(by + (unsigned)((P.wc ? D::DH : D::DP) + 2 * P.i) * 1024u);
                b[k] = lds4(bx + (unsigned)(P.wc ? 14 + P.ct : dw_xbase(P.i) + P.ct) * 1024u);
            });
            if constexpr (QD >= 2) {
                if (own_wo) { ao = lds4(bx + (unsigned)D::DO * 1024u); bo = lds4(bx + (unsigned)(D::H4 + QD - 2) * 1024u); }
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef ENS_EXP_DW_NOMFMA            // timing experiment (wrong results): the pipeline without its products
#pragma unroll
            for (int k = 0; k < NP; ++k) acc[k] += a[k] * b[k];
#else
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int k = 0; k < NP; ++k) acc[k] = MFMA16(a[k][s], b[k][s], acc[k]);
            }
            if constexpr (QD >= 2) {
                if (own_wo) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) accWo = MFMA16(ao[s], bo[s], accWo);
                    if constexpr (QD == 2) accBo += (ao[0] + ao[1]) + (ao[2] + ao[3]);
                }
            }
#endif
            static_for<NP>([&](auto ic) {
                constexpr int k = decltype(ic)::value;
                constexpr DwProd P = dw_prods(CT, QD).p[k];
                constexpr DwProd Q = dw_prods(CT, QD).p[k > 0 ? k - 1 : 0];
                constexpr bool same = k > 0 && Q.i == P.i && Q.wc == P.wc;
                constexpr int bs = dw_bias_slot(QD, P.i, P.wc);
                if constexpr (!same && bs >= 0) accB[bs] += (a[k][0] + a[k][1]) + (a[k][2] + a[k][3]);      // feature p, samples 4q..4q+3
            });
        };

        // ---- feature-gradient scatter: wave w scatters the tiles of items n = w (mod 8), two samples per step over the eight
        //      steps that follow (state carried in registers), so that every wave issues a few atomics per step and nobody
        //      carries a whole tile's burst
        ScatterSt sst;
        sst.acc[0] = sst.acc[1] = sst.acc[2] = sst.acc[3] = 0.f; sst.cur = 0u; sst.open = false;
        int sc_k = 8;                                               // next sub-piece of the tile in hand (8: none)
        const float* sc_dc = nullptr;                               // its dC [sample][32]
        int sri = 0, srx = 0, sry = 0, srz = 0;                     // its cell records (lane = sample)
        const int ch = lane & 31;
        // (the loads of a step -- the two samples' values, the cell records of a tile taken on -- are issued at the TOP of the step,
        // in front of the fill: what is younger than them when they are used is exactly one fill)
        float sv0 = 0.f, sv1 = 0.f;
        f32x4 srec = splat4(0.f);
        auto scatter_loads = [&](bool take, int tile) {
            if (sc_k < 8) { sv0 = sc_dc[(2 * sc_k) * 32 + ch]; sv1 = sc_dc[(2 * sc_k + 1) * 32 + ch]; }
            if (take) srec = ld4(A.act_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * ACT_STRIDE + ACT_VOX + p * 4);
        };
        auto scatter_step = [&]() {                                 // one sub-piece
            if (sc_k < 8) {
                scatter_two(sst, sv0, sv1, 2 * sc_k, sri, srx, sry, srz, ggrid, lane);
                ++sc_k;
                if (sc_k == 8 && sst.open) { scatter_flush(sst, ggrid, lane, 15, 0); sst.open = false; }
            }
        };
        auto scatter_take = [&](int tile) {                         // start on a new tile (the one in hand is finished: 8 steps ago)
            const float c0 = srec[0], c1 = srec[1], c2 = srec[2], c3 = srec[3];
            sri = __builtin_bit_cast(int, c0); srx = __builtin_bit_cast(int, c1); sry = __builtin_bit_cast(int, c2); srz = __builtin_bit_cast(int, c3);
            sc_dc = A.dh_ws + ((int64_t)tile * ACT_SLOTS + slot_idx) * DH_STRIDE + DH_DC;
            sc_k = 0;
            sst.acc[0] = sst.acc[1] = sst.acc[2] = sst.acc[3] = 0.f; sst.cur = 0u; sst.open = false;
        };

        // ---- software pipeline: fills run two items ahead (LDS-DMA + the register-layout loads), the deposit tiles of item
        //      n + 1 are made behind the products of item n, one workgroup barrier per item.
        int t_next = item_tile(2);                                  // the work-list entry of a fill is looked up one step ahead
        int t0 = item_tile(0), t1 = item_tile(1);
        fill(0, t0, IC(0));
        fill(1, t1, IC(1));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D::NFILL) : "memory");
        turn(0, IC(0));
        int t_cur = t0, t_nx1 = t1;                                 // tiles of items n and n + 1
        S2_DECL
        S2_START
        auto step = [&](int64_t n, auto par) {                      // par = n & 1: register set of item n + 1 is par ^ 1, of n + 2 par
            constexpr int PAR = decltype(par)::value;
            wg_barrier_lds();                                       // item n complete in LDS; everybody is done with item n - 1's stage
            S2(0)   // barrier
            const int tile = t_next;
            t_next = item_tile(n + 3);
            const bool take = want_g && (int)(n & 7) == wave;
            if (want_g) scatter_loads(take, t_cur);
            fill(n + 2, tile, IC(PAR));
            S2(1)   // fill issue
            compute(n);
            __builtin_amdgcn_sched_barrier(0);
            S2(2)   // fragment reads + products
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D::NFILL) : "memory");
            S2(3)   // wait for the fill of item n + 1
            turn(n + 1, IC(PAR ^ 1));
            S2(4)   // deposit tiles
            if (want_g) {
                scatter_step();
                if (take) scatter_take(t_cur);
            }
            S2(5)   // scatter
            t_cur = t_nx1; t_nx1 = tile;
        };
        int64_t n = 0;
        for (; n + 1 < my_n; n += 2) {
            step(n, IC(0));
            step(n + 1, IC(1));
        }
        if (n < my_n) step(n, IC(0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the surplus fills have landed: the flush image aliases the ring
        if (want_g) {                                               // the tiles still in hand
            for (int k = 0; k < 8; ++k) { scatter_loads(false, 0); scatter_step(); }
        }
        S2(6)       // drain
        S2_FLUSH(1, wave, lane)
    }

    // ---- flush: owned tiles -> packed-layout LDS image (aliases the ring) -> float atomics / partial rows
    float* sacc = smem;
    wg_barrier_lds();
    for (int e = threadIdx.x; e < GF; e += 512) sacc[e] = 0.f;
    wg_barrier_lds();
    static_for<NP>([&](auto ic) {
        constexpr int k = decltype(ic)::value;
        constexpr DwProd P = dw_prods(CT, QD).p[k];
        constexpr DwProd Q = dw_prods(CT, QD).p[k > 0 ? k - 1 : 0];
        constexpr bool same = k > 0 && Q.i == P.i && Q.wc == P.wc;
        constexpr int nc = P.wc ? CT : L.K(P.i) / 16;
        stage_tile(sacc + (P.wc ? L.oWc(P.i) : L.oW(P.i)), nc * 16, 0, nc, rt * nc + P.ct, acc[k], 32, nc * 16, p, q);
        constexpr int bs = dw_bias_slot(QD, P.i, P.wc);
        if constexpr (!same && bs >= 0) stage_bias_lane(sacc + (P.wc ? L.obc(P.i) : L.ob(P.i)), rt, accB[bs], 32, p, q);
    });
    if constexpr (QD >= 2) {
        if (own_wo) {
            stage_tile(sacc + L.oWo(), 32, 0, 2, QD - 2, accWo, NOUT, 32, p, q);
            if constexpr (QD == 2) stage_bias_lane(sacc + L.obo(), 0, accBo, NOUT, p, q);
        }
    }
    wg_barrier_lds();
#ifdef ENS_EXP_DW_NOFLUSH           // timing experiment (wrong results): no flush of the accumulated image
    if (sacc[threadIdx.x] == 12345.678f)
#endif
    flush_image(sacc, GF, A.gpacked[kind], A.gpart[kind], wg, n_wg, 512);
}

template <int CT, int NOUT>
ENS_DEV void dw2_role(const BwdArgs& A, int kind, int wg, int n_wg, float* smem) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rt = wave & 1;
    switch (wave >> 1) {
        case 0: dw2_body<CT, NOUT, 0>(A, kind, wg, n_wg, smem, rt); break;
        case 1: dw2_body<CT, NOUT, 1>(A, kind, wg, n_wg, smem, rt); break;
        case 2: dw2_body<CT, NOUT, 2>(A, kind, wg, n_wg, smem, rt); break;
        default: dw2_body<CT, NOUT, 3>(A, kind, wg, n_wg, smem, rt); break;
    }
}

__global__ __launch_bounds__(512, 1) void decoder_dw_kernel(BwdArgs A) {
    int role = 0;
#pragma unroll
    for (int r = 1; r < 4; ++r) role = (r < A.n_roles && (int)blockIdx.x >= A.role_begin[r]) ? r : role;
    const int wg = blockIdx.x - A.role_begin[role], n_wg = A.role_begin[role + 1] - A.role_begin[role];
    switch (A.role_kind[role]) {
        case 1: dw2_role<2, 1>(A, 1, wg, n_wg, ens_smem2); break;
        case 2: dw2_role<4, 1>(A, 2, wg, n_wg, ens_smem2); break;
        case 3: dw2_role<2, 4>(A, 3, wg, n_wg, ens_smem2); break;
        default: break;
    }
}

// The feature-gradient scatter as a launch of its own: one wave per (active tile, decoder).  The wave takes the tile's dC
// ([sample][32], left by the chain kernel) and cell records into registers FIRST, then runs the scatter state machine
// (scatter_piece, bwd_shared.hpp) and ends: nothing ever waits for an atomic, and the launch is as wide as the work list.
__global__ __launch_bounds__(256) void decoder_scatter_kernel(BwdArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t item = unit / A.n_roles;
    const int r = (int)(unit - item * A.n_roles);
    if (item >= work_count_u(A)) return;
    const int kind = A.role_kind[r];
    const DevGrid gg = A.ggrid[kind];
    if (gg.data == nullptr) return;
    const int tile = work_tile_u(A, item);
#ifdef ENS_EXP_SKIP_NEAR            // timing experiment (wrong results): no scatter for the tile next to the camera
    if (tile % A.ntl == 0) return;
#endif
#ifdef ENS_EXP_ONLY_NEAR            // timing experiment (wrong results): ONLY the tile next to the camera
    if (tile % A.ntl != 0) return;
#endif
    const int64_t blk = (int64_t)tile * ACT_SLOTS + (kind - 1);
    const f32x4 rec = ld4(A.act_ws + blk * ACT_STRIDE + ACT_VOX + (lane & 15) * 4);
    const float* __restrict__ dcw = A.dh_ws + blk * DH_STRIDE + DH_DC + (lane & 31);
    float val[16];
#pragma unroll
    for (int pt = 0; pt < 16; ++pt) val[pt] = dcw[pt * 32];
    const float c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3];
    const int ri = __builtin_bit_cast(int, c0), rx = __builtin_bit_cast(int, c1), ry = __builtin_bit_cast(int, c2), rz = __builtin_bit_cast(int, c3);
    ScatterSt st;
    st.acc[0] = st.acc[1] = st.acc[2] = st.acc[3] = 0.f; st.cur = 0u; st.open = false;
    scatter_piece<0>(st, val, ri, rx, ry, rz, gg, lane);
    scatter_piece<4>(st, val, ri, rx, ry, rz, gg, lane);
    scatter_piece<8>(st, val, ri, rx, ry, rz, gg, lane);
    scatter_piece<12>(st, val, ri, rx, ry, rz, gg, lane);
    if (st.open) scatter_flush(st, gg, lane, 15, 0);
}

// workgroups of a launch over its roles: a workgroup takes `per` tiles per pass, so a role with m workgroups needs
// ceil(groups / m) passes of relative cost c -- the split whose slowest role finishes first (as in render_bwd.hip)
void split_roles(int total, int groups, const float* cs, int n, int* split) {
    auto rounds = [&](int m) { return (groups + m - 1) / m; };
    if (n == 1) { split[0] = total; return; }
    float best = 1e30f;
    if (n == 2) {
        for (int a = 1; a < total; ++a) {
            const float t = fmaxf(rounds(a) * cs[0], rounds(total - a) * cs[1]);
            if (t < best) { best = t; split[0] = a; split[1] = total - a; }
        }
        return;
    }
    for (int a = 1; a < total - 1; ++a) {
        const float ta = rounds(a) * cs[0];
        if (ta >= best) continue;
        for (int b2 = 1; b2 < total - a; ++b2) {
            const float t = fmaxf(ta, fmaxf(rounds(b2) * cs[1], rounds(total - a - b2) * cs[2]));
            if (t < best) { best = t; split[0] = a; split[1] = b2; split[2] = total - a - b2; }
        }
    }
}

}  // namespace

#ifdef ENS_STAMPS
extern "C" int enslam_debug_set_stamp_buffer2(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf2), &v, sizeof(v)) == hipSuccess ? 0 : -2;
}
#endif

int ens_launch_decoder_bwd2(const BwdArgs& A0, const int* kinds, const float* costs, int n, int stage, int64_t n_tiles,
                            hipStream_t st) {
    if (n <= 0) return 0;
    if (A0.act_ws == nullptr || A0.act_light || A0.dh_ws == nullptr) return -1;
    static bool attr_done[ENS_MAX_DEVICES] = {};
    bool& attr_set = attr_done[ens_device_ordinal()];
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes_chain2()) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes_dw(4)) != hipSuccess)
            return -2;
        attr_set = true;
    }
    const int cus = device_cus();
    int ldsdw = 0;
    float cchain[3], cdw[3];
    for (int i = 0; i < n; ++i) {
        const int ct = kinds[i] == 2 ? 4 : 2;
        ldsdw = cmax(ldsdw, lds_bytes_dw(ct));
        cchain[i] = 1.f;                                            // the dX chain is the same size for the three decoders
        cdw[i] = (float)(24 + ct + 1);                              // KB streamed per tile
    }
    (void)costs; (void)stage;
    auto launch = [&](bool chain) -> int {
        const int per = chain ? (ENS_SCATTER_WHERE == 3 ? C2_CHAIN_WAVES_3 : C2_WAVES) : 1;
        const int groups = (int)((n_tiles + per - 1) / per);
        int total = cus;
        const int64_t max_useful = (int64_t)groups * n;
        if (total > max_useful) total = (int)max_useful;
        if (total < n) total = n;
        int split[3] = {0, 0, 0};
        split_roles(total, groups, chain ? cchain : cdw, n, split);
        BwdArgs B = A0;
        B.n_roles = n;
        int begin = 0;
        for (int i = 0; i < n; ++i) { B.role_kind[i] = kinds[i]; B.role_begin[i] = begin; begin += split[i]; }
        B.role_begin[n] = total;
        for (int i = n; i < 4; ++i) B.role_kind[i] = -1;
        if (chain) decoder_chain_kernel<<<dim3(total), dim3(C2_WAVES * 64), lds_bytes_chain2(), st>>>(B);
        else decoder_dw_kernel<<<dim3(total), dim3(512), ldsdw, st>>>(B);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    };
    const int r = launch(true);
    if (r != 0) return r;
    if (ENS_SCATTER_WHERE == 2) {
        bool any = false;
        for (int i = 0; i < n; ++i) any = any || A0.ggrid[kinds[i]].data != nullptr;
        if (any) {
            BwdArgs B = A0;
            B.n_roles = n;
            for (int i = 0; i < n; ++i) B.role_kind[i] = kinds[i];
            for (int i = n; i < 4; ++i) B.role_kind[i] = -1;
            const int64_t units = n_tiles * n;
            decoder_scatter_kernel<<<dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st>>>(B);
            if (hipGetLastError() != hipSuccess) return -2;
        }
    }
    return launch(false);
}
