// Device helpers and argument block shared by the two translation units of the decoder backward (render_bwd.hip: the
// persistent chain + dW kernel and its 4-wave / light forms; render_bwd2.hip: the two-kernel form -- dX-chain kernel +
// split-K weight-gradient kernel).  Everything here is internal (anonymous namespace: one private copy per unit).
#pragma once
#include "common.hpp"
#include "kernels.hpp"
#include "lds_util.hpp"

struct BwdArgs {
    int n_rays, ntl;
    const float* ro;
    const float* rd;
    const double* z;
    const float* d_raw;
    const double* draw_scale;   // null, or a device scalar every d_raw value is multiplied by (unit gradients of a fused loss)
    const int* work;            // saved-activation path: work list of the tiles with non-zero d_raw (null: every tile) ...
    const int* n_work;          // ... and its length
    const float* act_ws;     // forward activations (render_fwd_kernel) or null: recompute
    int act_light;           // act_ws holds the light layout (coordinates | masks | cell records): light kernel only
    float* dgrid_ws;         // decoder -> grid_bwd_kernel hand-off (saved path): [tile][slot][DG_STRIDE]
    float* dh_ws;            // two-kernel form: dh_i of every layer, chain kernel -> weight-gradient kernel: [tile][slot][DH_STRIDE]
    DevScene sc;
    DevGrid ggrid[4];        // gradient accumulators (data may be null)
    float* gpacked[4];       // packed-layout gradient accumulators (may be null)
    float* gpart[4];         // per-workgroup partial images of the packed-layout gradients (null: float atomics into gpacked)
    float* g_ro;
    float* g_rd;
    int role_begin[5];       // workgroup ranges of the roles (decoder kinds) of this launch
    int role_kind[4];
    int n_roles;
    int defer_mask;          // bit k: decoder kind k hands its feature gradient (dC) off through dgrid_ws instead of scattering it
                             // (ens_launch_grid_scatter, grid_scatter.hip, follows the decoder launches)
};
// (render_bwd2.hip) the two-kernel form of the saved-activation backward for the roles of A (roles with parameter gradients)
int ens_launch_decoder_bwd2(const BwdArgs& A, const int* kinds, const float* costs, int n, int stage, int64_t n_tiles, hipStream_t st);

namespace {

constexpr int cmax(int a, int b) { return a > b ? a : b; }

ENS_DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

ENS_DEV void lds_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// End of a decoder role: the workgroup's packed-layout gradient image (GF floats in LDS) leaves the kernel either as float
// atomics into the one accumulator all workgroups share, or -- gpart given -- as plain 16-byte stores into this workgroup's
// own row of a partial buffer [16-float header | n_wg rows of GF floats] (header word 0 = n_wg, written by workgroup 0 of
// the role), which the finish launch sums while it unpacks (pack_body, util_kernels.hip).  Round 3: all 256 workgroups
// reach this point within a few microseconds of each other, and their 17.6 MB of atomics on the same 53 k addresses ran at
// the chip-wide memory-side atomic rate (1.3 TB/s): 10-13 us at the tail of the kernel (tools/stamps_bwd.py, budget).
ENS_DEV void flush_image(const float* sacc, int GF, float* gpk, float* gpart, int wg, int n_wg, int nthr) {
    if (gpart != nullptr) {
        if (wg == 0 && threadIdx.x == 0) reinterpret_cast<int*>(gpart)[0] = n_wg;
        f32x4* row = reinterpret_cast<f32x4*>(gpart + 16 + (int64_t)wg * GF);
        for (int e = threadIdx.x; e < GF / 4; e += nthr) row[e] = *reinterpret_cast<const f32x4*>(sacc + 4 * e);
        return;
    }
    for (int e = threadIdx.x; e < GF; e += nthr) {
        const float vsum = sacc[e];
        if (vsum != 0.f) atomicAdd(gpk + e, vsum);
    }
}


// stage an owned tile into the packed-layout LDS image (plain stores: every element has one owner)
ENS_DEV void stage_tile(float* sacc, int ld, int col0, int nc, int t, const f32x4& acc, int rows_valid, int cols_valid,
                        int p, int q) {
    const int rt = t / nc, ct = t - rt * nc;
    const int col = 16 * ct + p;
    if (col < cols_valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * q + r;
            if (row < rows_valid) sacc[row * ld + col0 + col] = acc[r];
        }
    }
}
// per-lane bias partials of own_layer_a: lane (p,q) holds feature 16rt+p summed over its samples 4q..4q+3
ENS_DEV void stage_bias_lane(float* sbias, int rt, float v, int n_valid, int p, int q) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (q == 0 && 16 * rt + p < n_valid) sbias[16 * rt + p] = v;
}
ENS_DEV void stage_bias(float* sbias, int rt, const f32x4& acc, int n_valid, int p, int q) {
    if (p == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * rt + 4 * q + r < n_valid) sbias[16 * rt + 4 * q + r] = acc[r];
    }
}


ENS_DEV f32x4 mask4(const f32x4& v, unsigned bits, int sh) {
    return f32x4{(bits >> sh) & 1u ? v[0] : 0.f, (bits >> (sh + 1)) & 1u ? v[1] : 0.f,
                 (bits >> (sh + 2)) & 1u ? v[2] : 0.f, (bits >> (sh + 3)) & 1u ? v[3] : 0.f};
}

// Same scatter from the cell records the forward saved (vox_record): rec = this lane's sample p (any q).
ENS_DEV void scatter_tile_rec(const float* dep, const f32x4& rec, const DevGrid& gg, int lane) {
    const int ch = lane & 31, dxb = lane >> 5;
    const int rowy = gg.W * 32, rowz = gg.H * gg.W * 32;
    const int stepy = gg.W, stepz = gg.H * gg.W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned cur = 0u;
    bool open = false;
    // add the accumulators selected by kmask (bit k = row dy + 2 dz) of the lanes selected by half (0: all, 1: voxel x,
    // 2: voxel x+1) of cell `cur` to the gradient and clear them
    auto flush = [&](int kmask, int half) {
        const bool okx = !dxb || (cur >> 29 & 1u), oky = cur >> 30 & 1u, okz = cur >> 31;
        const bool mine = half == 0 || (half == 1) == (dxb == 0);
        float* base = gg.data + (int64_t)(cur & 0x1fffffffu) * 32 + dxb * 32 + ch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!((kmask >> k) & 1)) continue;
            const bool ok = mine && okx && (!(k & 1) || oky) && (!(k >> 1) || okz);
            if (ok && acc[k] != 0.f) atomicAdd(base + (k & 1) * rowy + (k >> 1) * rowz, acc[k]);
            if (mine) acc[k] = 0.f;
        }
    };
    // (scalar copies first: __builtin_bit_cast applied to a vector-element lvalue reads element 0)
    const float r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
    const int ri = __builtin_bit_cast(int, r0), rx = __builtin_bit_cast(int, r1), ry = __builtin_bit_cast(int, r2),
              rz = __builtin_bit_cast(int, r3);
#pragma unroll
    for (int pt = 0; pt < 16; ++pt) {
        const float val = dep[pt * 32 + ch];
        if (!__any(val != 0.f)) continue;                 // e.g. masked samples of an occupancy decoder
        const unsigned lin = (unsigned)__builtin_amdgcn_readlane(ri, pt);
        const float fx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rx, pt));
        const float fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ry, pt));
        const float fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rz, pt));
        if (!open) {
            cur = lin; open = true;
        } else if (lin != cur) {                          // scalar comparison: new cell
            // A ray mostly steps into a face neighbour, which shares 4 of the 8 corner voxels: their partial sums stay in
            // the registers (moved to the rows / voxel column they have in the new cell) and only the face left behind
            // is added to the gradient -- about half of the atomics of a full flush per cell.  The neighbour flags of
            // the records guard against index steps that wrap around a row or a slice.
            const int d = (int)(lin & 0x1fffffffu) - (int)(cur & 0x1fffffffu);
            if (d == 1 && (cur >> 29 & 1u)) {             // x + 1: old voxel column x+1 becomes column x
                flush(15, 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float o = __shfl_xor(acc[k], 32); acc[k] = dxb ? 0.f : o; }
            } else if (d == -1 && (lin >> 29 & 1u)) {     // x - 1: old column x becomes column x+1
                flush(15, 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float o = __shfl_xor(acc[k], 32); acc[k] = dxb ? o : 0.f; }
            } else if (d == stepy && (cur >> 30 & 1u)) {  // y + 1
                flush(5, 0);
                acc[0] = acc[1]; acc[2] = acc[3]; acc[1] = 0.f; acc[3] = 0.f;
            } else if (d == -stepy && (lin >> 30 & 1u)) { // y - 1
                flush(10, 0);
                acc[1] = acc[0]; acc[3] = acc[2]; acc[0] = 0.f; acc[2] = 0.f;
            } else if (d == stepz && (cur >> 31)) {       // z + 1
                flush(3, 0);
                acc[0] = acc[2]; acc[1] = acc[3]; acc[2] = 0.f; acc[3] = 0.f;
            } else if (d == -stepz && (lin >> 31)) {      // z - 1
                flush(12, 0);
                acc[2] = acc[0]; acc[3] = acc[1]; acc[0] = 0.f; acc[1] = 0.f;
            } else {
                flush(15, 0);
            }
            cur = lin;
        }
        const float wx = dxb ? fx : (1.f - fx);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float w = (wx * ((k & 1) ? fy : (1.f - fy))) * ((k >> 1) ? fz : (1.f - fz));
            acc[k] = fmaf(w, val, acc[k]);
        }
    }
    if (open) flush(15, 0);
}



ENS_DEV float draw_scale_of(const BwdArgs& A) { return A.draw_scale != nullptr ? (float)A.draw_scale[0] : 1.f; }
// number of work items of the saved-activation roles and the tile behind item v (v < count)
// Wave-uniform look-ups of the work list go through the scalar cache (a load from the constant address space: s_load_dword,
// tracked by lgkmcnt).  As a vector load the look-up needs a vmcnt(0) before its value can steer the next loads -- which also
// drains every LDS-DMA transfer, store and float atomic the wave has in flight.  The list is written by an earlier launch;
// nothing in these kernels writes it.
typedef __attribute__((address_space(4))) const int ens_cint;
ENS_DEV int uload(const int* p) { return *reinterpret_cast<ens_cint*>(reinterpret_cast<uintptr_t>(p)); }
ENS_DEV int64_t work_count(const BwdArgs& A) { return A.work != nullptr ? (int64_t)uload(A.n_work) : (int64_t)A.n_rays * A.ntl; }
ENS_DEV int64_t work_tile(const BwdArgs& A, int64_t v) { return A.work != nullptr ? (int64_t)uload(A.work + v) : v; }


// The same scatter as a resumable state machine: the dW waves of decoder_bwd_split_kernel run it in four 4-sample pieces
// between their owned products.  val[pt] = this lane's channel (lane & 31) of sample pt's feature gradient; the cell records
// come as four wave-wide integers (lane = sample).
struct ScatterSt {
    float acc[4];
    unsigned cur;
    bool open;
};
ENS_DEV void scatter_flush(ScatterSt& st, const DevGrid& gg, int lane, int kmask, int half) {
    const int ch = lane & 31, dxb = lane >> 5;
    const int rowy = gg.W * 32, rowz = gg.H * gg.W * 32;
    const unsigned cur = st.cur;
    const bool okx = !dxb || (cur >> 29 & 1u), oky = cur >> 30 & 1u, okz = cur >> 31;
    const bool mine = half == 0 || (half == 1) == (dxb == 0);
    float* base = gg.data + (int64_t)(cur & 0x1fffffffu) * 32 + dxb * 32 + ch;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!((kmask >> k) & 1)) continue;
        const bool ok = mine && okx && (!(k & 1) || oky) && (!(k >> 1) || okz);
#ifdef ENS_EXP_NO_ATOMICS           // timing experiment (wrong results): the scatter's code without its atomics
        if (ok && st.acc[k] == 12345.678f) atomicAdd(base + (k & 1) * rowy + (k >> 1) * rowz, st.acc[k]);
#else
        if (ok && st.acc[k] != 0.f) atomicAdd(base + (k & 1) * rowy + (k >> 1) * rowz, st.acc[k]);
#endif
        if (mine) st.acc[k] = 0.f;
    }
}
template <int P0>
ENS_DEV void scatter_piece(ScatterSt& st, const float (&val)[16], int ri, int rx, int ry, int rz, const DevGrid& gg, int lane) {
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

// scatter_piece with the values read from the [sample][32] staging tiles in LDS (no 16 value registers): the chain kernel of
// render_bwd2.hip runs the previous tile's scatter in four such pieces between its backward layers, so that no more than a
// piece's atomics (7-15) are issued in one go -- a whole tile's 30-60 in a burst exceed what a wave may have outstanding
// (16-32) and stall its MFMA chain behind them.
template <int P0>
ENS_DEV void scatter_piece_lds(ScatterSt& st, const float* dep, int ri, int rx, int ry, int rz, const DevGrid& gg, int lane) {
    const int dxb = lane >> 5, ch = lane & 31;
    const int stepy = gg.W, stepz = gg.H * gg.W;
#pragma unroll
    for (int pt = P0; pt < P0 + 4; ++pt) {
        const float v = dep[pt * 32 + ch];
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

// workgroup barrier that orders LDS traffic only (no vmcnt(0): global stores / atomics in flight are nobody's business here)
ENS_DEV void wg_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// Deposit tiles are SWIZZLED (as the forward's workspace tiles, which global_load_lds copies verbatim): the 16-byte chunk
// f = 4q + r (feature f, samples 4P..4P+3) of sample group P = p >> 2 sits at chunk f ^ P of its 64-float row.  Plainly
// laid out, the 32 lanes of a ds_write_b32 group land on 8 banks (4-way conflict: 4.6 M conflict cycles per launch, a
// third of the kernel's LDS time); with the XOR they cover all 32, and the fragment read of lane L = 16P + f -- chunk
// f ^ P, i.e. byte (L ^ (L >> 4)) * 16 of the tile -- stays conflict-free.
// dep[r] = slot base + (P*64 + (p&3) + 16q + 4*(r ^ P)) * 4: one lane base per register component, tile T at +T*1024.
ENS_DEV void dep_bases(unsigned (&dep)[4], unsigned slot_base, int p, int q) {
    const int P = p >> 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) { dep[r] = slot_base + (unsigned)(P * 64 + (p & 3) + 16 * q + 4 * (r ^ P)) * 4u; opaque(dep[r]); }
}
ENS_DEV unsigned frag_lane_off(int lane) { return (unsigned)(lane ^ (lane >> 4)) * 16u; }      // byte offset of lane's fragment in a tile
template <int T>
ENS_DEV void dep_tile(const unsigned (&dep)[4], const f32x4& x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_st1(dep[r] + T * 1024, x[r]);
}

int device_cus() { return ens_device_cus(); }


}  // namespace
