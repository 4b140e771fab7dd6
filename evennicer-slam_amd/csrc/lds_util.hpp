// LDS helpers shared by the forward and backward kernels: explicit (base VGPR + immediate) addressing, the
// layer-weight ring filled by async global_load_lds, and MFMA linear layers reading their A fragments from LDS.
#pragma once
#include "common.hpp"

// ---- explicit LDS addressing -------------------------------------------------------------------
// Every LDS access of the xyz role is  (one per-lane base VGPR) + (compile-time byte offset): the offset folds
// into the instruction's 16-bit offset field, so no address registers pile up (left to itself the compiler
// hoists hundreds of loop-invariant addresses out of the tile loop, runs out of VGPRs and stops overlapping
// loads with MFMAs).  The bases are made opaque once per round so they cannot be re-expanded and hoisted.
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
ENS_DEV f32x4 lds4(unsigned addr) { return *reinterpret_cast<const lds_f32x4*>(static_cast<uintptr_t>(addr)); }
ENS_DEV void lds_st4(unsigned addr, const f32x4& v) { *reinterpret_cast<lds_f32x4*>(static_cast<uintptr_t>(addr)) = v; }
ENS_DEV void lds_st1(unsigned addr, float v) { *reinterpret_cast<lds_float*>(static_cast<uintptr_t>(addr)) = v; }
ENS_DEV void opaque(unsigned& v) { asm volatile("" : "+v"(v)); }

// acc[rt] += W[(16rt+i)*LD + 16t + 4q ..] * x[t]; W at byte offset OFF behind the lane base  base = (p*LD + 4q)*4
template <int NR, int KT, int LD, int OFF>
ENS_DEV void lin_lds(f32x4 (&acc)[NR], unsigned base, const f32x4 (&x)[KT]) {
    // all fragment reads first (independent, distinct registers), then the MFMAs with the accumulators
    // alternating so that back-to-back issues never wait on their own result
    f32x4 a[KT][NR];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) a[t][rt] = lds4(base + OFF + (16 * rt * LD + 16 * t) * 4);
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

// Forward weight images (W_i, Wc_i of the xyz decoders; 32 rows, LD = 16*KT columns) are stored TILE-MAJOR: for the row
// tile rt and the column tile t the 16 x 16 block is one 1 KB unit at float offset rt*16*LD + t*256, and inside it row p
// keeps its four 16-byte chunks together at 64*v + 16*u with v = (p >> 1) & 3 (which 256-byte bank row), u = (p & 1) +
// 2 * (p >> 3) (which quarter of it), the chunks XOR-ed with v.  The 16 lanes of every ds_read_b128 group then cover all
// 64 banks (a plain [row][LD] image puts them on 16: 4-way conflicts), and ONE lane base serves every LD because the tile
// strides are compile-time immediates.  pack_kernel writes this layout (segment flag 4), ring_load copies it verbatim.
ENS_DEV int frag_off(int p, int q) {                              // float offset of lane (p, q)'s chunk inside a 16 x 16 unit
    const int v = (p >> 1) & 3, u = (p & 1) + 2 * (p >> 3);
    return 64 * v + 16 * u + 4 * (q ^ v);
}
template <int NR, int KT, int LD, int OFF>
ENS_DEV void lin_lds_tm(f32x4 (&acc)[NR], unsigned base, const f32x4 (&x)[KT]) {
    f32x4 a[KT][NR];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) a[t][rt] = lds4(base + OFF + (16 * rt * LD + 256 * t) * 4);
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

// Bank-conflict-free images (the backward's transposed matrices, rows of 32 or 96 floats): the 16-byte chunk c of row r
// sits at chunk  c ^ ((r & 15) >> 1)  of its row (packed that way by pack_kernel, copied verbatim by ring_load).  A
// plain [row][32] image puts the 16 lanes of one ds_read_b128 group on 4 bank groups (4-way conflict); with the XOR
// they cover all 64 banks.  Lane (p, q) reads chunk 4t+q of row 16rt+p:  (4t+q) ^ s = (q ^ (s&3)) + (4t ^ (s&4)),
// so two lane bases (even / odd column tile) and immediates are enough.
ENS_DEV unsigned swz_base_even(unsigned region, int LD, int p, int q) {
    const int s = p >> 1;
    return region + (unsigned)(p * LD * 4 + ((q ^ (s & 3)) + (s & 4)) * 16);
}
ENS_DEV unsigned swz_odd_delta(int p) { return (unsigned)(((p >> 1) & 4) * 32); }      // bytes: base_odd = base_even - delta
template <int NR, int KT, int LD, int OFF>
ENS_DEV void lin_lds_swz(f32x4 (&acc)[NR], unsigned base_even, unsigned odd_delta, const f32x4 (&x)[KT]) {
    f32x4 a[KT][NR];
    const unsigned base_odd = base_even - odd_delta;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) a[t][rt] = lds4(((t & 1) ? base_odd : base_even) + OFF + (16 * rt * LD + 16 * t) * 4);
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

// cooperative async copy global -> LDS of n4 float4 (all 256 threads; 1 KB per wave instruction, no VGPR staging)
ENS_DEV void ring_load(float* dst_lds, const float* __restrict__ src, int n4, int wave, int lane) {
    for (int j = 0; j * 256 < n4; ++j) {
        const int e = j * 256 + wave * 64 + lane;
        if (e < n4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * e),
                                             (__attribute__((address_space(3))) void*)(dst_lds + 4 * (j * 256 + wave * 64)),
                                             16, 0, 0);
    }
}
