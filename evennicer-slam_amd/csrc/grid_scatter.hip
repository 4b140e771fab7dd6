// Feature-gradient scatter of the saved-activation backward as a launch of its own -- round 4, opt-in (ENSLAM_DEFER_SCATTER=1: always,
// =2: when a decoder without parameter gradients has a grid gradient; DESIGN.md section 6.3 has the numbers: it wins on a random-init
// map with the reference mapper's gradient set, loses on a map with surfaces and wherever the persistent kernel can hide the atomics).
//
// The decoder backward leaves d(loss)/d(interpolated feature) of every (tile, decoder) in the hand-off workspace the ray-gradient
// role already reads (dgrid_ws: dC in register layout).  What is left is the transpose of the forward's trilinear gather
// (Renderer.py:168-175 / common.py:342-357 through autograd: grid_sampler_3d_backward's scatter-add into the 8 corner voxels).
// Inside the decoder kernels that is ~200 k float atomics of 256 B per 1000-ray step (50 MB at the chip-wide ~1.3 TB/s of
// memory-side float atomics): 35 us of the persistent kernel's 135 (it runs in 100 us without them).
//
// Here the sums that share a destination are formed ON CHIP first:
//   * the rays of a batch are ordered along a Morton curve (key: the cell, at 1/128 of the bound, of the point one tenth of
//     the bound's extent along the ray), in chunks of 1024 -- every workgroup sorts the chunk of its group itself (bitonic:
//     shuffles inside a wave, LDS across waves; no extra launch, no buffer, nothing for the host to provide);
//   * one workgroup takes 12 consecutive rays of that order and ONE grid (252 workgroups at 1000 rays: every CU; 16 rays: 189): neighbouring rays cross the same cells, so the
//     ~6 000 corner contributions of the group fall on ~500 distinct voxel rows (tools/sim_scatter_merge.py) -- they are
//     added into an LDS table of 576 rows x 32 channels (open addressing on the voxel index) and leave once per row:
//     52 k row-adds (6.6 MB) per step instead of ~400 k (50 MB);
//   * LDS float atomics are unusable for this (ds_add_f32: ~160 cycles per wave instruction on gfx950, measured; integer LDS
//     atomics ~9): the table holds 64-bit FIXED-POINT sums (scale: a power of two from the group's largest |dC|; exact and
//     order-independent above 2^-40 of that value, FLUSHED TO ZERO below it -- the one place where this path differs from
//     float32 accumulation: gradient elements some 12 decades below the group's largest come out as exact zeros);
//   * a row that finds no place in the table (8 probes) is added to the gradient directly, and a group that holds an inf / NaN
//     adds everything directly -- correctness does not depend on the table's size or on the order of the rays.
// Measured (1000 rays, room0, random-init map): 50 us with 12 rays per workgroup (60 with 16).  s_memtime stamps (tools/stamps_scatter.py,
// 16-ray form, cycles per wave): table clear 2.5 k, ordering 12.7 k, loads + geometry 9.7 k, scale 3.3 k, table probes 19 k (they return
// through the LDS queue the other waves' adds sit in), adds 13.7 k, waiting for the slowest wave 28 k, flush 3.5 k.
#include "common.hpp"
#include "kernels.hpp"
#include "raygrad.hpp"
#include "stamps.hpp"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifndef ENS_GS_G
#define ENS_GS_G 12
#endif
constexpr int GS_G = ENS_GS_G;           // rays per workgroup
constexpr int GS_CHUNK = (1024 / GS_G) * GS_G;   // rays ordered together (whole groups, at most one key per thread)
constexpr int GS_ROWS = 576;             // table rows (the largest group of the bench scene needs 570)
constexpr int GS_WAVES = 16;
constexpr int GS_THREADS = GS_WAVES * 64;
#ifndef ENS_GS_PROBES
#define ENS_GS_PROBES 8
#endif
constexpr int GS_PROBES = ENS_GS_PROBES;
constexpr int GS_MAXU = 4;               // units per wave: GS_G * ntl / GS_WAVES, ntl <= 4
constexpr unsigned GS_EMPTY = 0xFFFFFFFFu;
static_assert(GS_CHUNK <= GS_THREADS && GS_G * 4 <= GS_MAXU * GS_WAVES, "one key per thread; units per wave");

// LDS map (bytes)
constexpr int GS_VALS = 0;                                   // [GS_ROWS + 1][32] int64 fixed point (row GS_ROWS: dummy, never read)
constexpr int GS_KEYS = GS_VALS + (GS_ROWS + 1) * 256;       // [GS_ROWS] voxel row index or GS_EMPTY
constexpr int GS_SORT = GS_KEYS + GS_ROWS * 4;               // [GS_CHUNK] u32: bin counts of the counting sort
constexpr int GS_ORD = GS_SORT + GS_THREADS * 4;               // [GS_CHUNK] int: chunk position -> ray
constexpr int GS_RAYS = GS_ORD + GS_THREADS * 4;             // [GS_G] ray of the group (-1: none)
constexpr int GS_RED = GS_RAYS + 64;                         // [GS_WAVES] max |dC| per wave; before: key bounding box [8] + wave totals [16]
constexpr int GS_BYTES = GS_RED + 96;
static_assert(GS_BYTES <= 160 * 1024, "LDS");
static_assert(GS_ROWS % (2 * GS_WAVES) == 0 && GS_ROWS <= GS_THREADS, "flush / key init");

struct ScatterArgs {
    int n_rays, ntl, n_slots;
    const float* ro;
    const float* rd;
    const double* z;
    const float* dgrid_ws;       // [tile][ACT_SLOTS][DG_STRIDE]
    const float* d_raw;          // [tile * 16][4]: tiles with all-zero d_raw were never handed off (work list); null: every tile was
    double lo[3], hi[3];
    float key_t;                 // ray parameter of the key point
    DevGrid ggrid[3];            // gradient of slot 0..2 (voxel-major; data null: nothing to scatter)
    int slot_of[3];              // slots with a gradient, in workgroup order
    int n_active_slots;
};

ENS_DEV unsigned spread7(unsigned x) {           // 7 bits -> every third bit
    x &= 0x7fu;
    x = (x | (x << 8)) & 0x0000700fu;
    x = (x | (x << 4)) & 0x000430c3u;
    x = (x | (x << 2)) & 0x00049249u;
    return x;
}

constexpr double GS_MAGIC = 6755399441055744.0;          // 1.5 * 2^52: (double)(x * S) + MAGIC has the integer in its low mantissa bits

__global__ __launch_bounds__(GS_THREADS) void grid_scatter_kernel(ScatterArgs A) {
    extern __shared__ __align__(16) unsigned char gs_lds[];
    unsigned long long* vals = reinterpret_cast<unsigned long long*>(gs_lds + GS_VALS);
    unsigned* keys = reinterpret_cast<unsigned*>(gs_lds + GS_KEYS);
    int* rays = reinterpret_cast<int*>(gs_lds + GS_RAYS);
    float* red = reinterpret_cast<float*>(gs_lds + GS_RED);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_groups = (A.n_rays + GS_G - 1) / GS_G;
    const int group = blockIdx.x % n_groups, slot = A.slot_of[blockIdx.x / n_groups];
    const DevGrid gg = A.ggrid[slot];
    const int chunk = group / (GS_CHUNK / GS_G);
    const int r0 = chunk * GS_CHUNK;
    const int n_chunk = min(GS_CHUNK, A.n_rays - r0);

    STAMP_DECL
    STAMP_START
    // ---- A. empty table; order of the chunk's rays.  Exact order is not needed -- only that the 16 rays of a group are neighbours:
    //         a counting sort on 1024 Morton bins of the key points inside THEIR bounding box (one LDS atomic per ray, one scan).
#pragma unroll
    for (int i = 0; i < (GS_ROWS + 1) * 256 / 16 / GS_THREADS; ++i)
        reinterpret_cast<f32x4*>(gs_lds + GS_VALS)[i * GS_THREADS + tid] = splat4(0.f);
    if (tid < ((GS_ROWS + 1) * 256 / 16) % GS_THREADS)
        reinterpret_cast<f32x4*>(gs_lds + GS_VALS)[((GS_ROWS + 1) * 256 / 16 / GS_THREADS) * GS_THREADS + tid] = splat4(0.f);
    if (tid < GS_ROWS) keys[tid] = GS_EMPTY;
    unsigned* hist = reinterpret_cast<unsigned*>(gs_lds + GS_SORT);          // [GS_CHUNK] bin counts, then their exclusive prefix
    int* ordr = reinterpret_cast<int*>(gs_lds + GS_ORD);                     // [GS_CHUNK] ray of the chunk at each position (-1: none)
    unsigned* bb = reinterpret_cast<unsigned*>(gs_lds + GS_RED);             // bounding box of the key points (float bits: all >= 0)
    unsigned* wtot = bb + 8;
    const bool kvalid = tid < n_chunk;
    float ku[3] = {0.f, 0.f, 0.f};
    if (kvalid) {
        const int r = r0 + tid;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float pa = A.ro[r * 3 + a] + A.rd[r * 3 + a] * A.key_t;
            float u = (pa - (float)A.lo[a]) / (float)(A.hi[a] - A.lo[a]);
            u = u >= 0.f ? u : 0.f;                                           // (NaN -> 0)
            ku[a] = u <= 1.f ? u : 1.f;
        }
    }
    hist[tid] = 0u;
    ordr[tid] = -1;
    if (tid < 3) { bb[tid] = 0x7f800000u; bb[4 + tid] = 0u; }
    STAMP(0)
    __syncthreads();
#ifndef ENS_EXP_GS_NOSORT
    if (n_chunk > GS_G) {
        {   // (one atomic per wave and bound: 1024 same-address LDS atomics cost 20 us)
            unsigned mn[3], mx[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                mn[a] = kvalid ? __builtin_bit_cast(unsigned, ku[a]) : 0x7f800000u;
                mx[a] = kvalid ? __builtin_bit_cast(unsigned, ku[a]) : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { mn[a] = min(mn[a], (unsigned)__shfl_xor(mn[a], o)); mx[a] = max(mx[a], (unsigned)__shfl_xor(mx[a], o)); }
            }
            if (lane < 3) { atomicMin(&bb[lane], lane == 0 ? mn[0] : lane == 1 ? mn[1] : mn[2]); atomicMax(&bb[4 + lane], lane == 0 ? mx[0] : lane == 1 ? mx[1] : mx[2]); }
        }
        __syncthreads();
        unsigned bin = 0;
        {
            unsigned qa[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float lo_a = __builtin_bit_cast(float, bb[a]), hi_a = __builtin_bit_cast(float, bb[4 + a]);
                const float ext = hi_a - lo_a;
                const float t = ext > 0.f ? (ku[a] - lo_a) / ext : 0.f;
                const int nb = a == 0 ? 16 : 8;
                int qi = (int)(t * (float)nb);
                qa[a] = (unsigned)(qi < 0 ? 0 : (qi >= nb ? nb - 1 : qi));
            }
            bin = ((qa[0] >> 3) << 9) | spread7(qa[0] & 7u) | (spread7(qa[1]) << 1) | (spread7(qa[2]) << 2);
        }
        const unsigned in_bin = kvalid ? atomicAdd(&hist[bin], 1u) : 0u;
        __syncthreads();
        const unsigned cnt = hist[tid];
        unsigned inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        unsigned base = 0;
#pragma unroll
        for (int w = 0; w < GS_WAVES; ++w) base += w < wave ? wtot[w] : 0u;
        __syncthreads();                                                      // (every thread has read its count)
        hist[tid] = base + inc - cnt;
        __syncthreads();
        if (kvalid) ordr[hist[bin] + in_bin] = tid;
        __syncthreads();
        // the order inside a bin is the atomics' arrival order so far -- different in every workgroup, and the groups only partition
        // the rays if all workgroups agree: rank the bin's rays by index
        int mypos = 0;
        if (kvalid) {
            const int s0 = (int)hist[bin], n = (int)(bin + 1 < GS_THREADS ? hist[bin + 1] : (unsigned)n_chunk) - s0;
            int c = 0;
            for (int i = 0; i < n; ++i) c += ordr[s0 + i] < tid ? 1 : 0;
            mypos = s0 + c;
        }
        __syncthreads();
        if (kvalid) ordr[mypos] = tid;
    } else
#endif
    {
        if (kvalid) ordr[tid] = tid;
    }
    __syncthreads();
    if (tid < GS_G) {
        const int k = ordr[(group % (GS_CHUNK / GS_G)) * GS_G + tid];
        rays[tid] = k < 0 ? -1 : r0 + k;
    }
    __syncthreads();

    STAMP(1)
    // ---- B. this wave's units (ray of the group, tile index): everything they need from memory, all units at once
    const int p = lane & 15, q = lane >> 4;
    const int ch = lane & 31, dxb = lane >> 5;
    const int n_units = GS_G * A.ntl;
    int64_t tile[GS_MAXU];
    bool act[GS_MAXU];
    int cidx[GS_MAXU][2], ccell[GS_MAXU];
    float cw[GS_MAXU][2];
    float amax = 0.f;
    {
        typedef __attribute__((address_space(4))) const float cfloat;        // wave-uniform reads through the scalar cache
        f32x4 dr[GS_MAXU];
        int ray[GS_MAXU];
#pragma unroll
        for (int i = 0; i < GS_MAXU; ++i) {
            const int u = wave + GS_WAVES * i;
            ray[i] = __builtin_amdgcn_readfirstlane(u < n_units ? rays[u % GS_G] : -1);
            act[i] = ray[i] >= 0;
            tile[i] = act[i] ? (int64_t)ray[i] * A.ntl + u / GS_G : 0;
            dr[i] = f32x4{1.f, 0.f, 0.f, 0.f};
            if (act[i] && A.d_raw != nullptr) dr[i] = ld4(A.d_raw + (tile[i] * 16 + p) * 4);
        }
        f32x4 d0[GS_MAXU], d1[GS_MAXU];
        double zz[GS_MAXU];
        float o3[GS_MAXU][3], d3[GS_MAXU][3];
#pragma unroll
        for (int i = 0; i < GS_MAXU; ++i) {
            act[i] = act[i] && __any(dr[i][0] != 0.f || dr[i][1] != 0.f || dr[i][2] != 0.f || dr[i][3] != 0.f);   // else: not in the work list, no hand-off
            d0[i] = d1[i] = splat4(0.f);
            zz[i] = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) { o3[i][a] = 0.f; d3[i][a] = 0.f; }
            if (act[i]) {
                const float* dgw = A.dgrid_ws + (tile[i] * ACT_SLOTS + slot) * DG_STRIDE;
                d0[i] = ld4(dgw + lane * 4); d1[i] = ld4(dgw + 256 + lane * 4);
                zz[i] = A.z[tile[i] * 16 + p];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    o3[i][a] = *reinterpret_cast<cfloat*>(reinterpret_cast<uintptr_t>(A.ro + ray[i] * 3 + a));
                    d3[i][a] = *reinterpret_cast<cfloat*>(reinterpret_cast<uintptr_t>(A.rd + ray[i] * 3 + a));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GS_MAXU; ++i) {
            float m = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaxf(fabsf(d0[i][r]), fabsf(d1[i][r])));
            const bool bad = !(m <= 3.0e38f);                                 // inf / NaN somewhere: keep float semantics (direct adds)
            if (__any(bad)) m = __builtin_inff();
            act[i] = act[i] && __any(m != 0.f);
            amax = fmaxf(amax, m);
            cidx[i][0] = cidx[i][1] = 0; cw[i][0] = cw[i][1] = 0.f; ccell[i] = 0;
            if (act[i]) {
                double pw[3];                                                 // tile_geo's point (raygrad.hpp), from the operands loaded above
#pragma unroll
                for (int a = 0; a < 3; ++a) pw[a] = (double)o3[i][a] + (double)d3[i][a] * zz[i];
                const Vox v = make_vox(pw, A.lo, A.hi, gg);
                ccell[i] = (v.iz * gg.H + v.iy) * gg.W + v.ix;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int64_t idx; float w;
                    const int c = q + 4 * j;
                    corner(v, gg, c, idx, w);
                    // a corner beyond the last voxel has no row (the same for every sample of the cell: runs of samples in one
                    // cell share their rows); a corner whose weight merely happens to be 0 keeps its row
                    const bool ok = (v.ix + (c & 1) < gg.W) && (v.iy + ((c >> 1) & 1) < gg.H) && (v.iz + (c >> 2) < gg.D);
                    cidx[i][j] = ok ? (int)idx : -1; cw[i][j] = w;
                }
            }
        }
    }
    STAMP(2)
    // ---- C. fixed-point scale of the workgroup: 2^e with n_samples * max|dC| * 2^e < 2^50 (weights are <= 1)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    float gmax = 0.f;
#pragma unroll
    for (int w = 0; w < GS_WAVES; ++w) gmax = fmaxf(gmax, red[w]);
    if (gmax == 0.f) return;                                                  // (uniform) nothing flows into this group's rows
    const bool direct = !(gmax <= 3.0e38f);
    int ex;
    (void)frexpf(direct ? 1.f : gmax, &ex);                                   // gmax < 2^ex
    const double scale = __builtin_ldexp(1.0, 50 - 10 - ex), inv_scale = __builtin_ldexp(1.0, ex + 10 - 50);   // (16 rays x 64 samples = 2^10)
    const long long magic_bits = __builtin_bit_cast(long long, GS_MAGIC);

    STAMP(3)
    // ---- D. unit by unit: table rows of the corners, then the adds
    // (a run-time loop: unrolled, the 4 x 64 add sites are 40 KB of code for 16 waves to share; the unit's registers are picked by selects)
    unsigned act_bits = 0;
#pragma unroll
    for (int i = 0; i < GS_MAXU; ++i) act_bits |= act[i] ? 1u << i : 0u;
#ifdef ENS_EXP_GS_NOUNITS
    act_bits = 0;
#endif
#pragma unroll 1
    for (int iu = 0; iu < GS_MAXU; ++iu) {
        if (!((act_bits >> iu) & 1u)) continue;                               // (wave-uniform)
        int64_t tile_u = tile[0];
        int cidx_u[2] = {cidx[0][0], cidx[0][1]}, ccell_u = ccell[0];
        float cw_u[2] = {cw[0][0], cw[0][1]};
#pragma unroll
        for (int i = 1; i < GS_MAXU; ++i)
            if (iu == i) { tile_u = tile[i]; cidx_u[0] = cidx[i][0]; cidx_u[1] = cidx[i][1]; ccell_u = ccell[i]; cw_u[0] = cw[i][0]; cw_u[1] = cw[i][1]; }
        const float* dgw = A.dgrid_ws + (tile_u * ACT_SLOTS + slot) * DG_STRIDE;
        // dC in channel layout (lane = channel, both halves): element (q' * 16 + pt) * 4 + r of the register layout, L2-warm
        float vv[16];
        const float* dch = dgw + (ch >> 4) * 256 + ((ch & 15) >> 2) * 64 + (ch & 3);
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) vv[pt] = dch[pt * 4];
        int rowj[2];                                                          // table row of this lane's corners q, q + 4 of sample p
        {
            unsigned key[2], h[2];
            bool need[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                key[j] = (unsigned)cidx_u[j];
                need[j] = cidx_u[j] >= 0 && !direct;
                rowj[j] = cidx_u[j] >= 0 ? -((int)key[j] + 2) : -1;
                h[j] = (((key[j] * 2654435761u) >> 22) * GS_ROWS) >> 10;
            }
            // first probe of both corners in flight together (most keys find their row there); the few that collide go on alone
            const unsigned o0 = need[0] ? atomicCAS(&keys[h[0]], GS_EMPTY, key[0]) : 0u;
            const unsigned o1 = need[1] ? atomicCAS(&keys[h[1]], GS_EMPTY, key[1]) : 0u;
            const unsigned oo[2] = {o0, o1};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!need[j]) continue;
                if (oo[j] == GS_EMPTY || oo[j] == key[j]) { rowj[j] = (int)h[j]; need[j] = false; }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll 1
                for (int t = 1; t < GS_PROBES && need[j]; ++t) {
                    h[j] = h[j] + 1 == GS_ROWS ? 0u : h[j] + 1;
                    const unsigned old = atomicCAS(&keys[h[j]], GS_EMPTY, key[j]);
                    if (old == GS_EMPTY || old == key[j]) { rowj[j] = (int)h[j]; need[j] = false; }
                }
            }
        }
        STAMP(4)
#ifndef ENS_EXP_GS_NOADD
        {   // The adds, back to back.  Rows, weights and cells reach the (channel, x-half) lanes by v_readlane: LDS operations
            // complete in order, so an LDS read issued behind the adds would wait for them.  Integer adds: ds_add_f32 runs at ~160
            // cycles per wave instruction on gfx950, ds_add_u64 at ~10 -- so the sums are kept in 64-bit fixed point (exact,
            // order-independent) and become float once, at the flush.
            bool ovf = false;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            int prow[4] = {GS_ROWS, GS_ROWS, GS_ROWS, GS_ROWS}, pcell = 0;
            auto emit = [&]() {                                               // the run of samples in cell pcell ends
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double d = __builtin_fma((double)acc[k], scale, GS_MAGIC);
                    const long long fx = __builtin_bit_cast(long long, d) - magic_bits;
                    __hip_atomic_fetch_add(vals + prow[k] * 32 + ch, (unsigned long long)fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    acc[k] = 0.f;
                }
            };
            const int wb0 = __builtin_bit_cast(int, cw_u[0]), wb1 = __builtin_bit_cast(int, cw_u[1]);
#pragma unroll
            for (int pt = 0; pt < 16; ++pt) {
                // consecutive samples of a ray in one cell (the surface samples) are summed in registers first
                const int cnow = __builtin_amdgcn_readlane(ccell_u, pt);
#ifdef ENS_EXP_GS_NORUN
                if (pt > 0) emit();
#else
                if (pt > 0 && cnow != pcell) emit();
#endif
                pcell = cnow;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // corner c = dxb + 2 k of sample pt: register j = k >> 1 of lane pt + 16 * (2 * (k & 1) + dxb)
                    const int src = pt + 32 * (k & 1);
                    const int r_lo = __builtin_amdgcn_readlane(k < 2 ? rowj[0] : rowj[1], src);
                    const int r_hi = __builtin_amdgcn_readlane(k < 2 ? rowj[0] : rowj[1], src + 16);
                    const int w_lo = __builtin_amdgcn_readlane(k < 2 ? wb0 : wb1, src);
                    const int w_hi = __builtin_amdgcn_readlane(k < 2 ? wb0 : wb1, src + 16);
                    const int row = dxb ? r_hi : r_lo;
                    const float w = __builtin_bit_cast(float, dxb ? w_hi : w_lo);
                    ovf = ovf || row <= -2;
                    prow[k] = row >= 0 ? row : GS_ROWS;
                    acc[k] = fmaf(w, vv[pt], acc[k]);
                }
            }
            emit();
            if (__any(ovf)) {                                                 // rows without a place in the table: straight to the gradient
#pragma unroll 1
                for (int pt = 0; pt < 16; ++pt) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int src = pt + 32 * (k & 1);
                        const int r_lo = __builtin_amdgcn_readlane(k < 2 ? rowj[0] : rowj[1], src);
                        const int r_hi = __builtin_amdgcn_readlane(k < 2 ? rowj[0] : rowj[1], src + 16);
                        const int w_lo = __builtin_amdgcn_readlane(k < 2 ? wb0 : wb1, src);
                        const int w_hi = __builtin_amdgcn_readlane(k < 2 ? wb0 : wb1, src + 16);
                        const int row = dxb ? r_hi : r_lo;
                        const float x = __builtin_bit_cast(float, dxb ? w_hi : w_lo) * dch[pt * 4];
                        if (row <= -2 && x != 0.f) atomicAdd(gg.data + (int64_t)(-(row + 2)) * 32 + ch, x);
                    }
                }
            }
        }
#endif
    }
    STAMP(5)
    __syncthreads();
    STAMP(6)
    // ---- E. the table leaves: two rows per wave instruction (two 128-byte segments: the full-rate float-atomic shape)
    {
        const int i0 = wave * (GS_ROWS / GS_WAVES);                            // 36 rows per wave
        unsigned kk[GS_ROWS / GS_WAVES / 2];
        long long xx[GS_ROWS / GS_WAVES / 2];
#pragma unroll
        for (int i = 0; i < GS_ROWS / GS_WAVES / 2; ++i) {
            kk[i] = keys[i0 + 2 * i + dxb];
            xx[i] = (long long)vals[(i0 + 2 * i + dxb) * 32 + ch];
        }
#pragma unroll
        for (int i = 0; i < GS_ROWS / GS_WAVES / 2; ++i) {
            const float x = (float)((double)xx[i] * inv_scale);
            if (kk[i] != GS_EMPTY && x != 0.f) atomicAdd(gg.data + (int64_t)kk[i] * 32 + ch, x);
        }
    }
    STAMP(7)
#ifdef ENS_STAMPS
    {
        unsigned long long rt1_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_)::"memory");
        st_acc[ENS_NSEG - 1] = rt1_ - st_rt0;
        if (g_stamp_buf && lane == 0)
            for (int k_ = 0; k_ < ENS_NSEG; ++k_) g_stamp_buf[((size_t)blockIdx.x * GS_WAVES + wave) * ENS_NSEG + k_] = st_acc[k_];
    }
#endif
}

}  // namespace

int ens_launch_grid_scatter(int stage, int ntl, int n_rays, const float* ro, const float* rd, const double* z, const DevScene& sc,
                            const float* dgrid_ws, const float* d_raw, const DevGrid* grad_grids, hipStream_t st) {
    if (n_rays <= 0 || stage < 1 || stage > 3 || !dgrid_ws) return 0;
    ScatterArgs A;
    A.n_rays = n_rays; A.ntl = ntl; A.n_slots = stage;
    A.ro = ro; A.rd = rd; A.z = z; A.dgrid_ws = dgrid_ws; A.d_raw = d_raw;
    double ext = 0.0;
    for (int a = 0; a < 3; ++a) { A.lo[a] = sc.lo[a]; A.hi[a] = sc.hi[a]; ext = sc.hi[a] - sc.lo[a] > ext ? sc.hi[a] - sc.lo[a] : ext; }
    A.key_t = (float)(0.1 * ext);
    A.n_active_slots = 0;
    for (int s = 0; s < 3; ++s) {
        A.ggrid[s] = DevGrid{nullptr, 0, 0, 0};
        A.slot_of[s] = 0;
    }
    for (int s = 0; s < stage; ++s) {
        if (grad_grids[s + 1].data == nullptr) continue;
        A.ggrid[s] = grad_grids[s + 1];
        A.slot_of[A.n_active_slots++] = s;
    }
    if (A.n_active_slots == 0) return 0;
    static bool attr_done[ENS_MAX_DEVICES] = {};
    bool& attr_set = attr_done[ens_device_ordinal()];
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(grid_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                GS_BYTES) != hipSuccess)
            return -2;
        attr_set = true;
    }
    const int64_t n_groups = ((int64_t)n_rays + GS_G - 1) / GS_G;
    if (n_groups * A.n_active_slots > 0x7fffffff) return -1;
    grid_scatter_kernel<<<dim3((unsigned)(n_groups * A.n_active_slots)), dim3(GS_THREADS), GS_BYTES, st>>>(A);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

#ifdef ENS_STAMPS
extern "C" int enslam_debug_set_stamp_buffer3(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#endif
