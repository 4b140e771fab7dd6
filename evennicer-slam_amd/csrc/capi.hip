// extern "C" surface of libenslam_hip.so (declared in include/enslam_hip.h).
// Host-only logic: argument validation, decoder re-layout tables, struct marshalling.
#include "../../include/enslam_hip.h"
#include "kernels.hpp"

namespace {

bool is_xyz(int kind) { return kind == ENSLAM_MLP_MIDDLE || kind == ENSLAM_MLP_FINE || kind == ENSLAM_MLP_COLOR; }
int cdim(int kind) { return kind == ENSLAM_MLP_FINE ? 64 : 32; }
int nout(int kind) { return kind == ENSLAM_MLP_COLOR ? 4 : 1; }

thread_local int g_seg_dec = 0;     // decoder slot recorded in segments built next (multi-decoder jobs)

void add(PackJob& j, float* src, int off, int rows, int cols, int src_ld, int dst_ld, int tr) {
    if (src == nullptr || j.n >= ENS_MAX_SEGS) return;
    j.seg[j.n++] = PackSeg{src, off, (unsigned short)rows, (unsigned short)cols, (unsigned short)src_ld,
                           (unsigned short)dst_ld, (unsigned char)tr, (unsigned char)g_seg_dec};
}
void clear_job(PackJob& j) {
    j.n = 0;
    for (int i = 0; i < 4; ++i) { j.packed[i] = nullptr; j.part[i] = nullptr; j.part_stride[i] = 0; }
}

// Build the segment table of one decoder.  `with_transposed` adds the backward-only copies.
// Returns false when a required pointer is missing.
bool build_job(int kind, const enslam_mlp_params& P, bool with_transposed, PackJob& j, bool append = false) {
    if (!append) clear_job(j);
    const int n0 = j.n;
    if (is_xyz(kind)) {
        const XyzLay L{cdim(kind)};
        const int CD = cdim(kind), NO = nout(kind);
        const int kin[5] = {93, 32, 32, 125, 32};
        // Weight images the kernels read as MFMA fragments are tile-major (4: lds_util.hpp) so that a ds_read_b128 group
        // covers all 64 banks; the gradient accumulators (with_transposed == false: the unpack direction) stay row-major.
        const int fs = with_transposed ? 4 : 0;
        add(j, P.B, L.oBT(), 93, 3, 93, 4, 1);                                  // BT[f][k] = B[k][f]
        for (int i = 0; i < 5; ++i) {
            if (i == 3) {
                add(j, P.W[3], L.oW(3), 32, 93, 125, 128, fs);                    // embedding columns
                add(j, P.W[3] ? P.W[3] + 93 : nullptr, L.oW(3) + (fs ? 6 * 256 : 96), 32, 32, 125, 128, fs);   // h2 columns: column tiles 6, 7
            } else {
                add(j, P.W[i], L.oW(i), 32, kin[i], kin[i], L.K(i), fs);
            }
            add(j, P.b[i], L.ob(i), 1, 32, 32, 32, 0);
            add(j, P.Wc[i], L.oWc(i), 32, CD, CD, CD, fs);
            add(j, P.bc[i], L.obc(i), 1, 32, 32, 32, 0);
        }
        add(j, P.Wo, L.oWo(), NO, 32, 32, 32, 0);
        add(j, P.bo, L.obo(), 1, NO, NO, 16, 0);
        if (with_transposed) {
            for (int i = 0; i < 5; ++i) {
                if (i == 3) {
                    add(j, P.W[3], L.oWT(3), 93, 32, 125, 32, 3);              // 3: transposed + swizzled (lds_util.hpp)
                    add(j, P.W[3] ? P.W[3] + 93 : nullptr, L.oWT(3) + 96 * 32, 32, 32, 125, 32, 3);
                } else {
                    add(j, P.W[i], L.oWT(i), kin[i], 32, kin[i], 32, 3);
                }
                add(j, P.Wc[i], L.oWcT(i), 32, 32, CD, 32, 3);                   // grid channels only
            }
            add(j, P.Wo, L.oWoT(), 32, NO, 32, 4, 1);
            add(j, P.B, L.oBp(), 3, 93, 93, 96, 2);                              // swizzled, not transposed
        }
        const int need = with_transposed ? 24 + 13 : 24;
        return j.n - n0 == need;
    }
    if (kind == ENSLAM_MLP_COARSE) {
        const FeatLay L{};
        const int kin[5] = {32, 32, 32, 64, 32};
        for (int i = 0; i < 5; ++i) {
            add(j, P.W[i], L.oW(i), 32, kin[i], kin[i], L.K(i), 0);
            add(j, P.b[i], L.ob(i), 1, 32, 32, 32, 0);
        }
        add(j, P.Wo, L.oWo(), 1, 32, 32, 32, 0);
        add(j, P.bo, L.obo(), 1, 1, 1, 16, 0);
        if (with_transposed) {
            for (int i = 0; i < 5; ++i) add(j, P.W[i], L.oWT(i), kin[i], 32, kin[i], 32, 1);
            add(j, P.Wo, L.oWoT(), 32, 1, 32, 4, 1);
        }
        return j.n - n0 == (with_transposed ? 12 + 6 : 12);
    }
    return false;
}

bool to_dev_scene(const enslam_scene* s, DevScene& d) {
    if (s == nullptr) return false;
    for (int a = 0; a < 3; ++a) {
        d.lo[a] = s->bound[2 * a]; d.hi[a] = s->bound[2 * a + 1];
        d.clo[a] = s->coarse_bound[2 * a]; d.chi[a] = s->coarse_bound[2 * a + 1];
        d.gs[a] = (float)(2.0 / (d.hi[a] - d.lo[a]));      // IEEE double division, as the kernels' own (axis_coord)
    }
    for (int k = 0; k < 4; ++k) {
        d.grid[k] = DevGrid{s->grids[k].data, s->grids[k].D, s->grids[k].H, s->grids[k].W};
        d.packed[k] = s->packed[k];
    }
    return true;
}

// which grids / decoders a stage reads (NICE.forward, decoder.py:312-342)
bool stage_ok(int stage, const DevScene& d) {
    auto has = [&](int k) { return d.grid[k].data != nullptr && d.packed[k] != nullptr && d.grid[k].D > 0 &&
                                   d.grid[k].H > 0 && d.grid[k].W > 0; };
    switch (stage) {
        case ENSLAM_STAGE_COARSE: return has(0);
        case ENSLAM_STAGE_MIDDLE: return has(1);
        case ENSLAM_STAGE_FINE: return has(1) && has(2);
        case ENSLAM_STAGE_COLOR: return has(1) && has(2) && has(3);
        default: return false;
    }
}

}  // namespace

extern "C" {

// samples per ray the render kernels take: whole 16-sample tiles, at most 64 (one wave composites a ray).  64 = the padded
// second pass of hierarchical sampling (N_samples + N_importance + N_surface = 56 in the reference's iMAP-style configs),
// served by the tile-per-wave forward only (ENSLAM_EUNSUPPORTED from the launcher beyond its ray limit).
static bool samples_ok(int32_t n) { return n == 16 || n == 32 || n == 48 || n == 64; }

int enslam_abi_version(void) { return ENSLAM_ABI_VERSION; }
const char* enslam_arch(void) { return "gfx950"; }

size_t enslam_packed_floats(int kind) {
    if (is_xyz(kind)) return (size_t)XyzLay{cdim(kind)}.total();
    if (kind == ENSLAM_MLP_COARSE) return (size_t)FeatLay{}.total();
    return 0;
}
size_t enslam_packed_grad_floats(int kind) {
    if (is_xyz(kind)) return (size_t)XyzLay{cdim(kind)}.fwd_floats();
    if (kind == ENSLAM_MLP_COARSE) return (size_t)FeatLay{}.fwd_floats();
    return 0;
}

size_t enslam_bwd_partial_floats(int kind) {
    const size_t gf = enslam_packed_grad_floats(kind);
    return gf == 0 ? 0 : 16 + (size_t)ens_bwd_max_workgroups() * gf;
}

int enslam_pack_mlp(int kind, const enslam_mlp_params* params, float* packed, void* stream) {
    if (params == nullptr || packed == nullptr) return ENSLAM_EINVAL;
    PackJob job;
    if (!build_job(kind, *params, true, job)) return ENSLAM_EINVAL;
    return ens_launch_pack(job, packed, false, (hipStream_t)stream);
}

int enslam_pack_mlp_multi(int32_t n, const int32_t* kinds, const enslam_mlp_params* params, float* const* packed,
                          void* stream) {
    if (n < 0 || n > 3) return ENSLAM_EINVAL;
    if (n == 0) return ENSLAM_OK;
    if (!kinds || !params || !packed) return ENSLAM_EINVAL;
    PackJob job;
    clear_job(job);
    for (int i = 0; i < n; ++i) {
        if (!packed[i]) return ENSLAM_EINVAL;
        g_seg_dec = i;
        job.packed[i] = packed[i];
        const bool ok = build_job(kinds[i], params[i], true, job, true);
        g_seg_dec = 0;
        if (!ok) return ENSLAM_EINVAL;
    }
    return ens_launch_pack(job, nullptr, false, (hipStream_t)stream);
}

int enslam_unpack_mlp_grads(int kind, const float* packed_grad, const enslam_mlp_params* grads, void* stream) {
    if (grads == nullptr || packed_grad == nullptr) return ENSLAM_EINVAL;
    if (!is_xyz(kind) && kind != ENSLAM_MLP_COARSE) return ENSLAM_EINVAL;
    PackJob job;
    build_job(kind, *grads, false, job);        // NULL outputs are simply skipped
    return ens_launch_pack(job, const_cast<float*>(packed_grad), true, (hipStream_t)stream);
}

int enslam_unpack_mlp_grads_multi(int32_t n, const int32_t* kinds, const float* const* packed_grads,
                                  const enslam_mlp_params* grads, void* stream) {
    if (n < 0 || n > 4) return ENSLAM_EINVAL;
    if (n == 0) return ENSLAM_OK;
    if (!kinds || !packed_grads || !grads) return ENSLAM_EINVAL;
    PackJob job;
    clear_job(job);
    for (int i = 0; i < n; ++i) {
        if (!packed_grads[i] || (!is_xyz(kinds[i]) && kinds[i] != ENSLAM_MLP_COARSE)) return ENSLAM_EINVAL;
        g_seg_dec = i;
        job.packed[i] = const_cast<float*>(packed_grads[i]);
        build_job(kinds[i], grads[i], false, job, true);
    }
    g_seg_dec = 0;
    return ens_launch_pack(job, nullptr, true, (hipStream_t)stream);
}

namespace {
bool make_conv_job(int32_t n, const float* const* src, float* const* dst, const int64_t* n_voxels,
                   const uint8_t* const* need, uint8_t* const* valid, bool src_required, ConvJob& job) {
    if (n < 0 || n > 4 || !dst || !n_voxels || (src_required && !src)) return false;
    job.n = n;
    int begin = 0;
    for (int i = 0; i < 4; ++i) {
        job.block_begin[i] = begin;
        job.src[i] = nullptr; job.dst[i] = nullptr; job.V[i] = 0; job.need[i] = nullptr; job.valid[i] = nullptr;
        if (i < n) {
            if ((src_required && !src[i]) || !dst[i] || n_voxels[i] < 0) return false;
            job.src[i] = src ? src[i] : nullptr; job.dst[i] = dst[i]; job.V[i] = n_voxels[i];
            job.need[i] = need ? need[i] : nullptr;
            job.valid[i] = valid ? valid[i] : nullptr;
            begin += (int)((n_voxels[i] + 63) / 64);
        }
    }
    job.block_begin[4] = begin;
    for (int i = n; i < 4; ++i) job.block_begin[i] = begin;
    return true;
}
}  // namespace

namespace {
void empty_conv_job(ConvJob& job) {
    job.n = 0;
    for (int i = 0; i < 4; ++i) { job.src[i] = nullptr; job.dst[i] = nullptr; job.V[i] = 0; job.need[i] = nullptr; job.valid[i] = nullptr; }
    for (int i = 0; i < 5; ++i) job.block_begin[i] = 0;
}
}  // namespace

int enslam_step_prepare(int32_t n_dec, const int32_t* kinds, const enslam_mlp_params* params, float* const* packed,
                        int32_t n_conv, const float* const* src, float* const* dst, const int64_t* n_voxels,
                        const uint8_t* const* need, uint8_t* const* valid, int32_t n_zero, float* const* zero_dst,
                        const int64_t* zero_voxels, const uint8_t* const* zero_need, float* flat, int64_t n_flat,
                        void* stream) {
    if (n_dec < 0 || n_dec > 3 || n_conv < 0 || n_conv > 4 || n_zero < 0 || n_zero > 4 || n_flat < 0 || (n_flat > 0 && !flat))
        return ENSLAM_EINVAL;
    PackJob pj;
    clear_job(pj);
    if (n_dec > 0 && (!kinds || !params || !packed)) return ENSLAM_EINVAL;
    for (int i = 0; i < n_dec; ++i) {
        if (!packed[i]) return ENSLAM_EINVAL;
        g_seg_dec = i;
        pj.packed[i] = packed[i];
        const bool ok = build_job(kinds[i], params[i], true, pj, true);
        g_seg_dec = 0;
        if (!ok) return ENSLAM_EINVAL;
    }
    ConvJob cj, zj;
    empty_conv_job(cj);
    empty_conv_job(zj);
    if (n_conv > 0) {
        if (!need) return ENSLAM_EINVAL;
        for (int i = 0; i < n_conv; ++i)
            if (!need[i]) return ENSLAM_EINVAL;
        if (!make_conv_job(n_conv, src, dst, n_voxels, need, valid, true, cj)) return ENSLAM_EINVAL;
    }
    if (n_zero > 0 && !make_conv_job(n_zero, nullptr, zero_dst, zero_voxels, zero_need, nullptr, false, zj)) return ENSLAM_EINVAL;
    return ens_launch_step(pj, false, cj, true, zj, flat, n_flat, nullptr, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}

// optional work list arguments: both given or both NULL
static bool work_list_of(int32_t* tiles, int32_t* count, WorkList& w) {
    w.tiles = tiles; w.count = count;
    return (tiles == nullptr) == (count == nullptr);
}
static int step_finish_impl(int32_t n_conv, const float* const* src, float* const* dst, const int64_t* n_voxels,
                            const uint8_t* const* need, int32_t n_dec, const int32_t* kinds, const float* const* packed_grads,
                            const enslam_mlp_params* grads, const RayGradArgs* rg, void* stream, uint8_t* const* prev = nullptr,
                            const float* const* partials = nullptr, uint8_t* mv_need = nullptr, uint8_t* mv_prev = nullptr,
                            int64_t n_move = 0) {
    if (n_dec < 0 || n_dec > 4 || n_conv < 0 || n_conv > 4) return ENSLAM_EINVAL;
    PackJob pj;
    clear_job(pj);
    if (n_dec > 0 && (!kinds || !packed_grads || !grads)) return ENSLAM_EINVAL;
    for (int i = 0; i < n_dec; ++i) {
        if (!packed_grads[i] || (!is_xyz(kinds[i]) && kinds[i] != ENSLAM_MLP_COARSE)) return ENSLAM_EINVAL;
        g_seg_dec = i;
        pj.packed[i] = const_cast<float*>(packed_grads[i]);
        if (partials != nullptr && partials[i] != nullptr) {
            pj.part[i] = partials[i];
            pj.part_stride[i] = (int)enslam_packed_grad_floats(kinds[i]);
        }
        build_job(kinds[i], grads[i], false, pj, true);
    }
    g_seg_dec = 0;
    ConvJob cj, zj;
    empty_conv_job(cj);
    empty_conv_job(zj);
    if (n_conv > 0) {
        if (!need) return ENSLAM_EINVAL;
        for (int i = 0; i < n_conv; ++i)
            if (!need[i]) return ENSLAM_EINVAL;
        if (!make_conv_job(n_conv, src, dst, n_voxels, need, need ? prev : nullptr, true, cj)) return ENSLAM_EINVAL;
    }
    if (n_move < 0 || (n_move > 0 && (!mv_need || !mv_prev))) return ENSLAM_EINVAL;
    return ens_launch_step(pj, true, cj, false, zj, nullptr, 0, rg, (hipStream_t)stream, mv_need, mv_prev, n_move) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}
int enslam_step_finish(int32_t n_conv, const float* const* src, float* const* dst, const int64_t* n_voxels,
                       const uint8_t* const* need, int32_t n_dec, const int32_t* kinds, const float* const* packed_grads,
                       const enslam_mlp_params* grads, void* stream) {
    return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, nullptr, stream);
}
int enslam_step_finish_rays(int32_t n_conv, const float* const* src, float* const* dst, const int64_t* n_voxels,
                            const uint8_t* const* need, int32_t n_dec, const int32_t* kinds, const float* const* packed_grads,
                            const enslam_mlp_params* grads, int32_t stage, int32_t n_rays, int32_t n_samples,
                            const float* rays_o, const float* rays_d, const double* z_vals, const enslam_scene* scene,
                            float* dgrid_ws, float* g_rays_o, float* g_rays_d, const int32_t* work_list,
                            const int32_t* work_count, void* stream) {
    if (n_rays < 0) return ENSLAM_EINVAL;
    if (n_rays == 0 || stage == ENSLAM_STAGE_COARSE)
        return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, nullptr, stream);
    if (!samples_ok(n_samples)) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !dgrid_ws || !g_rays_o || !g_rays_d) return ENSLAM_EINVAL;
    RayGradArgs rg;
    WorkList wl;
    if (!work_list_of(const_cast<int32_t*>(work_list), const_cast<int32_t*>(work_count), wl)) return ENSLAM_EINVAL;
    if (!ens_ray_grad_args(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, d, dgrid_ws, g_rays_o, g_rays_d, rg,
                           work_list ? &wl : nullptr))
        return ENSLAM_EINVAL;
    return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, &rg, stream);
}
int enslam_step_finish_rays_prev(int32_t n_conv, const float* const* src, float* const* dst, const int64_t* n_voxels,
                                 const uint8_t* const* need, uint8_t* const* prev, int32_t n_dec, const int32_t* kinds,
                                 const float* const* packed_grads, const enslam_mlp_params* grads, int32_t stage, int32_t n_rays,
                                 int32_t n_samples, const float* rays_o, const float* rays_d, const double* z_vals,
                                 const enslam_scene* scene, float* dgrid_ws, float* g_rays_o, float* g_rays_d,
                                 const int32_t* work_list, const int32_t* work_count, void* stream) {
    if (n_rays < 0) return ENSLAM_EINVAL;
    if (prev != nullptr) {
        if (!need || n_conv < 0 || n_conv > 4) return ENSLAM_EINVAL;
        for (int i = 0; i < n_conv; ++i)
            if (!need[i] || !prev[i]) return ENSLAM_EINVAL;
    }
    if (n_rays == 0 || stage == ENSLAM_STAGE_COARSE)
        return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, nullptr, stream, prev);
    if (!samples_ok(n_samples)) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !dgrid_ws || !g_rays_o || !g_rays_d) return ENSLAM_EINVAL;
    RayGradArgs rg;
    WorkList wl;
    if (!work_list_of(const_cast<int32_t*>(work_list), const_cast<int32_t*>(work_count), wl)) return ENSLAM_EINVAL;
    if (!ens_ray_grad_args(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, d, dgrid_ws, g_rays_o, g_rays_d, rg,
                           work_list ? &wl : nullptr))
        return ENSLAM_EINVAL;
    return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, &rg, stream, prev);
}

int enslam_step_finish_partials(int32_t n_conv, const float* const* src, float* const* dst, const int64_t* n_voxels,
                                const uint8_t* const* need, uint8_t* const* prev, int32_t n_dec, const int32_t* kinds,
                                const float* const* packed_grads, const float* const* grad_partials, const enslam_mlp_params* grads,
                                int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                                const double* z_vals, const enslam_scene* scene, float* dgrid_ws, float* g_rays_o, float* g_rays_d,
                                const int32_t* work_list, const int32_t* work_count, void* stream) {
    if (n_rays < 0) return ENSLAM_EINVAL;
    if (prev != nullptr) {
        if (!need || n_conv < 0 || n_conv > 4) return ENSLAM_EINVAL;
        for (int i = 0; i < n_conv; ++i)
            if (!need[i] || !prev[i]) return ENSLAM_EINVAL;
    }
    if (n_rays == 0 || stage == ENSLAM_STAGE_COARSE || dgrid_ws == nullptr)
        return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, nullptr, stream, prev, grad_partials);
    if (!samples_ok(n_samples)) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !g_rays_o || !g_rays_d) return ENSLAM_EINVAL;
    RayGradArgs rg;
    WorkList wl;
    if (!work_list_of(const_cast<int32_t*>(work_list), const_cast<int32_t*>(work_count), wl)) return ENSLAM_EINVAL;
    if (!ens_ray_grad_args(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, d, dgrid_ws, g_rays_o, g_rays_d, rg,
                           work_list ? &wl : nullptr))
        return ENSLAM_EINVAL;
    return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, &rg, stream, prev, grad_partials);
}

int enslam_step_finish_native(int32_t n_conv, const float* const* src, float* const* dst, const int64_t* n_voxels,
                              const uint8_t* const* need, uint8_t* const* prev, int32_t n_dec, const int32_t* kinds,
                              const float* const* packed_grads, const float* const* grad_partials, const enslam_mlp_params* grads,
                              int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                              const double* z_vals, const enslam_scene* scene, float* dgrid_ws, float* g_rays_o, float* g_rays_d,
                              const int32_t* work_list, const int32_t* work_count, uint8_t* move_need, uint8_t* move_prev,
                              int64_t n_move, void* stream) {
    if (n_rays < 0 || n_move < 0 || (n_move > 0 && (!move_need || !move_prev))) return ENSLAM_EINVAL;
    if (prev != nullptr) {
        if (!need || n_conv < 0 || n_conv > 4) return ENSLAM_EINVAL;
        for (int i = 0; i < n_conv; ++i)
            if (!need[i] || !prev[i]) return ENSLAM_EINVAL;
    }
    if (n_rays == 0 || stage == ENSLAM_STAGE_COARSE || dgrid_ws == nullptr)
        return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, nullptr, stream, prev, grad_partials,
                                move_need, move_prev, n_move);
    if (!samples_ok(n_samples)) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !g_rays_o || !g_rays_d) return ENSLAM_EINVAL;
    RayGradArgs rg;
    WorkList wl;
    if (!work_list_of(const_cast<int32_t*>(work_list), const_cast<int32_t*>(work_count), wl)) return ENSLAM_EINVAL;
    if (!ens_ray_grad_args(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, d, dgrid_ws, g_rays_o, g_rays_d, rg,
                           work_list ? &wl : nullptr))
        return ENSLAM_EINVAL;
    return step_finish_impl(n_conv, src, dst, n_voxels, need, n_dec, kinds, packed_grads, grads, &rg, stream, prev, grad_partials,
                            move_need, move_prev, n_move);
}

int enslam_grids_convert(int32_t n, const float* const* src, float* const* dst, const int64_t* n_voxels,
                         int32_t to_voxel_major, void* stream) {
    if (n == 0) return ENSLAM_OK;
    ConvJob job;
    if (!make_conv_job(n, src, dst, n_voxels, nullptr, nullptr, true, job)) return ENSLAM_EINVAL;
    return ens_launch_convert(job, to_voxel_major != 0, (hipStream_t)stream);
}

int enslam_grids_convert_sparse(int32_t n, const float* const* src, float* const* dst, const int64_t* n_voxels,
                                const uint8_t* const* need, uint8_t* const* valid, int32_t to_voxel_major,
                                void* stream) {
    if (n == 0) return ENSLAM_OK;
    if (!need || n < 0 || n > 4) return ENSLAM_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!need[i]) return ENSLAM_EINVAL;
    ConvJob job;
    if (!make_conv_job(n, src, dst, n_voxels, need, valid, true, job)) return ENSLAM_EINVAL;
    return ens_launch_convert(job, to_voxel_major != 0, (hipStream_t)stream);
}

int enslam_zero_blocks(int32_t n, float* const* dst, const int64_t* n_voxels, const uint8_t* const* need,
                       float* flat, int64_t n_flat, void* stream) {
    if (n_flat < 0 || (n_flat > 0 && !flat)) return ENSLAM_EINVAL;
    if (n == 0 && n_flat == 0) return ENSLAM_OK;
    ConvJob job;
    if (n == 0) {
        job.n = 0;
        for (int i = 0; i < 5; ++i) job.block_begin[i] = 0;
    } else if (!make_conv_job(n, nullptr, dst, n_voxels, need, nullptr, false, job)) {
        return ENSLAM_EINVAL;
    }
    return ens_launch_zero_blocks(job, flat, n_flat, (hipStream_t)stream);
}

int enslam_tracker_loss_fwd(int32_t n, const double* depth, const double* uncertainty, const float* color,
                            const float* gt_depth, const float* gt_color, float w_color, double* loss, void* stream) {
    if (n < 0 || !loss || (n > 0 && (!depth || !uncertainty || !gt_depth)) || ((color == nullptr) != (gt_color == nullptr)))
        return ENSLAM_EINVAL;
    return ens_launch_tracker_loss(n, depth, uncertainty, color, gt_depth, gt_color, w_color, nullptr, loss, nullptr, nullptr,
                                   (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}
int enslam_tracker_loss_bwd(int32_t n, const double* depth, const double* uncertainty, const float* color,
                            const float* gt_depth, const float* gt_color, float w_color, const double* g_loss,
                            double* g_depth, float* g_color, void* stream) {
    if (n < 0 || !g_loss || (n > 0 && (!depth || !uncertainty || !gt_depth || !g_depth)) ||
        ((color == nullptr) != (gt_color == nullptr)))
        return ENSLAM_EINVAL;
    if (n == 0) return ENSLAM_OK;
    return ens_launch_tracker_loss(n, depth, uncertainty, color, gt_depth, gt_color, w_color, g_loss, nullptr, g_depth,
                                   g_color, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}
int enslam_gather_pixels(int32_t n, const int64_t* pixel_index, int32_t h0, int32_t w0, int32_t window_w, int32_t image_w,
                         int32_t image_h, const float* depth, const void* color, int32_t color_is_f64, float* pix_i, float* pix_j,
                         float* depth_out, void* color_out, void* stream) {
    if (n < 0 || h0 < 0 || w0 < 0 || window_w <= 0 || w0 + window_w > image_w || image_h <= h0) return ENSLAM_EINVAL;
    if (n == 0) return ENSLAM_OK;
    if (!pixel_index || !depth || !color || !pix_i || !pix_j || !depth_out || !color_out) return ENSLAM_EINVAL;
    return ens_launch_gather_pixels(n, pixel_index, h0, w0, window_w, image_w, depth, color, color_is_f64 != 0, pix_i, pix_j,
                                    depth_out, color_out, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}
int enslam_pose_rays_fwd(int32_t n, const float* camera_tensor, const float* pix_i, const float* pix_j, float fx,
                         float fy, float cx, float cy, float* rays_o, float* rays_d, void* stream) {
    if (n < 0 || !camera_tensor || (n > 0 && (!pix_i || !pix_j || !rays_o || !rays_d))) return ENSLAM_EINVAL;
    if (n == 0) return ENSLAM_OK;
    return ens_launch_pose_rays(n, camera_tensor, pix_i, pix_j, fx, fy, cx, cy, nullptr, nullptr, rays_o, rays_d, nullptr,
                                (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}
int enslam_pose_rays_bwd(int32_t n, const float* camera_tensor, const float* pix_i, const float* pix_j, float fx,
                         float fy, float cx, float cy, const float* g_rays_o, const float* g_rays_d,
                         float* g_camera_tensor, void* stream) {
    if (n < 0 || !camera_tensor || !g_camera_tensor || (n > 0 && (!pix_i || !pix_j))) return ENSLAM_EINVAL;
    return ens_launch_pose_rays(n, camera_tensor, pix_i, pix_j, fx, fy, cx, cy, g_rays_o, g_rays_d, nullptr, nullptr,
                                g_camera_tensor, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}

int enslam_adam_masked(int32_t n, float* const* param, float* const* grad, float* const* exp_avg,
                       float* const* exp_avg_sq, const uint8_t* const* mask, const int64_t* n_voxels,
                       const double* const* lr, const int32_t* const* step, double beta1, double beta2, double eps,
                       void* stream) {
    if (n == 0) return ENSLAM_OK;
    if (n < 0 || n > 4 || !param || !grad || !exp_avg || !exp_avg_sq || !n_voxels || !lr || !step) return ENSLAM_EINVAL;
    AdamJob job;
    job.n = n;
    job.beta1 = beta1; job.beta2 = beta2; job.eps = eps;
    int64_t blocks = 0;
    for (int i = 0; i < 4; ++i) {
        job.block_begin[i] = (int)blocks;
        if (i >= n) { job.p[i] = job.g[i] = job.m[i] = job.v[i] = nullptr; job.mask[i] = nullptr; job.V[i] = 0; job.lr[i] = nullptr; job.step[i] = nullptr; continue; }
        if (!param[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i] || !lr[i] || !step[i] || n_voxels[i] < 0) return ENSLAM_EINVAL;
        job.p[i] = param[i]; job.g[i] = grad[i]; job.m[i] = exp_avg[i]; job.v[i] = exp_avg_sq[i];
        job.mask[i] = mask ? mask[i] : nullptr;
        job.V[i] = n_voxels[i]; job.lr[i] = lr[i]; job.step[i] = step[i];
        blocks += (n_voxels[i] + 63) / 64;
        if (blocks > 0x7fffffff) return ENSLAM_EUNSUPPORTED;
    }
    job.block_begin[4] = (int)blocks;
    for (int i = n; i < 4; ++i) job.block_begin[i] = (int)blocks;
    return ens_launch_adam(job, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}

static int adam_tensors_impl(int32_t n, float* const* param, const float* const* grad, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numel, const double* lr, const int32_t* step,
                             double beta1, double beta2, double eps, void* stream, int self_inc);
int enslam_adam_tensors(int32_t n, float* const* param, const float* const* grad, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, const double* lr, const int32_t* step,
                        double beta1, double beta2, double eps, void* stream) {
    return adam_tensors_impl(n, param, grad, exp_avg, exp_avg_sq, numel, lr, step, beta1, beta2, eps, stream, 0);
}
int enslam_adam_tensors_step(int32_t n, float* const* param, const float* const* grad, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numel, const double* lr, int32_t* step,
                             double beta1, double beta2, double eps, void* stream) {
    return adam_tensors_impl(n, param, grad, exp_avg, exp_avg_sq, numel, lr, step, beta1, beta2, eps, stream, 1);
}
static int adam_tensors_impl(int32_t n, float* const* param, const float* const* grad, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numel, const double* lr, const int32_t* step,
                             double beta1, double beta2, double eps, void* stream, int self_inc) {
    if (n == 0) return ENSLAM_OK;
    if (n < 0 || n > ENS_ADAM_MAX_TENSORS) return ENSLAM_EUNSUPPORTED;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !numel || !lr || !step) return ENSLAM_EINVAL;
    AdamTensorsJob job;
    job.n = n; job.beta1 = beta1; job.beta2 = beta2; job.eps = eps; job.lr = lr; job.step = step; job.self_inc = self_inc;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!param[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0 || numel[i] > 0x7fffffff) return ENSLAM_EINVAL;
        job.p[i] = param[i]; job.g[i] = grad[i]; job.m[i] = exp_avg[i]; job.v[i] = exp_avg_sq[i];
        job.numel[i] = (int)numel[i];
        job.block_begin[i] = (int)blocks;
        blocks += (numel[i] + 1023) / 1024;
        if (blocks > 0x7fffffff) return ENSLAM_EUNSUPPORTED;
    }
    job.block_begin[n] = (int)blocks;
    if (self_inc && blocks != 1) return ENSLAM_EUNSUPPORTED;   // the count is read and replaced inside ONE workgroup
    return ens_launch_adam_tensors(job, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}

static int bucket_job(int32_t n_grids, float* const* grid_grad, int32_t channels, const int64_t* n_voxels,
                      const int32_t* layout, const uint8_t* flags, const int32_t* pos, int32_t n_small,
                      float* const* small, const int64_t* small_numel, int64_t small_base, float* bucket, BucketJob& job,
                      int block_voxels = 64) {
    if (n_grids < 0 || n_grids > 4 || n_small < 0 || n_small > ENS_ADAM_MAX_TENSORS) return ENSLAM_EUNSUPPORTED;
    if (block_voxels != 64 && block_voxels != 32 && block_voxels != 16 && block_voxels != 8) return ENSLAM_EINVAL;
    job.bs = block_voxels;
    if (!bucket || channels < 1 || small_base < 0) return ENSLAM_EINVAL;
    if (n_grids > 0 && (!grid_grad || !n_voxels || !layout || !flags || !pos)) return ENSLAM_EINVAL;
    if (n_small > 0 && (!small || !small_numel)) return ENSLAM_EINVAL;
    job.n_grids = n_grids; job.C = channels; job.flags = flags; job.pos = pos; job.bucket = bucket; job.n_small = n_small;
    int64_t blocks = 0;
    for (int g = 0; g < 4; ++g) { job.grid[g] = nullptr; job.V[g] = 0; job.layout[g] = 0; }
    for (int g = 0; g < n_grids; ++g) {
        if (!grid_grad[g] || n_voxels[g] < 0 || (layout[g] != 0 && layout[g] != 1)) return ENSLAM_EINVAL;
        job.grid[g] = grid_grad[g]; job.V[g] = n_voxels[g]; job.layout[g] = layout[g];
        job.blk_begin[g] = (int)blocks;
        blocks += (n_voxels[g] + block_voxels - 1) / block_voxels;
        if (blocks > 0x3fffffff) return ENSLAM_EUNSUPPORTED;
    }
    for (int g = n_grids; g <= 4; ++g) job.blk_begin[g] = (int)blocks;
    int64_t sblocks = 0, off = small_base;
    for (int i = 0; i < n_small; ++i) {
        if (!small[i] || small_numel[i] < 0 || small_numel[i] > 0x7fffffff) return ENSLAM_EINVAL;
        job.small[i] = small[i]; job.numel[i] = (int)small_numel[i];
        job.small_off[i] = off; off += small_numel[i];
        job.small_blk_begin[i] = (int)sblocks;
        sblocks += (small_numel[i] + 1023) / 1024;
        if (sblocks > 0x3fffffff) return ENSLAM_EUNSUPPORTED;
    }
    job.small_blk_begin[n_small] = (int)sblocks;
    return ENSLAM_OK;
}
int enslam_bucket_pack(int32_t n_grids, const float* const* grid_grad, int32_t channels, const int64_t* n_voxels,
                       const int32_t* layout, const uint8_t* flags, const int32_t* pos, int32_t n_small,
                       const float* const* small, const int64_t* small_numel, int64_t small_base, float* bucket,
                       void* stream) {
    BucketJob job;
    const int rc = bucket_job(n_grids, const_cast<float* const*>(grid_grad), channels, n_voxels, layout, flags, pos, n_small,
                              const_cast<float* const*>(small), small_numel, small_base, bucket, job);
    if (rc != ENSLAM_OK) return rc;
    return ens_launch_bucket(job, false, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}
int enslam_bucket_unpack(int32_t n_grids, float* const* grid_grad, int32_t channels, const int64_t* n_voxels,
                         const int32_t* layout, const uint8_t* flags, const int32_t* pos, int32_t n_small,
                         float* const* small, const int64_t* small_numel, int64_t small_base, const float* bucket,
                         void* stream) {
    BucketJob job;
    const int rc = bucket_job(n_grids, grid_grad, channels, n_voxels, layout, flags, pos, n_small, small, small_numel,
                              small_base, const_cast<float*>(bucket), job);
    if (rc != ENSLAM_OK) return rc;
    return ens_launch_bucket(job, true, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}

int enslam_mark_blocks(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                       const double* z_vals, const enslam_scene* scene, uint8_t* const* flags, void* stream) {
    return enslam_mark_blocks_g(stage, n_rays, n_samples, rays_o, rays_d, z_vals, scene, flags, 64, stream);
}
int enslam_mark_blocks_g(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                         const double* z_vals, const enslam_scene* scene, uint8_t* const* flags, int32_t block_voxels,
                         void* stream) {
    if (n_rays < 0 || n_samples < 1 || stage < 0 || stage > 3) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    DevScene d;
    if (!to_dev_scene(scene, d) || !rays_o || !rays_d || !z_vals || !flags) return ENSLAM_EINVAL;
    const int rc = ens_launch_mark_blocks(stage, n_rays, n_samples, rays_o, rays_d, z_vals, d, flags, (hipStream_t)stream, block_voxels);
    return rc == 0 ? ENSLAM_OK : (rc == -1 ? ENSLAM_EINVAL : ENSLAM_ELAUNCH);
}
int enslam_bucket_pack_g(int32_t n_grids, const float* const* grid_grad, int32_t channels, const int64_t* n_voxels,
                         const int32_t* layout, const uint8_t* flags, const int32_t* pos, int32_t n_small,
                         const float* const* small, const int64_t* small_numel, int64_t small_base, float* bucket,
                         int32_t block_voxels, void* stream) {
    BucketJob job;
    const int rc = bucket_job(n_grids, const_cast<float* const*>(grid_grad), channels, n_voxels, layout, flags, pos, n_small,
                              const_cast<float* const*>(small), small_numel, small_base, bucket, job, block_voxels);
    if (rc != ENSLAM_OK) return rc;
    return ens_launch_bucket(job, false, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}
int enslam_bucket_unpack_g(int32_t n_grids, float* const* grid_grad, int32_t channels, const int64_t* n_voxels,
                           const int32_t* layout, const uint8_t* flags, const int32_t* pos, int32_t n_small,
                           float* const* small, const int64_t* small_numel, int64_t small_base, const float* bucket,
                           int32_t block_voxels, void* stream) {
    BucketJob job;
    const int rc = bucket_job(n_grids, grid_grad, channels, n_voxels, layout, flags, pos, n_small, small, small_numel,
                              small_base, const_cast<float*>(bucket), job, block_voxels);
    if (rc != ENSLAM_OK) return rc;
    return ens_launch_bucket(job, true, (hipStream_t)stream) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}

int enslam_grid_to_voxel_major(const float* src, float* dst, int64_t n_voxels, void* stream) {
    if (src == nullptr || dst == nullptr || n_voxels < 0) return ENSLAM_EINVAL;
    return ens_launch_transpose(src, dst, n_voxels, true, (hipStream_t)stream);
}
int enslam_grid_from_voxel_major(const float* src, float* dst, int64_t n_voxels, void* stream) {
    if (src == nullptr || dst == nullptr || n_voxels < 0) return ENSLAM_EINVAL;
    return ens_launch_transpose(src, dst, n_voxels, false, (hipStream_t)stream);
}

int enslam_sample_rays(int32_t n_rays, int32_t n_lin, int32_t n_surf, const float* rays_o, const float* rays_d,
                       const float* gt_depth, const double* bound_host, const float* t_lin, const double* t_surf,
                       int32_t lindisp, const float* t_rand, float* scratch, int32_t depth_max_given, double* z_vals,
                       int32_t mark_stage, const enslam_scene* mark_scene, uint8_t* const* mark_flags, void* stream) {
    return enslam_sample_rays_g(n_rays, n_lin, n_surf, rays_o, rays_d, gt_depth, bound_host, t_lin, t_surf, lindisp, t_rand, scratch,
                                depth_max_given, z_vals, mark_stage, mark_scene, mark_flags, 64, nullptr, stream);
}
int enslam_sample_rays_g(int32_t n_rays, int32_t n_lin, int32_t n_surf, const float* rays_o, const float* rays_d,
                         const float* gt_depth, const double* bound_host, const float* t_lin, const double* t_surf,
                         int32_t lindisp, const float* t_rand, float* scratch, int32_t depth_max_given, double* z_vals,
                         int32_t mark_stage, const enslam_scene* mark_scene, uint8_t* const* mark_flags,
                         int32_t mark_block_voxels, uint8_t* const* mark_flags64, void* stream) {
    if (n_rays < 0 || n_lin < 1 || n_surf < 0) return ENSLAM_EINVAL;
    int shift = 6;
    switch (mark_block_voxels) { case 64: shift = 6; break; case 32: shift = 5; break; case 16: shift = 4; break; case 8: shift = 3; break; default: return ENSLAM_EINVAL; }
    if (n_rays == 0) return ENSLAM_OK;
    if (!rays_o || !rays_d || !bound_host || !t_lin || !z_vals) return ENSLAM_EINVAL;
    if (gt_depth != nullptr && (scratch == nullptr || (n_surf > 0 && t_surf == nullptr))) return ENSLAM_EINVAL;
    MarkArgs mk;
    const bool marking = mark_scene != nullptr && mark_flags != nullptr;
    if (marking) {
        if (mark_stage < 0 || mark_stage > 3 || !to_dev_scene(mark_scene, mk.sc)) return ENSLAM_EINVAL;
        mk.kmask = mark_stage == 0 ? 1 : (mark_stage == 1 ? 2 : (mark_stage == 2 ? 6 : 14));
        for (int k = 0; k < 4; ++k) { mk.flags[k] = mark_flags[k]; mk.flags64[k] = (mark_flags64 != nullptr && shift < 6) ? mark_flags64[k] : nullptr; }
        mk.shift = shift;
    }
    return ens_launch_sample(n_rays, n_lin, n_surf, rays_o, rays_d, gt_depth, bound_host, t_lin, t_surf, lindisp,
                             t_rand, scratch, depth_max_given, z_vals, marking ? &mk : nullptr, (hipStream_t)stream);
}

int enslam_sample_prepare(int32_t n_rays, int32_t n_lin, int32_t n_surf, const float* rays_o, const float* rays_d,
                          const float* gt_depth, const double* bound_host, const float* t_lin, const double* t_surf,
                          int32_t lindisp, const float* t_rand, float* scratch, int32_t depth_max_given, double* z_vals,
                          int32_t mark_stage, const enslam_scene* mark_scene, uint8_t* const* mark_flags,
                          int32_t mark_block_voxels, uint8_t* const* mark_flags64,
                          int32_t n_dec, const int32_t* kinds, const enslam_mlp_params* params, float* const* packed,
                          int32_t n_zero, float* const* zero_dst, const int64_t* zero_voxels, const uint8_t* const* zero_need,
                          float* flat, int64_t n_flat, void* stream) {
    if (n_rays < 0 || n_lin < 1 || n_surf < 0) return ENSLAM_EINVAL;
    if (n_dec < 0 || n_dec > 3 || n_zero < 0 || n_zero > 4 || n_flat < 0 || (n_flat > 0 && !flat)) return ENSLAM_EINVAL;
    int shift = 6;
    switch (mark_block_voxels) { case 64: shift = 6; break; case 32: shift = 5; break; case 16: shift = 4; break; case 8: shift = 3; break; default: return ENSLAM_EINVAL; }
    if (n_rays > 0) {
        if (!rays_o || !rays_d || !bound_host || !t_lin || !z_vals) return ENSLAM_EINVAL;
        if (gt_depth != nullptr && (scratch == nullptr || (n_surf > 0 && t_surf == nullptr))) return ENSLAM_EINVAL;
    } else if (!bound_host) {
        return ENSLAM_EINVAL;
    }
    MarkArgs mk;
    const bool marking = mark_scene != nullptr && mark_flags != nullptr;
    if (marking) {
        if (mark_stage < 0 || mark_stage > 3 || !to_dev_scene(mark_scene, mk.sc)) return ENSLAM_EINVAL;
        mk.kmask = mark_stage == 0 ? 1 : (mark_stage == 1 ? 2 : (mark_stage == 2 ? 6 : 14));
        for (int k = 0; k < 4; ++k) { mk.flags[k] = mark_flags[k]; mk.flags64[k] = (mark_flags64 != nullptr && shift < 6) ? mark_flags64[k] : nullptr; }
        mk.shift = shift;
    }
    PackJob pj;
    clear_job(pj);
    if (n_dec > 0 && (!kinds || !params || !packed)) return ENSLAM_EINVAL;
    for (int i = 0; i < n_dec; ++i) {
        if (!packed[i]) return ENSLAM_EINVAL;
        g_seg_dec = i;
        pj.packed[i] = packed[i];
        const bool ok = build_job(kinds[i], params[i], true, pj, true);
        g_seg_dec = 0;
        if (!ok) return ENSLAM_EINVAL;
    }
    ConvJob zj;
    empty_conv_job(zj);
    if (n_zero > 0 && !make_conv_job(n_zero, nullptr, zero_dst, zero_voxels, zero_need, nullptr, false, zj)) return ENSLAM_EINVAL;
    const int rc = ens_launch_sample_prepare(n_rays, n_lin, n_surf, rays_o, rays_d, gt_depth, bound_host, t_lin, t_surf, lindisp, t_rand,
                                             scratch, depth_max_given, z_vals, marking ? &mk : nullptr, pj, zj, flat, n_flat,
                                             (hipStream_t)stream);
    return rc == 0 ? ENSLAM_OK : (rc == -1 ? ENSLAM_EINVAL : ENSLAM_ELAUNCH);
}

int enslam_render_fwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                      const double* z_vals, const enslam_scene* scene, double* depth, double* var, float* rgb,
                      float* raw_out, float* act_ws, int32_t act_light, void* stream) {
    if (n_rays < 0) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    if (!samples_ok(n_samples)) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !depth || !var || !rgb) return ENSLAM_EINVAL;
    if (act_ws != nullptr)                               // the saved cell records hold the voxel index in 29 bits
        for (int k = 1; k < 4; ++k)
            if (d.grid[k].data && (int64_t)d.grid[k].D * d.grid[k].H * d.grid[k].W >= ACT_MAX_VOXELS) return ENSLAM_EUNSUPPORTED;
    return ens_launch_render_fwd(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, nullptr, 0, 1, d, depth, var,
                                 rgb, raw_out, stage == ENSLAM_STAGE_COARSE ? nullptr : act_ws, act_light != 0, (hipStream_t)stream);
}

int enslam_render_loss_fwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                           const double* z_vals, const enslam_scene* scene, double* depth, double* var, float* rgb,
                           float* raw_out, float* act_ws, int32_t act_light, const float* gt_depth, const float* gt_color,
                           float w_color, double* loss, float* d_raw_unit, int32_t* work_list, int32_t* work_count,
                           void* stream) {
    if (n_rays < 0) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    if (!samples_ok(n_samples)) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !depth || !var || !rgb || !raw_out || !gt_depth || !loss) return ENSLAM_EINVAL;
    if (act_ws != nullptr)
        for (int k = 1; k < 4; ++k)
            if (d.grid[k].data && (int64_t)d.grid[k].D * d.grid[k].H * d.grid[k].W >= ACT_MAX_VOXELS) return ENSLAM_EUNSUPPORTED;
    const LossSpec ls{gt_depth, gt_color, w_color, loss, nullptr, d_raw_unit};
    WorkList wl;
    if (!work_list_of(work_list, work_count, wl) || (work_list && !d_raw_unit)) return ENSLAM_EINVAL;
    const int rc = ens_launch_render_fwd(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, nullptr, 0, 1, d, depth, var, rgb,
                                         raw_out, stage == ENSLAM_STAGE_COARSE ? nullptr : act_ws, act_light != 0,
                                         (hipStream_t)stream, &ls, work_list ? &wl : nullptr);
    return rc == 0 ? ENSLAM_OK : (rc == -1 ? ENSLAM_EUNSUPPORTED : ENSLAM_ELAUNCH);
}
int enslam_render_tracker_loss_fwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                                   const double* z_vals, const enslam_scene* scene, double* depth, double* var, float* rgb,
                                   float* raw_out, float* act_ws, int32_t act_light, const float* gt_depth, const float* gt_color,
                                   float w_color, const uint8_t* inside, int32_t handle_dynamic, double* tmp_scratch, double* loss,
                                   float* d_raw_unit, int32_t* work_list, int32_t* work_count, void* stream) {
    if (n_rays < 0) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    if (!samples_ok(n_samples) || n_rays > ENS_TRACKER_TAIL_MAX_RAYS || stage == ENSLAM_STAGE_COARSE) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !depth || !var || !rgb || !raw_out || !gt_depth || !loss || !tmp_scratch) return ENSLAM_EINVAL;
    if (act_ws != nullptr)
        for (int k = 1; k < 4; ++k)
            if (d.grid[k].data && (int64_t)d.grid[k].D * d.grid[k].H * d.grid[k].W >= ACT_MAX_VOXELS) return ENSLAM_EUNSUPPORTED;
    const LossSpec ls{gt_depth, gt_color, w_color, loss, nullptr, d_raw_unit};
    const TrackerSpec ts{inside, handle_dynamic != 0 ? 1 : 0, tmp_scratch};
    WorkList wl;
    if (!work_list_of(work_list, work_count, wl) || (work_list && !d_raw_unit)) return ENSLAM_EINVAL;
    const int rc = ens_launch_render_fwd(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, nullptr, 0, 1, d, depth, var, rgb,
                                         raw_out, act_ws, act_light != 0, (hipStream_t)stream, &ls, work_list ? &wl : nullptr, &ts);
    return rc == 0 ? ENSLAM_OK : (rc == -1 ? ENSLAM_EUNSUPPORTED : ENSLAM_ELAUNCH);
}
int enslam_tracker_tail_max_rays(void) { return ENS_TRACKER_TAIL_MAX_RAYS; }

int enslam_tracker_rays(int32_t n, const float* camera_tensor, const int64_t* pixel_index, int32_t H0, int32_t W0, int32_t window_w,
                        int32_t image_w, int32_t image_h, const float* depth_image, const void* color_image, int32_t color_is_f64, float fx, float fy,
                        float cx, float cy, const double* bound_host, int32_t prefilter, float* pix_i, float* pix_j, float* rays_o,
                        float* rays_d, float* gt_depth, float* gt_color, uint8_t* inside, float* depth_max, int32_t* draw_counter,
                        int32_t n_draws, void* stream) {
    if (n < 0 || window_w <= 0 || image_w <= 0 || image_h <= 0 || H0 < 0 || W0 < 0 || W0 + window_w > image_w || H0 >= image_h ||
        (draw_counter && n_draws < 1))
        return ENSLAM_EINVAL;
    if (n == 0) return ENSLAM_OK;
    if (!camera_tensor || !pixel_index || !depth_image || !color_image || !bound_host || !pix_i || !pix_j || !rays_o || !rays_d ||
        !gt_depth || !gt_color || !depth_max || (prefilter && !inside))
        return ENSLAM_EINVAL;
    return ens_launch_tracker_rays(n, camera_tensor, pixel_index, H0, W0, window_w, image_w, image_h, depth_image, color_image, color_is_f64, fx, fy,
                                   cx, cy, bound_host, pix_i, pix_j, rays_o, rays_d, gt_depth, gt_color, prefilter ? inside : nullptr,
                                   depth_max, (hipStream_t)stream, draw_counter, n_draws) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}

int enslam_composite_loss_bwd(int32_t n_rays, int32_t n_samples, const float* raw, const double* z_vals,
                              const double* depth, const float* rgb, const float* gt_depth, const float* gt_color,
                              float w_color, const double* g_loss, float* d_raw, int32_t* work_list, int32_t* work_count,
                              void* stream) {
    if (n_rays < 0 || n_samples < 1 || n_samples > 64) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    if (!raw || !z_vals || !depth || !gt_depth || !g_loss || !d_raw || (gt_color && !rgb)) return ENSLAM_EINVAL;
    const LossSpec ls{gt_depth, gt_color, w_color, nullptr, g_loss, nullptr};
    WorkList wl;
    if (!work_list_of(work_list, work_count, wl)) return ENSLAM_EINVAL;
    return ens_launch_composite_bwd(n_rays, n_samples, raw, z_vals, depth, nullptr, nullptr, nullptr, d_raw,
                                    (hipStream_t)stream, &ls, rgb, work_list ? &wl : nullptr) == 0 ? ENSLAM_OK : ENSLAM_ELAUNCH;
}

size_t enslam_grid_handoff_floats(int32_t stage, int32_t n_rays, int32_t n_samples) {
    if (stage == ENSLAM_STAGE_COARSE || n_rays <= 0 || n_samples <= 0) return 0;
    return (size_t)n_rays * (size_t)(n_samples / 16) * ACT_SLOTS * DG_STRIDE;
}

size_t enslam_activation_floats(int32_t stage, int32_t n_rays, int32_t n_samples, int32_t act_light) {
    if (stage == ENSLAM_STAGE_COARSE || n_rays <= 0 || n_samples <= 0) return 0;
    // full workspace: the activation blocks, then the dh region of the two-kernel backward (render_bwd2.hip)
    return (size_t)n_rays * (size_t)(n_samples / 16) * ACT_SLOTS * (act_light ? ACTL_STRIDE : ACT_STRIDE + DH_STRIDE);
}

int enslam_eval_points(int32_t stage, int64_t n_points, const double* points, const enslam_scene* scene,
                       int32_t apply_mask, float* raw_out, void* stream) {
    if (n_points < 0) return ENSLAM_EINVAL;
    if (n_points == 0) return ENSLAM_OK;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d) || !points || !raw_out) return ENSLAM_EINVAL;
    const int64_t units = (n_points + 47) / 48;
    return ens_launch_render_fwd(stage, 3, units, nullptr, nullptr, nullptr, points, n_points, apply_mask, d, nullptr,
                                 nullptr, nullptr, raw_out, nullptr, 0, (hipStream_t)stream);
}

int enslam_composite_fwd(int32_t n_rays, int32_t n_samples, const float* raw, const double* z_vals, double* depth,
                         double* var, float* rgb, float* weights, void* stream) {
    if (n_rays < 0 || n_samples < 1 || n_samples > 64) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    if (!raw || !z_vals || !depth || !var || !rgb) return ENSLAM_EINVAL;
    return ens_launch_composite_fwd(n_rays, n_samples, raw, z_vals, depth, var, rgb, weights, (hipStream_t)stream);
}

int enslam_composite_bwd(int32_t n_rays, int32_t n_samples, const float* raw, const double* z_vals,
                         const double* depth, const double* g_depth, const double* g_var, const float* g_rgb,
                         float* d_raw, void* stream) {
    if (n_rays < 0 || n_samples < 1 || n_samples > 64) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    if (!raw || !z_vals || !depth || !d_raw) return ENSLAM_EINVAL;
    return ens_launch_composite_bwd(n_rays, n_samples, raw, z_vals, depth, g_depth, g_var, g_rgb, d_raw,
                                    (hipStream_t)stream);
}

int enslam_composite_bwd_list(int32_t n_rays, int32_t n_samples, const float* raw, const double* z_vals,
                              const double* depth, const double* g_depth, const double* g_var, const float* g_rgb,
                              float* d_raw, int32_t* work_list, int32_t* work_count, void* stream) {
    if (n_rays < 0 || n_samples < 1 || n_samples > 64) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    WorkList wl;
    if (!raw || !z_vals || !depth || !d_raw || !work_list_of(work_list, work_count, wl)) return ENSLAM_EINVAL;
    return ens_launch_composite_bwd(n_rays, n_samples, raw, z_vals, depth, g_depth, g_var, g_rgb, d_raw, (hipStream_t)stream,
                                    nullptr, nullptr, work_list ? &wl : nullptr);
}

int enslam_decoder_bwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                       const double* z_vals, const enslam_scene* scene, const float* d_raw, const float* act_ws,
                       int32_t act_light, float* dgrid_ws, const enslam_grid* grad_grids, float* const* grad_packed,
                       float* g_rays_o, float* g_rays_d, void* stream) {
    return enslam_decoder_bwd_scaled(stage, n_rays, n_samples, rays_o, rays_d, z_vals, scene, d_raw, nullptr, act_ws, act_light,
                                     dgrid_ws, grad_grids, grad_packed, g_rays_o, g_rays_d, nullptr, nullptr, stream);
}
int enslam_decoder_bwd_scaled(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                              const double* z_vals, const enslam_scene* scene, const float* d_raw, const double* d_raw_scale,
                              const float* act_ws, int32_t act_light, float* dgrid_ws, const enslam_grid* grad_grids,
                              float* const* grad_packed, float* g_rays_o, float* g_rays_d, const int32_t* work_list,
                              const int32_t* work_count, void* stream) {
    return enslam_decoder_bwd_partials(stage, n_rays, n_samples, rays_o, rays_d, z_vals, scene, d_raw, d_raw_scale, act_ws, act_light,
                                       dgrid_ws, grad_grids, grad_packed, nullptr, g_rays_o, g_rays_d, work_list, work_count, stream);
}
int enslam_decoder_bwd_partials(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                                const double* z_vals, const enslam_scene* scene, const float* d_raw, const double* d_raw_scale,
                                const float* act_ws, int32_t act_light, float* dgrid_ws, const enslam_grid* grad_grids,
                                float* const* grad_packed, float* const* grad_partial, float* g_rays_o, float* g_rays_d,
                                const int32_t* work_list, const int32_t* work_count, void* stream) {
    if (n_rays < 0) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    if (!samples_ok(n_samples)) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !d_raw || !grad_grids || !grad_packed) return ENSLAM_EINVAL;
    if (act_ws != nullptr && act_light)                  // the light workspace cannot serve parameter gradients
        for (int k = 1; k < 4; ++k)
            if (grad_packed[k] != nullptr) return ENSLAM_EINVAL;
    DevGrid gg[4];
    for (int k = 0; k < 4; ++k) {
        gg[k] = DevGrid{grad_grids[k].data, d.grid[k].D, d.grid[k].H, d.grid[k].W};
        if (grad_grids[k].data != nullptr &&
            (grad_grids[k].D != d.grid[k].D || grad_grids[k].H != d.grid[k].H || grad_grids[k].W != d.grid[k].W))
            return ENSLAM_EINVAL;
    }
    WorkList wl;
    if (!work_list_of(const_cast<int32_t*>(work_list), const_cast<int32_t*>(work_count), wl)) return ENSLAM_EINVAL;
    return ens_launch_decoder_bwd(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, d, d_raw, act_ws, act_light != 0, dgrid_ws, gg,
                                  grad_packed, g_rays_o, g_rays_d, (hipStream_t)stream, d_raw_scale, work_list ? &wl : nullptr,
                                  grad_partial);
}

int enslam_ray_grad_bwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                        const double* z_vals, const enslam_scene* scene, float* dgrid_ws, float* g_rays_o,
                        float* g_rays_d, void* stream) {
    if (n_rays < 0) return ENSLAM_EINVAL;
    if (n_rays == 0 || stage == ENSLAM_STAGE_COARSE) return ENSLAM_OK;
    if (!samples_ok(n_samples)) return ENSLAM_EUNSUPPORTED;
    DevScene d;
    if (!to_dev_scene(scene, d) || !stage_ok(stage, d)) return ENSLAM_EINVAL;
    if (!rays_o || !rays_d || !z_vals || !dgrid_ws || !g_rays_o || !g_rays_d) return ENSLAM_EINVAL;
    const int rc = ens_launch_ray_grad_bwd(stage, n_samples / 16, n_rays, rays_o, rays_d, z_vals, d, dgrid_ws, g_rays_o,
                                           g_rays_d, (hipStream_t)stream);
    return rc == 0 ? ENSLAM_OK : (rc == -1 ? ENSLAM_EINVAL : ENSLAM_ELAUNCH);
}

int enslam_render_bwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                      const double* z_vals, const enslam_scene* scene, const float* raw, const double* depth,
                      const double* g_depth, const double* g_var, const float* g_rgb, const enslam_grid* grad_grids,
                      float* const* grad_packed, float* g_rays_o, float* g_rays_d, float* d_raw, const float* act_ws,
                      int32_t act_light, float* dgrid_ws, void* stream) {
    if (!g_depth && !g_var && !g_rgb) return ENSLAM_EINVAL;
    const int rc = enslam_composite_bwd(n_rays, n_samples, raw, z_vals, depth, g_depth, g_var, g_rgb, d_raw, stream);
    if (rc != ENSLAM_OK) return rc;
    const int rd = enslam_decoder_bwd(stage, n_rays, n_samples, rays_o, rays_d, z_vals, scene, d_raw, act_ws, act_light, dgrid_ws,
                                      grad_grids, grad_packed, g_rays_o, g_rays_d, stream);
    if (rd != ENSLAM_OK || !act_ws || !g_rays_o || !g_rays_d || stage == ENSLAM_STAGE_COARSE) return rd;
    return enslam_ray_grad_bwd(stage, n_rays, n_samples, rays_o, rays_d, z_vals, scene, dgrid_ws, g_rays_o, g_rays_d,
                               stream);
}

int enslam_rgbd_loss_fwd(int32_t n, const double* depth, const float* color, const float* gt_depth,
                         const float* gt_color, float w_color, double* loss, void* stream) {
    if (n < 0 || !loss || (n > 0 && (!depth || !gt_depth)) || ((color == nullptr) != (gt_color == nullptr)))
        return ENSLAM_EINVAL;
    return ens_launch_rgbd_loss(n, depth, color, gt_depth, gt_color, w_color, nullptr, loss, nullptr, nullptr,
                                (hipStream_t)stream);
}
int enslam_rgbd_loss_bwd(int32_t n, const double* depth, const float* color, const float* gt_depth,
                         const float* gt_color, float w_color, const double* g_loss, double* g_depth, float* g_color,
                         void* stream) {
    if (n < 0 || !g_loss || (n > 0 && (!depth || !gt_depth || !g_depth)) || ((color == nullptr) != (gt_color == nullptr)))
        return ENSLAM_EINVAL;
    if (n == 0) return ENSLAM_OK;
    return ens_launch_rgbd_loss(n, depth, color, gt_depth, gt_color, w_color, g_loss, nullptr, g_depth, g_color,
                                (hipStream_t)stream);
}

int enslam_voxel_index(int64_t n_points, const double* points, const double* bound_host, int32_t D, int32_t H,
                       int32_t W, int32_t* ix, int32_t* iy, int32_t* iz, float* fx, float* fy, float* fz,
                       void* stream) {
    if (n_points < 0 || D < 1 || H < 1 || W < 1) return ENSLAM_EINVAL;
    if (n_points == 0) return ENSLAM_OK;
    if (!points || !bound_host || !ix || !iy || !iz || !fx || !fy || !fz) return ENSLAM_EINVAL;
    return ens_launch_voxel_index(n_points, points, bound_host, D, H, W, ix, iy, iz, fx, fy, fz, (hipStream_t)stream);
}

int enslam_ray_points(int32_t n_rays, int32_t n_samples, const float* rays_o, const float* rays_d,
                      const double* z_vals, const double* bound_host, double* points, uint8_t* mask, void* stream) {
    if (n_rays < 0 || n_samples < 1) return ENSLAM_EINVAL;
    if (n_rays == 0) return ENSLAM_OK;
    if (!rays_o || !rays_d || !z_vals || !bound_host || !points || !mask) return ENSLAM_EINVAL;
    return ens_launch_ray_points(n_rays, n_samples, rays_o, rays_d, z_vals, bound_host, points, mask,
                                 (hipStream_t)stream);
}

}  // extern "C"

int enslam_fourier_sincos(int64_t n, const float* x, float* sin_out, float* cos_out, void* stream) {
    if (n < 0) return ENSLAM_EINVAL;
    if (n == 0) return ENSLAM_OK;
    if (!x || (!sin_out && !cos_out)) return ENSLAM_EINVAL;
    return ens_launch_sincos(n, x, sin_out, cos_out, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------------
// step plans (include/enslam_hip.h): the launches of one differentiable render call behind two entry points
// ------------------------------------------------------------------------------------------------------------------
namespace {
int64_t plan_align(int64_t b) { return (b + 255) & ~(int64_t)255; }
bool plan_ok(const enslam_step_plan* p) {
    if (!p || p->stage < ENSLAM_STAGE_MIDDLE || p->stage > ENSLAM_STAGE_COLOR || p->n_rays <= 0 || p->n_lin < 1 || p->n_surf < 0) return false;
    if (!samples_ok(p->n_lin + p->n_surf) || !p->t_lin || (p->n_surf > 0 && !p->t_surf)) return false;
    for (int k = 1; k < 4; ++k) {
        const bool used = k <= p->stage;
        if (used != (p->grid_mode[k] != 0) || p->grid_mode[k] < 0 || p->grid_mode[k] > 3) return false;
        if (used && (p->grid_D[k] < 1 || p->grid_H[k] < 1 || p->grid_W[k] < 1)) return false;
        if (!used && p->par_grad[k]) return false;
    }
    if (p->grid_mode[0] != 0 || p->par_grad[0]) return false;
    return p->loss_kind == 0 || p->loss_kind == 1;
}
int64_t plan_voxels(const enslam_step_plan* p, int k) { return (int64_t)p->grid_D[k] * p->grid_H[k] * p->grid_W[k]; }
template <typename T> T* at(void* base, int64_t off) { return off < 0 ? nullptr : reinterpret_cast<T*>(static_cast<char*>(base) + off); }
}  // namespace

int64_t enslam_plan_struct_bytes(int32_t which) { return which == 0 ? (int64_t)sizeof(enslam_step_plan) : (int64_t)sizeof(enslam_step_layout); }

int enslam_plan_layout(const enslam_step_plan* plan, enslam_step_layout* L) {
    if (!plan_ok(plan) || !L) return ENSLAM_EINVAL;
    const int N = plan->n_rays, S = plan->n_lin + plan->n_surf;
    const int64_t ntiles = (int64_t)N * (S / 16);
    bool any_par = false, any_m3 = false;
    for (int k = 1; k < 4; ++k) { any_par = any_par || plan->par_grad[k]; any_m3 = any_m3 || plan->grid_mode[k] == 3; }
    if ((plan->act_light != 0) == any_par) return ENSLAM_EINVAL;           // the light workspace iff no decoder parameter wants a gradient
    L->n_samples = S;
    L->finish_needed = (any_par || any_m3) ? 1 : 0;
    L->inline_rays = L->finish_needed ? 0 : 1;
    L->merged = any_m3 ? 0 : 1;
    // ---- scratch blob
    int64_t o = 0;
    for (int k = 0; k < 4; ++k) {
        L->s_flags[k] = -1;
        if (plan->grid_mode[k] >= 2) { L->s_flags[k] = o; o = plan_align(o + (plan_voxels(plan, k) + 63) / 64); }
    }
    for (int k = 0; k < 4; ++k) {
        L->s_packed[k] = -1;
        if (plan->grid_mode[k] != 0) { L->s_packed[k] = o; o = plan_align(o + 4 * (int64_t)enslam_packed_floats(k)); }
    }
    L->s_zero_bytes = o;
    L->s_z = o; o = plan_align(o + 8 * (int64_t)N * S);
    L->s_dmax = o; o = plan_align(o + 8);
    L->s_raw = o; o = plan_align(o + 16 * (int64_t)N * S);
    L->s_act = o; o = plan_align(o + 4 * (int64_t)enslam_activation_floats(plan->stage, N, S, plan->act_light));
    L->s_work = -1;
    if (plan->use_work_list) { L->s_work = o; o = plan_align(o + 4 * ntiles); }
    L->s_draw = o; o = plan_align(o + 16 * (int64_t)N * S);
    L->s_dgw = -1;
    bool any_ggrid = false;                   // the grid gradients' scatter reads the hand-off (grid_scatter.hip; policy: ens_launch_decoder_bwd)
    {
        static const int defer_mode = [] { const char* e = getenv("ENSLAM_DEFER_SCATTER"); return e == nullptr ? 0 : (e[0] == '1' ? 1 : (e[0] == '2' ? 2 : 0)); }();
        bool any_grid = false, light_grid = false;
        for (int k = 1; k < 4; ++k) {
            any_grid = any_grid || plan->grid_mode[k] >= 2;
            light_grid = light_grid || (plan->grid_mode[k] >= 2 && !plan->par_grad[k]);
        }
        any_ggrid = defer_mode == 1 ? any_grid : (defer_mode == 2 && light_grid);
    }
    if ((plan->need_rays && !L->inline_rays) || any_ggrid) { L->s_dgw = o; o = plan_align(o + 4 * (int64_t)enslam_grid_handoff_floats(plan->stage, N, S)); }
    for (int k = 0; k < 4; ++k) {
        L->s_vm[k] = L->s_gacc[k] = -1;
        if (plan->grid_mode[k] == 3) {
            L->s_vm[k] = o; o = plan_align(o + 128 * plan_voxels(plan, k));
            L->s_gacc[k] = o; o = plan_align(o + 128 * plan_voxels(plan, k));
        }
    }
    L->scratch_bytes = o;
    // ---- gradient blob: [flat range cleared by the forward | parameter gradients | dense channel-major grid gradients]
    int64_t f = 0;                                                         // floats
    for (int k = 0; k < 4; ++k) {
        L->g_packed[k] = -1;
        if (plan->par_grad[k]) { L->g_packed[k] = 4 * f; f += (int64_t)enslam_packed_grad_floats(k); f = (f + 3) & ~(int64_t)3; }
    }
    L->g_ro = L->g_rd = -1;
    if (plan->need_rays) { L->g_ro = 4 * f; f += 3 * (int64_t)N; L->g_rd = 4 * f; f += 3 * (int64_t)N; f = (f + 3) & ~(int64_t)3; }
    for (int k = 0; k < 4; ++k) {
        L->g_nat[k] = -1;
        if (plan->grid_mode[k] == 2) { L->g_nat[k] = 4 * f; f += 32 * plan_voxels(plan, k); }
    }
    f = (f + 3) & ~(int64_t)3;
    L->g_counter = 4 * f; f += 4;                                          // int32 work-list counter (+ padding)
    L->g_flat = 0;
    L->g_flat_floats = f;
    o = plan_align(4 * f);
    L->g_params = -1;
    if (any_par) { L->g_params = o; o = plan_align(o + 4 * plan->pgrad_floats); }
    for (int k = 0; k < 4; ++k) {
        L->g_dense[k] = -1;
        if (plan->grid_mode[k] == 3) { L->g_dense[k] = o; o = plan_align(o + 128 * plan_voxels(plan, k)); }
    }
    L->grad_bytes = o;
    // ---- outputs
    o = 0;
    L->o_depth = o; o = plan_align(o + 8 * (int64_t)N);
    L->o_var = o; o = plan_align(o + 8 * (int64_t)N);
    L->o_rgb = o; o = plan_align(o + 12 * (int64_t)N);
    L->o_loss = o; o = plan_align(o + 8);
    L->out_bytes = o;
    return ENSLAM_OK;
}

namespace {
// scene of a plan call: grid values (converted copy for mode 3), packed decoders
void plan_scene(const enslam_step_plan* plan, const enslam_step_layout* L, void* scratch, const float* const* grid_values, enslam_scene& sc) {
    for (int i = 0; i < 6; ++i) { sc.bound[i] = plan->bound[i]; sc.coarse_bound[i] = plan->coarse_bound[i]; }
    for (int k = 0; k < 4; ++k) {
        sc.grids[k].data = nullptr; sc.grids[k].D = sc.grids[k].H = sc.grids[k].W = 0; sc.packed[k] = nullptr;
        if (plan->grid_mode[k] == 0) continue;
        sc.grids[k].D = plan->grid_D[k]; sc.grids[k].H = plan->grid_H[k]; sc.grids[k].W = plan->grid_W[k];
        sc.grids[k].data = plan->grid_mode[k] == 3 ? at<float>(scratch, L->s_vm[k]) : const_cast<float*>(grid_values[k]);
        sc.packed[k] = at<float>(scratch, L->s_packed[k]);
    }
}
}  // namespace

int enslam_plan_forward(const enslam_step_plan* plan, const enslam_step_layout* L, void* scratch, void* grad, void* out,
                        const float* rays_o, const float* rays_d, const float* gt_depth, const float* gt_color,
                        const float* depth_max, const float* const* grid_values, void* stream) {
    if (!plan_ok(plan) || !L || !scratch || !out || !rays_o || !rays_d || !gt_depth || !grid_values) return ENSLAM_EINVAL;
    if (L->grad_bytes > 0 && !grad) return ENSLAM_EINVAL;
    const int N = plan->n_rays, S = L->n_samples;
    for (int k = 1; k < 4; ++k)
        if (plan->grid_mode[k] != 0 && !grid_values[k]) return ENSLAM_EINVAL;
    if (L->s_zero_bytes > 0 && hipMemsetAsync(scratch, 0, (size_t)L->s_zero_bytes, (hipStream_t)stream) != hipSuccess) return ENSLAM_ELAUNCH;
    // block marking of the grids with gradient
    enslam_scene msc;
    for (int i = 0; i < 6; ++i) { msc.bound[i] = plan->bound[i]; msc.coarse_bound[i] = plan->coarse_bound[i]; }
    uint8_t* fptr[4] = {nullptr, nullptr, nullptr, nullptr};
    bool marking = false;
    for (int k = 0; k < 4; ++k) {
        msc.grids[k].data = nullptr; msc.grids[k].D = msc.grids[k].H = msc.grids[k].W = 0; msc.packed[k] = nullptr;
        if (plan->grid_mode[k] >= 2) {
            msc.grids[k].D = plan->grid_D[k]; msc.grids[k].H = plan->grid_H[k]; msc.grids[k].W = plan->grid_W[k];
            fptr[k] = at<uint8_t>(scratch, L->s_flags[k]);
            marking = true;
        }
    }
    // prepare roles: pack every decoder of the stage, clear the accumulators
    int32_t kinds[3]; enslam_mlp_params structs[3]; float* packed[3]; int nd = 0;
    for (int k = 1; k < 4; ++k)
        if (plan->grid_mode[k] != 0) { kinds[nd] = k; structs[nd] = plan->params[k]; packed[nd] = at<float>(scratch, L->s_packed[k]); ++nd; }
    float* zdst[4]; int64_t zvox[4]; const uint8_t* zneed[4]; int nz = 0;
    const float* csrc[4]; float* cdst[4]; int64_t cvox[4]; const uint8_t* cneed[4]; int nc = 0;
    for (int k = 1; k < 4; ++k)
        if (plan->grid_mode[k] == 3) {
            zdst[nz] = at<float>(scratch, L->s_gacc[k]); zvox[nz] = plan_voxels(plan, k); zneed[nz] = fptr[k]; ++nz;
            csrc[nc] = grid_values[k]; cdst[nc] = at<float>(scratch, L->s_vm[k]); cvox[nc] = plan_voxels(plan, k); cneed[nc] = fptr[k]; ++nc;
        }
    float* flat = L->g_flat_floats > 0 ? at<float>(grad, L->g_flat) : nullptr;
    float* dmax = depth_max ? const_cast<float*>(depth_max) : at<float>(scratch, L->s_dmax);
    double* z = at<double>(scratch, L->s_z);
    int rc;
    if (L->merged) {
        rc = enslam_sample_prepare(N, plan->n_lin, plan->n_surf, rays_o, rays_d, gt_depth, plan->bound, plan->t_lin, plan->t_surf, plan->lindisp,
                                   nullptr, dmax, depth_max != nullptr, z, plan->stage, marking ? &msc : nullptr, marking ? fptr : nullptr, 64,
                                   nullptr, nd, kinds, structs, packed, nz, zdst, zvox, zneed, flat, L->g_flat_floats, stream);
        if (rc != ENSLAM_OK) return rc;
    } else {
        rc = enslam_sample_rays(N, plan->n_lin, plan->n_surf, rays_o, rays_d, gt_depth, plan->bound, plan->t_lin, plan->t_surf, plan->lindisp,
                                nullptr, dmax, depth_max != nullptr, z, plan->stage, marking ? &msc : nullptr, marking ? fptr : nullptr, stream);
        if (rc != ENSLAM_OK) return rc;
        rc = enslam_step_prepare(nd, kinds, structs, packed, nc, csrc, cdst, cvox, cneed, nullptr, nz, zdst, zvox, zneed, flat,
                                 L->g_flat_floats, stream);
        if (rc != ENSLAM_OK) return rc;
    }
    enslam_scene sc;
    plan_scene(plan, L, scratch, grid_values, sc);
    float* act = at<float>(scratch, L->s_act);
    if (plan->loss_kind == 0)
        return enslam_render_fwd(plan->stage, N, S, rays_o, rays_d, z, &sc, at<double>(out, L->o_depth), at<double>(out, L->o_var),
                                 at<float>(out, L->o_rgb), at<float>(scratch, L->s_raw), act, plan->act_light, stream);
    if (hipMemsetAsync(at<double>(out, L->o_loss), 0, 8, (hipStream_t)stream) != hipSuccess) return ENSLAM_ELAUNCH;
    const bool want_grad = L->grad_bytes > 0;
    return enslam_render_loss_fwd(plan->stage, N, S, rays_o, rays_d, z, &sc, at<double>(out, L->o_depth), at<double>(out, L->o_var),
                                  at<float>(out, L->o_rgb), at<float>(scratch, L->s_raw), act, plan->act_light, gt_depth,
                                  plan->use_color ? gt_color : nullptr, plan->w_color, at<double>(out, L->o_loss),
                                  want_grad ? at<float>(scratch, L->s_draw) : nullptr,
                                  (want_grad && plan->use_work_list) ? at<int32_t>(scratch, L->s_work) : nullptr,
                                  (want_grad && plan->use_work_list) ? at<int32_t>(grad, L->g_counter) : nullptr, stream);
}

int enslam_plan_backward(const enslam_step_plan* plan, const enslam_step_layout* L, void* scratch, void* grad, void* out,
                         const float* rays_o, const float* rays_d, const float* gt_depth, const float* gt_color,
                         const float* const* grid_values, const double* g_depth, const double* g_var, const float* g_rgb,
                         const double* g_loss, void* stream) {
    if (!plan_ok(plan) || !L || !scratch || !grad || !out || !rays_o || !rays_d || !grid_values) return ENSLAM_EINVAL;
    (void)gt_depth; (void)gt_color;
    const int N = plan->n_rays, S = L->n_samples;
    enslam_scene sc;
    plan_scene(plan, L, scratch, grid_values, sc);
    double* z = at<double>(scratch, L->s_z);
    int32_t* work = plan->use_work_list ? at<int32_t>(scratch, L->s_work) : nullptr;
    int32_t* wcount = plan->use_work_list ? at<int32_t>(grad, L->g_counter) : nullptr;
    float* d_raw = at<float>(scratch, L->s_draw);
    const double* d_scale = nullptr;
    int rc;
    if (plan->loss_kind == 1) {
        if (!g_loss) return ENSLAM_EINVAL;
        d_scale = g_loss;                                   // unit gradients from the forward, scaled inside the decoder backward
    } else {
        rc = enslam_composite_bwd_list(N, S, at<float>(scratch, L->s_raw), z, at<double>(out, L->o_depth), g_depth, g_var, g_rgb, d_raw,
                                       work, wcount, stream);
        if (rc != ENSLAM_OK) return rc;
    }
    enslam_grid gg[4];
    float* gpk[4]; float* gpart[4];
    for (int k = 0; k < 4; ++k) {
        gg[k].data = nullptr; gg[k].D = plan->grid_D[k]; gg[k].H = plan->grid_H[k]; gg[k].W = plan->grid_W[k];
        gpk[k] = nullptr; gpart[k] = nullptr;
        if (plan->grid_mode[k] == 2) gg[k].data = at<float>(grad, L->g_nat[k]);
        else if (plan->grid_mode[k] == 3) gg[k].data = at<float>(scratch, L->s_gacc[k]);
        if (plan->par_grad[k]) gpk[k] = at<float>(grad, L->g_packed[k]);
    }
    float* p_ro = plan->need_rays ? at<float>(grad, L->g_ro) : nullptr;
    float* p_rd = plan->need_rays ? at<float>(grad, L->g_rd) : nullptr;
    float* dgw = at<float>(scratch, L->s_dgw);
    rc = enslam_decoder_bwd_partials(plan->stage, N, S, rays_o, rays_d, z, &sc, d_raw, d_scale, at<float>(scratch, L->s_act), plan->act_light,
                                     dgw, gg, gpk, gpart, p_ro, p_rd, work, wcount, stream);
    if (rc != ENSLAM_OK) return rc;
    // finish launch: channel-major grid gradients, decoder gradients into the parameter region, ray gradients from the hand-off
    const float* csrc[4]; float* cdst[4]; int64_t cvox[4]; const uint8_t* cneed[4]; int nc = 0;
    for (int k = 1; k < 4; ++k)
        if (plan->grid_mode[k] == 3) {
            csrc[nc] = at<float>(scratch, L->s_gacc[k]); cdst[nc] = at<float>(grad, L->g_dense[k]); cvox[nc] = plan_voxels(plan, k);
            cneed[nc] = at<uint8_t>(scratch, L->s_flags[k]); ++nc;
        }
    int32_t kinds[4]; const float* pk[4]; const float* parts[4]; enslam_mlp_params gs[4]; int npk = 0;
    float* pbase = at<float>(grad, L->g_params);
    for (int k = 1; k < 4; ++k)
        if (plan->par_grad[k]) {
            kinds[npk] = k; pk[npk] = gpk[k]; parts[npk] = nullptr;
            enslam_mlp_params& g = gs[npk];
            const int64_t* off = plan->pgrad_off[k];
            int j = 0;
            for (int i = 0; i < 5; ++i) { g.W[i] = pbase + off[j++]; g.b[i] = pbase + off[j++]; }
            for (int i = 0; i < 5; ++i) { g.Wc[i] = pbase + off[j++]; g.bc[i] = pbase + off[j++]; }
            g.Wo = pbase + off[j++]; g.bo = pbase + off[j++]; g.B = pbase + off[j++];
            ++npk;
        }
    const bool ray_pending = dgw != nullptr && plan->need_rays;
    if (nc == 0 && npk == 0 && !ray_pending) return ENSLAM_OK;
    return enslam_step_finish_native(nc, csrc, cdst, cvox, cneed, nullptr, npk, kinds, pk, parts, gs, plan->stage, ray_pending ? N : 0, S, rays_o,
                                     rays_d, z, &sc, ray_pending ? dgw : nullptr, p_ro, p_rd, work, wcount, nullptr, nullptr, 0, stream);
}
