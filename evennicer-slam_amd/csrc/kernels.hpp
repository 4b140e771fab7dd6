// Internal launcher declarations shared by the C ABI translation unit and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hpp"
#include "raygrad.hpp"

// Function attributes (dynamic LDS limits) and CU counts belong to a DEVICE, and the reference lets one process render on two
// (cfg['tracking']['device'] != cfg['mapping']['device']): the launchers keep their one-time flags and cached counts per device
// ordinal, not per process.
constexpr int ENS_MAX_DEVICES = 16;
inline int ens_device_ordinal() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < ENS_MAX_DEVICES) ? d : 0;
}
inline int ens_device_cus() {
    static int cus[ENS_MAX_DEVICES] = {};
    const int d = ens_device_ordinal();
    if (cus[d] == 0) {
        hipDeviceProp_t prop;
        cus[d] = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[d];
}

// Mapper RGB-D loss (Mapper.py:553-562) fused into the compositing launches: gd != null switches it on
struct LossSpec {
    const float* gd;             // gt depth [N] (term only where > 0)
    const float* gc;             // gt colour [N,3] or null (depth term only)
    float w;                     // colour weight
    double* loss;                // forward: += sum of the rays' terms (zero on entry)
    const double* g_loss;        // backward: d(total)/d(loss), device scalar
    float* d_raw_unit;           // forward (optional): d(loss)/d(raw) for g_loss = 1, [N*S,4] -- lets the backward start at the
                                 // decoders (ens_launch_decoder_bwd's draw_scale = g_loss) without a compositing launch
};
// Tracker's RGB-D loss (Tracker.py:176-195) folded into the compositing launch: the LossSpec fields gd / gc / w / loss /
// d_raw_unit as for the mapper loss, plus what only the tracker has
struct TrackerSpec {
    const uint8_t* inside;       // [N] rays the in-bound prefilter keeps (Tracker.py:164-174, applied as a mask) or null: all
    int dynamic;                 // handle_dynamic (:180-182): keep rays with tmp < 10 * median(tmp over the kept rays)
    double* tmp;                 // [N] scratch: |gd - depth| / sqrt(var + 1e-10)
};
// Work list of the saved-activation backward: the 16-sample tiles (ray * ntl + tl) whose d_raw is not all zero, appended
// ray by ray (a ray's tiles stay adjacent) by the kernel that produces d_raw; count[0] must be zero before that kernel.
// Behind a converged surface the transmittance underflows to 0 and the far tiles of most rays drop out.
struct WorkList {
    int* tiles;                  // [n_rays * ntl]
    int* count;                  // [1]
};

// one segment of a decoder re-layout: dst[r*dst_ld + c] = T ? src[c*src_ld + r] : src[r*src_ld + c]
struct PackSeg {
    float* src;                              // caller tensor (input of pack, output of unpack)
    int off;                                 // float offset in the packed buffer
    unsigned short rows, cols;               // extent in packed orientation
    unsigned short src_ld, dst_ld;
    unsigned char transpose, dec;            // transpose: bit 0 transposed copy, bit 1 XOR-swizzled chunks; dec: index into PackJob::packed
};
#define ENS_MAX_SEGS 120
struct PackJob {
    PackSeg seg[ENS_MAX_SEGS];
    float* packed[4];                        // packed buffers of the decoders of this job
    const float* part[4];                    // unpack only: per-workgroup partial images to sum instead of packed[] (flush_image,
    int part_stride[4];                      // render_bwd.hip): [16-float header, word 0 = rows | rows x part_stride floats]
    int n;
};
struct AdamJob {                 // masked Adam over up to 4 voxel-major grids in one launch (Mapper.py:328-361,573-602)
    float* p[4];                 // parameters   [V,32]
    float* g[4];                 // gradients    [V,32]  (consumed and cleared)
    float* m[4];                 // exp_avg      [V,32]
    float* v[4];                 // exp_avg_sq   [V,32]
    const uint8_t* mask[4];      // [V] 1 = optimisable voxel (null: every voxel)
    int64_t V[4];
    const double* lr[4];         // device scalars: learning rate of each grid (float64 like the Python float)
    const int* step[4];          // device scalars: Adam step count of each grid (already incremented; <= 0: clear only)
    int block_begin[5];
    int n;
    double beta1, beta2, eps;    // Python floats of the optimiser
};
#define ENS_ADAM_MAX_TENSORS 72
struct AdamTensorsJob {          // torch.optim.Adam over a list of small dense tensors (decoder parameters) in one launch
    float* p[ENS_ADAM_MAX_TENSORS];
    const float* g[ENS_ADAM_MAX_TENSORS];
    float* m[ENS_ADAM_MAX_TENSORS];
    float* v[ENS_ADAM_MAX_TENSORS];
    int numel[ENS_ADAM_MAX_TENSORS];
    int block_begin[ENS_ADAM_MAX_TENSORS + 1];      // 1024 elements per workgroup
    int n;
    double beta1, beta2, eps;
    const double* lr;            // device scalar
    const int* step;             // device scalar (already incremented; self_inc: the count BEFORE this step)
    int self_inc;                // single-workgroup jobs only: use step[0] + 1 and store it back (no separate increment launch)
};
// Gradient bucket of the ray-sharded step (parallel.py): the touched 64-voxel blocks of up to 4 feature-grid gradients
// followed by up to ENS_ADAM_MAX_TENSORS small dense tensors, packed into / unpacked from one flat all-reduce buffer.
struct BucketJob {
    float* grid[4];              // gradient of grid g: [C, V] (layout 0, the reference's [1,C,D,H,W]) or [V, C] (layout 1)
    int64_t V[4];
    int layout[4];
    int blk_begin[5];            // block range of each grid inside flags / pos (ceil(V / 64) blocks per grid)
    int n_grids, C;
    int bs;                      // voxels per block (64, 32, 16 or 8): granularity of flags / pos / bucket slots
    const uint8_t* flags;        // [blk_begin[n_grids]] union over ranks of the touched blocks
    const int* pos;              // inclusive prefix sum of flags: a flagged block b owns bucket slot pos[b] - 1 (C * 64 floats)
    float* small[ENS_ADAM_MAX_TENSORS];
    int numel[ENS_ADAM_MAX_TENSORS];
    int small_blk_begin[ENS_ADAM_MAX_TENSORS + 1];      // 1024 elements per workgroup
    int64_t small_off[ENS_ADAM_MAX_TENSORS];            // float offset of each small tensor in the bucket
    int n_small;
    float* bucket;
};
int ens_launch_bucket(const BucketJob& job, bool unpack, hipStream_t st);
struct ConvJob {                 // up to 4 grids converted in one launch
    const float* src[4];
    float* dst[4];
    int64_t V[4];
    const uint8_t* need[4];      // sparse path: per-64-voxel-block flags (null: dense)
    uint8_t* valid[4];           // sparse to-voxel-major: blocks already converted
    int block_begin[5];
    int n;
};

int ens_launch_pack(const PackJob& job, float* packed, bool unpack, hipStream_t st);
int ens_launch_transpose(const float* src, float* dst, int64_t n_vox, bool to_voxel_major, hipStream_t st);
int ens_launch_convert(const ConvJob& job, bool to_voxel_major, hipStream_t st);
int ens_launch_ray_grad_bwd(int stage, int ntl, int n_rays, const float* ro, const float* rd, const double* z,
                            const DevScene& sc, float* dgrid_ws, float* g_ro, float* g_rd, hipStream_t st);
int ens_launch_tracker_loss(int n, const double* depth, const double* unc, const float* color, const float* gd,
                            const float* gc, float w, const double* g_loss, double* loss, double* g_depth, float* g_color,
                            hipStream_t st);
int ens_launch_gather_pixels(int n, const int64_t* idx, int H0, int W0, int ww, int Wimg, const float* depth, const void* color,
                             int color_f64, float* oi, float* oj, float* od, void* oc, hipStream_t st);
int ens_launch_pose_rays(int n, const float* ct, const float* pi, const float* pj, float fx, float fy, float cx, float cy,
                         const float* g_ro, const float* g_rd, float* ro, float* rd, float* g_ct, hipStream_t st);
int ens_launch_step(const PackJob& pj, bool unpack, const ConvJob& cj, bool to_vm, const ConvJob& zj, float* flat,
                    int64_t n_flat, const RayGradArgs* rg, hipStream_t st, uint8_t* mv_need = nullptr, uint8_t* mv_prev = nullptr,
                    int64_t n_move = 0);
bool ens_ray_grad_args(int stage, int ntl, int n_rays, const float* ro, const float* rd, const double* z, const DevScene& sc,
                       float* dgrid_ws, float* g_ro, float* g_rd, RayGradArgs& A, const WorkList* wl = nullptr);
int ens_launch_adam(const AdamJob& job, hipStream_t st);
int ens_launch_adam_tensors(const AdamTensorsJob& job, hipStream_t st);
int ens_launch_zero_blocks(const ConvJob& job, float* flat, int64_t n_flat, hipStream_t st);
int ens_launch_mark_blocks(int stage, int n_rays, int S, const float* ro, const float* rd, const double* z,
                           const DevScene& sc, uint8_t* const* flags, hipStream_t st, int block_voxels = 64);
struct MarkArgs {                // optional block marking inside the sampler (mark_blocks_kernel's work, one launch less)
    DevScene sc;                 // bounds and grid dims (data pointers unused)
    int kmask;                   // grids to mark (bit k)
    uint8_t* flags[4];
    int shift;                   // log2(voxels per flag): 6 (the renderer's own 64-voxel blocks) .. 3
    uint8_t* flags64[4];         // optional (shift < 6): the same marks at 64 voxels per flag as well
};
struct SampleArgs {              // sample_kernel's scalar arguments (Renderer.py:95-171)
    int n_rays, n_lin, n_surf, lindisp, dmax_inline;
    const float* ro;
    const float* rd;
    const float* gd;             // gt depth [n_rays] or null
    double lo[3], hi[3];         // Renderer.bound
    const float* t_lin;
    const double* t_surf;
    const float* t_rand;         // perturbation draws or null
    const float* dmax;           // {max(gd), fl32(max * 1.2f)} when not reduced inline
    double* zout;
};
int ens_launch_sample_prepare(int n_rays, int n_lin, int n_surf, const float* ro, const float* rd, const float* gd,
                              const double* bound, const float* t_lin, const double* t_surf, int lindisp, const float* t_rand,
                              float* scratch, int dmax_given, double* z, const MarkArgs* mark, const PackJob& pj, const ConvJob& zj,
                              float* flat, int64_t n_flat, hipStream_t st);
int ens_launch_sample(int n_rays, int n_lin, int n_surf, const float* ro, const float* rd, const float* gd,
                      const double* bound, const float* t_lin, const double* t_surf, int lindisp,
                      const float* t_rand, float* scratch, int dmax_given, double* z, const MarkArgs* mark, hipStream_t st);
int ens_launch_rgbd_loss(int n, const double* depth, const float* color, const float* gd, const float* gc, float w,
                         const double* g_loss, double* loss, double* g_depth, float* g_color, hipStream_t st);
int ens_launch_ray_points(int n_rays, int S, const float* ro, const float* rd, const double* z, const double* bound,
                          double* pts, uint8_t* mask, hipStream_t st);
int ens_launch_voxel_index(int64_t n, const double* pts, const double* bound, int D, int H, int W, int* ix, int* iy,
                           int* iz, float* fx, float* fy, float* fz, hipStream_t st);
int ens_launch_sincos(int64_t n, const float* x, float* s, float* c, hipStream_t st);
int ens_launch_render_fwd(int stage, int ntl, int64_t n_units, const float* ro, const float* rd, const double* z,
                          const double* pts, int64_t n_points, int apply_mask, const DevScene& sc, double* depth,
                          double* var, float* rgb, float* raw, float* act_ws, int act_light, hipStream_t st,
                          const LossSpec* ls = nullptr, const WorkList* wl = nullptr, const TrackerSpec* ts = nullptr);
int ens_launch_composite_fwd(int n_rays, int S, const float* raw, const double* z, double* depth, double* var,
                             float* rgb, float* weights, hipStream_t st, const LossSpec* ls = nullptr,
                             const WorkList* wl = nullptr);
constexpr int ENS_TRACKER_TAIL_MAX_RAYS = 4096;       // one workgroup sorts the batch in LDS for the median
int ens_launch_tracker_tail(int n_rays, int S, const float* raw, const double* z, double* depth, double* var, float* rgb,
                            const LossSpec& ls, const TrackerSpec& ts, const WorkList* wl, hipStream_t st);
int ens_launch_tracker_rays(int n, const float* ct, const int64_t* idx, int H0, int W0, int ww, int Wimg, int Himg, const float* depth,
                            const void* color, int color_f64, float fx, float fy, float cx, float cy, const double* bound,
                            float* oi, float* oj, float* ro, float* rd, float* gd, float* gc, uint8_t* inside, float* dmax,
                            hipStream_t st, int* iter = nullptr, int iter_count = 0);
int ens_launch_composite_bwd(int n_rays, int S, const float* raw, const double* z, const double* depth,
                             const double* g_depth, const double* g_var, const float* g_rgb, float* d_raw,
                             hipStream_t st, const LossSpec* ls = nullptr, const float* rgb = nullptr,
                             const WorkList* wl = nullptr);
int ens_launch_decoder_bwd(int stage, int ntl, int n_rays, const float* ro, const float* rd, const double* z,
                           const DevScene& sc, const float* d_raw, const float* act_ws, int act_light, float* dgrid_ws,
                           const DevGrid* grad_grids, float* const* grad_packed, float* g_ro, float* g_rd,
                           hipStream_t st, const double* draw_scale = nullptr, const WorkList* wl = nullptr,
                           float* const* grad_partial = nullptr);
// feature-gradient scatter from the decoder kernels' hand-off (grid_scatter.hip); grad_grids indexed by decoder kind, data null: not scattered;
// d_raw null: every tile was handed off, else only the tiles whose d_raw is not all zero (the work list's criterion)
int ens_launch_grid_scatter(int stage, int ntl, int n_rays, const float* ro, const float* rd, const double* z, const DevScene& sc,
                            const float* dgrid_ws, const float* d_raw, const DevGrid* grad_grids, hipStream_t st);
int ens_bwd_max_workgroups();      // upper bound of the workgroups of one decoder role (rows of a partial buffer)
