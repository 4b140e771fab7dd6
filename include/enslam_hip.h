/*
 * enslam_hip.h -- C ABI of the MI355X (gfx950) volume-rendering hot path.
 *
 * The reference (cs-vision/EvenNICER-SLAM) has no FFI on this path: its boundary
 * is a Python object protocol.  This library is what a
 * Python (ctypes) host binds instead of the PyTorch op sequence; each entry
 * point names the reference code it replaces (paths relative to the reference
 * checkout).  All pointers are DEVICE pointers unless marked "host".  Nothing
 * here allocates, frees, synchronises or takes ownership: the caller owns every
 * buffer, every call is stream-ordered on `stream` (a hipStream_t passed as
 * void*), re-entrant, and returns 0 on success or a negative ENSLAM_E* code.
 *
 * Memory layouts
 *   rays_o, rays_d      float32 [N,3]
 *   gt_depth            float32 [N]            (may be NULL: no depth guidance)
 *   z_vals              float64 [N,S]          sorted sample distances
 *   grid (reference)    float32 [1,32,D,H,W]   "channel-major", as the callers hold it
 *   grid (kernel)       float32 [D*H*W,32]     "voxel-major": one voxel corner = 128 contiguous bytes
 *   raw                 float32 [N*S,4]        (r,g,b,occ) after the out-of-bound mask
 *   depth, var          float64 [N];   rgb float32 [N,3]
 *   packed MLP          float32 [enslam_packed_floats(kind)]   (see enslam_pack_mlp)
 */
#ifndef ENSLAM_HIP_H
#define ENSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ENSLAM_ABI_VERSION 1

/* error codes */
#define ENSLAM_OK 0
#define ENSLAM_EINVAL (-1)      /* bad argument (NULL where required, unsupported size) */
#define ENSLAM_ELAUNCH (-2)     /* hipLaunch / hipMemsetAsync reported an error */
#define ENSLAM_EUNSUPPORTED (-3)

/* stages of NICE.forward, src/conv_onet/models/decoder.py:312-342 */
#define ENSLAM_STAGE_COARSE 0
#define ENSLAM_STAGE_MIDDLE 1
#define ENSLAM_STAGE_FINE 2
#define ENSLAM_STAGE_COLOR 3

/* decoder kinds (index into the arrays below) */
#define ENSLAM_MLP_COARSE 0     /* MLP_no_xyz, decoder.py:206-274 */
#define ENSLAM_MLP_MIDDLE 1     /* MLP,        decoder.py:91-203  */
#define ENSLAM_MLP_FINE 2       /* MLP with concat_feature (c_dim 64) */
#define ENSLAM_MLP_COLOR 3      /* MLP with 4 outputs */

/* Parameter tensors of one decoder, in the reference's state_dict layout
 * (row-major nn.Linear weights [out,in]).  Used as input by enslam_pack_mlp and
 * as OUTPUT (gradient tensors of the same shapes) by enslam_unpack_mlp_grads. */
typedef struct enslam_mlp_params {
    float *W[5];    /* pts_linears.{i}.weight  [32,K_i]; K = 93,32,32,125,32 (coarse: 32,32,32,64,32) */
    float *b[5];    /* pts_linears.{i}.bias    [32] */
    float *Wc[5];   /* fc_c.{i}.weight         [32,c_dim] (c_dim 32; fine 64); NULL for coarse */
    float *bc[5];   /* fc_c.{i}.bias           [32]; NULL for coarse */
    float *Wo;      /* output_linear.weight    [n_out,32]  (n_out 1; color 4) */
    float *bo;      /* output_linear.bias      [n_out] */
    float *B;       /* embedder._B             [3,93]; NULL for coarse */
} enslam_mlp_params;

/* One feature grid in voxel-major layout. */
typedef struct enslam_grid {
    float *data;    /* [D*H*W,32] (const for the forward; accumulation target for gradients) */
    int32_t D, H, W;
} enslam_grid;

/* Everything a render call reads.  grids/packed are indexed by ENSLAM_MLP_*.
 * Entries the stage does not use may be NULL. */
typedef struct enslam_scene {
    double bound[6];         /* host: x_lo,x_hi,y_lo,y_hi,z_lo,z_hi  (Renderer.bound, Renderer.py:20)   */
    double coarse_bound[6];  /* host: bound * coarse_bound_enlarge (EvenNICER_SLAM.py:182)             */
    enslam_grid grids[4];    /* coarse, middle, fine, color                                            */
    const float *packed[4];  /* packed decoders (enslam_pack_mlp)                                       */
} enslam_scene;

int enslam_abi_version(void);
const char *enslam_arch(void);          /* "gfx950" */

/* number of float32 in the packed form of decoder `kind` (forward + transposed sections) and
 * in its gradient accumulator (forward section only). */
size_t enslam_packed_floats(int kind);
size_t enslam_packed_grad_floats(int kind);

/* Re-layout one decoder's parameters for the MFMA kernels (zero-padded K, transposed copies for
 * the backward).  `packed` must have been zero-filled once by the caller; padding is never written.
 * Replaces nothing in the reference (layout glue for decoder.py:149-159 parameters). */
int enslam_pack_mlp(int kind, const enslam_mlp_params *params, float *packed, void *stream);
/* The same for several decoders in ONE launch (host arrays of length n <= 3). */
int enslam_pack_mlp_multi(int32_t n, const int32_t *kinds, const enslam_mlp_params *params, float *const *packed,
                          void *stream);

/* Inverse for gradients: packed_grad (accumulated by enslam_render_bwd) -> tensors shaped like the
 * reference parameters.  Pointers in `grads` that are NULL are skipped. */
int enslam_unpack_mlp_grads(int kind, const float *packed_grad, const enslam_mlp_params *grads, void *stream);
/* The same for several decoders in ONE launch (host arrays of length n <= 4). */
int enslam_unpack_mlp_grads_multi(int32_t n, const int32_t *kinds, const float *const *packed_grads,
                                  const enslam_mlp_params *grads, void *stream);

/* [C,V] -> [V,C] and back (C = 32).  Replaces the implicit layout of F.grid_sample's input
 * (decoder.py:173-174).  `to` reads the caller's grid; `from` writes a gradient the caller's
 * autograd expects ([1,32,D,H,W]). */
int enslam_grid_to_voxel_major(const float *src, float *dst, int64_t n_voxels, void *stream);
int enslam_grid_from_voxel_major(const float *src, float *dst, int64_t n_voxels, void *stream);
/* The same for up to 4 grids in ONE launch (n <= 4; src/dst/n_voxels are host arrays of length n). */
int enslam_grids_convert(int32_t n, const float *const *src, float *const *dst, const int64_t *n_voxels,
                         int32_t to_voxel_major, void *stream);

/* Sparse layout path.  A batch touches only a few percent of a grid; blocks of 64 consecutive voxels are the unit.
 * enslam_mark_blocks: flags[k][b] = 1 for every block of grid k (stage's grids) holding a corner that a sample of
 *   the batch reads (flags: uint8 [ceil(V_k/64)], caller-zeroed; entries of unused grids may be NULL).
 * enslam_grids_convert_sparse, to_voxel_major != 0: converts blocks with need && !valid and sets valid (valid: uint8
 *   bitmap that lives with the voxel-major copy; zero it when the source grid changes).
 *   to_voxel_major == 0 (gradients back to [32,V]): blocks with need are transposed, all others written as zeros.
 * enslam_zero_blocks: zero-fills the flagged blocks of voxel-major buffers (gradient accumulators) and, in the same
 *   launch, the flat float range [flat, flat + n_flat) (the small decoder / ray accumulators; may be NULL / 0). */
int enslam_mark_blocks(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                       const double *z_vals, const enslam_scene *scene, uint8_t *const *flags, void *stream);
/* enslam_mark_blocks / enslam_bucket_pack / enslam_bucket_unpack at a finer block granularity: block_voxels in {64, 32, 16, 8}
 * voxels per flag (flags uint8 [ceil(n_voxels / block_voxels)] per grid, pos and bucket slots accordingly).  The bucket of a
 * ray-sharded step carries the union over ranks of the touched blocks; measured on one real step (room0, 1000 rays): 6.52 MB at
 * 64 voxels per block, 3.35 MB at 16 (the non-zero entries are 1.45 MB).  New functionality (the reference has no multi-GPU
 * path). */
int enslam_mark_blocks_g(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                         const double *z_vals, const enslam_scene *scene, uint8_t *const *flags, int32_t block_voxels,
                         void *stream);
int enslam_bucket_pack_g(int32_t n_grids, const float *const *grid_grad, int32_t channels, const int64_t *n_voxels,
                         const int32_t *layout, const uint8_t *flags, const int32_t *pos, int32_t n_small,
                         const float *const *small, const int64_t *small_numel, int64_t small_base, float *bucket,
                         int32_t block_voxels, void *stream);
int enslam_bucket_unpack_g(int32_t n_grids, float *const *grid_grad, int32_t channels, const int64_t *n_voxels,
                           const int32_t *layout, const uint8_t *flags, const int32_t *pos, int32_t n_small,
                           float *const *small, const int64_t *small_numel, int64_t small_base, const float *bucket,
                           int32_t block_voxels, void *stream);
int enslam_grids_convert_sparse(int32_t n, const float *const *src, float *const *dst, const int64_t *n_voxels,
                                const uint8_t *const *need, uint8_t *const *valid, int32_t to_voxel_major,
                                void *stream);
int enslam_zero_blocks(int32_t n, float *const *dst, const int64_t *n_voxels, const uint8_t *const *need,
                       float *flat, int64_t n_flat, void *stream);

/* Tracker glue (RGB-D part).
 * enslam_pose_rays_fwd: camera tensor float32 [7] = (qr,qi,qj,qk, tx,ty,tz), pixels (pix_i = column, pix_j = row,
 *   float32 [n]) -> rays_o, rays_d float32 [n,3]: quad2rotation + get_camera_from_tensor (common.py:189-229) and
 *   get_rays_from_uv (:74-89) in one launch.  enslam_pose_rays_bwd: ray gradients (either may be NULL) ->
 *   g_camera_tensor float32 [7] (written, not accumulated); one workgroup, float64 sums, deterministic.
 * enslam_tracker_loss_fwd/bwd: Tracker.py:179-195 with handle_dynamic off:
 *   sum_{gt_depth>0} |gt_depth - depth| / sqrt(uncertainty + 1e-10)  +  w_color * sum_{gt_depth>0} |gt_color - color|
 *   (colour term when color/gt_color are given); the uncertainty carries no gradient (:179). */
int enslam_pose_rays_fwd(int32_t n, const float *camera_tensor, const float *pix_i, const float *pix_j, float fx,
                         float fy, float cx, float cy, float *rays_o, float *rays_d, void *stream);
/* Pixel samples of a frame behind the caller's single torch.randint draw (common.py:125-141, get_sample_uv / select_uv):
 * pixel_index int64 [n] in [0, window_h * window_w) of the window starting at (h0, w0) of an image_h x image_w frame ->
 * pix_i (column) / pix_j (row) float32 [n], depth_out float32 [n] from depth float32 [image_h, image_w], color_out [n,3] from
 * color [image_h, image_w, 3] (float32, or float64 with color_is_f64: the dataset readers hand out float64 colours).  The
 * caller keeps indices inside the window (they come from randint(window_h * window_w)); nothing is range-checked on the device. */
int enslam_gather_pixels(int32_t n, const int64_t *pixel_index, int32_t h0, int32_t w0, int32_t window_w, int32_t image_w,
                         int32_t image_h, const float *depth, const void *color, int32_t color_is_f64, float *pix_i,
                         float *pix_j, float *depth_out, void *color_out, void *stream);
int enslam_pose_rays_bwd(int32_t n, const float *camera_tensor, const float *pix_i, const float *pix_j, float fx,
                         float fy, float cx, float cy, const float *g_rays_o, const float *g_rays_d,
                         float *g_camera_tensor, void *stream);
int enslam_tracker_loss_fwd(int32_t n, const double *depth, const double *uncertainty, const float *color,
                            const float *gt_depth, const float *gt_color, float w_color, double *loss, void *stream);
int enslam_tracker_loss_bwd(int32_t n, const double *depth, const double *uncertainty, const float *color,
                            const float *gt_depth, const float *gt_color, float w_color, const double *g_loss,
                            double *g_depth, float *g_color, void *stream);

/* Mapper glue.  torch.optim.Adam (defaults) on voxel-major grids [V,32], restricted to the voxels whose
 * mask byte is 1 -- how Mapper.optimize_map optimises val_grad = val[mask] (Mapper.py:328-361, 573-575) without the
 * val[mask] = val_grad re-materialisation (:448-458, :596-602).  Per grid i: param / grad / exp_avg / exp_avg_sq are
 * float32 [n_voxels[i]*32]; mask uint8 [n_voxels[i]] or NULL (all voxels); lr[i] (float64) and step[i] (int32) are
 * DEVICE scalars (learning rate; Adam step count t >= 1 of this update, t <= 0: only clear the gradients) so that a captured
 * launch follows stage changes.  The gradients are cleared (everywhere, masked or not) by the same launch. */
int enslam_adam_masked(int32_t n, float *const *param, float *const *grad, float *const *exp_avg,
                       float *const *exp_avg_sq, const uint8_t *const *mask, const int64_t *n_voxels,
                       const double *const *lr, const int32_t *const *step, double beta1, double beta2, double eps,
                       void *stream);

/* The same Adam update for up to 72 small dense tensors (the decoder parameters the mapper optimises,
 * Mapper.py:363-369) in one launch; lr (float64) and step (int32, count of this update) are device scalars. */
int enslam_adam_tensors(int32_t n, float *const *param, const float *const *grad, float *const *exp_avg,
                        float *const *exp_avg_sq, const int64_t *numel, const double *lr, const int32_t *step,
                        double beta1, double beta2, double eps, void *stream);
/* enslam_adam_tensors for jobs of ONE workgroup (at most 1024 parameters in one tensor -- a camera tensor): *step holds the count
 * BEFORE this step; the launch uses *step + 1 and stores it back, so no separate increment launch precedes it.  More than one
 * workgroup: ENSLAM_EUNSUPPORTED. */
int enslam_adam_tensors_step(int32_t n, float *const *param, const float *const *grad, float *const *exp_avg,
                             float *const *exp_avg_sq, const int64_t *numel, const double *lr, int32_t *step, double beta1,
                             double beta2, double eps, void *stream);

/* Gradient bucket of the ray-sharded step (new functionality, SURVEY.md 8e: the reference has no multi-GPU path; the
 * tensors are the leaf gradients of Mapper.py:573-575).  Packs into / unpacks from one flat all-reduce buffer:
 *   - of up to 4 feature-grid gradients with `channels` channels each -- layout[g] 0: [channels, n_voxels[g]] (the
 *     reference's [1,C,D,H,W]), 1: voxel-major [n_voxels[g], channels] -- only the 64-voxel blocks whose flag is set.
 *     flags (uint8) and pos (int32, the INCLUSIVE prefix sum of flags) run over the blocks of all grids concatenated,
 *     ceil(n_voxels[g] / 64) per grid; flagged block b occupies bucket floats [(pos[b]-1) * channels * 64, +channels*64)
 *     in the gradient's own layout ([channels][64] or [64][channels]); voxels past the end of a grid travel as zeros;
 *   - then n_small (<= 72) dense tensors, concatenated from float offset small_base on.
 * The flags must be the union over ranks (so every rank lays the bucket out identically); unpack writes only the
 * flagged blocks and the small tensors. */
int enslam_bucket_pack(int32_t n_grids, const float *const *grid_grad, int32_t channels, const int64_t *n_voxels,
                       const int32_t *layout, const uint8_t *flags, const int32_t *pos, int32_t n_small,
                       const float *const *small, const int64_t *small_numel, int64_t small_base, float *bucket,
                       void *stream);
int enslam_bucket_unpack(int32_t n_grids, float *const *grid_grad, int32_t channels, const int64_t *n_voxels,
                         const int32_t *layout, const uint8_t *flags, const int32_t *pos, int32_t n_small,
                         float *const *small, const int64_t *small_numel, int64_t small_base, const float *bucket,
                         void *stream);

/* The small independent jobs around a render call in ONE launch each (they are 4-6 us kernels when issued separately):
 * enslam_step_prepare = enslam_pack_mlp_multi (n_dec decoders) + enslam_grids_convert_sparse to voxel-major (n_conv
 *   grids) + enslam_zero_blocks (n_zero accumulators and the flat range); any of the three parts may be empty.
 * enslam_step_finish  = enslam_grids_convert_sparse back to [32,V] (gradients) + enslam_unpack_mlp_grads_multi. */
int enslam_step_prepare(int32_t n_dec, const int32_t *kinds, const enslam_mlp_params *params, float *const *packed,
                        int32_t n_conv, const float *const *src, float *const *dst, const int64_t *n_voxels,
                        const uint8_t *const *need, uint8_t *const *valid, int32_t n_zero, float *const *zero_dst,
                        const int64_t *zero_voxels, const uint8_t *const *zero_need, float *flat, int64_t n_flat,
                        void *stream);
int enslam_step_finish(int32_t n_conv, const float *const *src, float *const *dst, const int64_t *n_voxels,
                       const uint8_t *const *need, int32_t n_dec, const int32_t *kinds,
                       const float *const *packed_grads, const enslam_mlp_params *grads, void *stream);
/* enslam_step_finish with enslam_ray_grad_bwd riding in the same launch (both depend only on enslam_decoder_bwd):
 * one dependent launch fewer per step.  stage COARSE or n_rays == 0: plain enslam_step_finish. */
int enslam_step_finish_rays(int32_t n_conv, const float *const *src, float *const *dst, const int64_t *n_voxels,
                            const uint8_t *const *need, int32_t n_dec, const int32_t *kinds,
                            const float *const *packed_grads, const enslam_mlp_params *grads, int32_t stage,
                            int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                            const double *z_vals, const enslam_scene *scene, float *dgrid_ws, float *g_rays_o,
                            float *g_rays_d, const int32_t *work_list, const int32_t *work_count, void *stream);
/* enslam_step_finish_rays into PERSISTENT gradient tensors: dst[i] is the same memory from call to call (the dense
 * gradient of a grid inside a captured hipGraph step) and prev[i] (uint8 per 64-voxel block, as need[i]; all zero and
 * dst[i] all zero before the first call) holds the blocks the previous call wrote.  Blocks touched neither then nor now
 * keep their zeros and are not written; the others are written / cleared as in enslam_step_finish_rays, and need[i] is
 * MOVED to prev[i] on the way (need[i] is all zero afterwards: a captured step keeps its flags in memory that nothing
 * re-zeroes between replays; read the touched blocks from prev[i]).  The result in dst is identical; the 48 MB zero-fill of room0's three dense gradients
 * becomes ~6 MB.  prev NULL: enslam_step_finish_rays.  Replaces nothing in the reference: autograd allocates a fresh dense
 * gradient per backward there (Mapper.py:573-575 then reads three mostly-zero 16 MB tensors). */
int enslam_step_finish_rays_prev(int32_t n_conv, const float *const *src, float *const *dst, const int64_t *n_voxels,
                                 const uint8_t *const *need, uint8_t *const *prev, int32_t n_dec, const int32_t *kinds,
                                 const float *const *packed_grads, const enslam_mlp_params *grads, int32_t stage,
                                 int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                                 const double *z_vals, const enslam_scene *scene, float *dgrid_ws, float *g_rays_o,
                                 float *g_rays_d, const int32_t *work_list, const int32_t *work_count, void *stream);

/* The general finish launch: enslam_step_finish_rays_prev (prev NULL: no persistent destinations; dgrid_ws NULL or
 * n_rays == 0: no ray-gradient role) which, for every decoder i with grad_partials[i] != NULL, forms the gradient as the SUM
 * of the per-workgroup partial images enslam_decoder_bwd_partials left there instead of reading packed_grads[i]. */
int enslam_step_finish_partials(int32_t n_conv, const float *const *src, float *const *dst, const int64_t *n_voxels,
                                const uint8_t *const *need, uint8_t *const *prev, int32_t n_dec, const int32_t *kinds,
                                const float *const *packed_grads, const float *const *grad_partials,
                                const enslam_mlp_params *grads, int32_t stage, int32_t n_rays, int32_t n_samples,
                                const float *rays_o, const float *rays_d, const double *z_vals, const enslam_scene *scene,
                                float *dgrid_ws, float *g_rays_o, float *g_rays_d, const int32_t *work_list,
                                const int32_t *work_count, void *stream);

/* enslam_step_finish_partials plus the flag hand-over for gradients that live in the kernels' own layout (feature grids given
 * as channels_last_3d tensors: their storage IS [V][32], nothing is converted on the way in and the gradient buffer the backward
 * adds into IS the tensor autograd receives, so no transposed-back copy exists whose launch could carry the flags): for the
 * n_move block flags at move_need (uint8, the flags this step's sampler marked, all such grids back to back) the launch does
 * move_prev[b] = move_need[b]; move_need[b] = 0.  `move_prev` is what the next step's enslam_sample_prepare clears the
 * persistent gradient by (zero_need) and what the gradient bucket reads.  n_move = 0: enslam_step_finish_partials. */
int enslam_step_finish_native(int32_t n_conv, const float *const *src, float *const *dst, const int64_t *n_voxels,
                              const uint8_t *const *need, uint8_t *const *prev, int32_t n_dec, const int32_t *kinds,
                              const float *const *packed_grads, const float *const *grad_partials,
                              const enslam_mlp_params *grads, int32_t stage, int32_t n_rays, int32_t n_samples,
                              const float *rays_o, const float *rays_d, const double *z_vals, const enslam_scene *scene,
                              float *dgrid_ws, float *g_rays_o, float *g_rays_d, const int32_t *work_list,
                              const int32_t *work_count, uint8_t *move_need, uint8_t *move_prev, int64_t n_move, void *stream);

/* enslam_sample_rays_g and enslam_step_prepare (without its conversion role) as ONE launch, for render calls whose
 * feature grids all arrive in the kernels' own layout: with nothing to convert, no prepare role waits for the sampler's
 * block marks, and the decoder packing / accumulator clearing run under the sampler's latency (a ray's wave is one dependent
 * chain of float64 divisions and a 21-step shuffle sort).  Arguments: enslam_sample_rays_g's, then enslam_step_prepare's
 * (n_dec .. packed, n_zero .. n_flat).  n_rays = 0 runs the prepare roles alone.
 *   Replaces Renderer.render_batch_ray lines 83-171 (sampling) of the reference; the prepare roles replace nothing. */
int enslam_sample_prepare(int32_t n_rays, int32_t n_lin, int32_t n_surf, const float *rays_o, const float *rays_d,
                          const float *gt_depth, const double *bound_host, const float *t_lin, const double *t_surf,
                          int32_t lindisp, const float *t_rand, float *scratch, int32_t depth_max_given, double *z_vals,
                          int32_t mark_stage, const enslam_scene *mark_scene, uint8_t *const *mark_flags,
                          int32_t mark_block_voxels, uint8_t *const *mark_flags64, int32_t n_dec, const int32_t *kinds,
                          const enslam_mlp_params *params, float *const *packed, int32_t n_zero, float *const *zero_dst,
                          const int64_t *zero_voxels, const uint8_t *const *zero_need, float *flat, int64_t n_flat,
                          void *stream);

/* Head of the tracker's camera iteration in one launch (one workgroup; any n): for the n window pixels `pixel_index` (int64,
 * row-major inside the window that starts at (H0, W0) and is window_w wide -- the single torch.randint draw of
 * common.get_sample_uv) it writes their column / row as floats (pix_i, pix_j), their depth and colour samples (gt_depth,
 * gt_color as float32 [n,3]), their rays from the camera tensor [qr,qi,qj,qk,tx,ty,tz] (rays_o, rays_d: enslam_pose_rays_fwd's
 * arithmetic) and -- prefilter != 0 -- the in-bound mask  inside[k] = min_axis max_side((bound - o) / d) >= gt_depth[k]  in
 * float64, with depth_max = {max over the inside rays of gt_depth (0 if none), that * 1.2f}: the sampler's batch maxima when the
 * rays the reference drops are rendered but masked (prefilter == 0: maxima over all rays, inside untouched).
 * draw_counter (optional, int32 [1] on the device): pixel_index then holds n_draws rows of n indices drawn ahead (one
 * torch.randint per frame instead of one per iteration -- under hipGraph replay every captured randint costs its launch plus two
 * fill launches for the generator's seed and offset); the call takes row *draw_counter % n_draws and increments the counter.
 * An index outside [0, (image_h - H0) * window_w) is clamped into it (no read ever leaves the images).
 *   Replaces Tracker.optimize_cam_in_batch lines 160-174 (get_samples with the camera tensor's pose, the `inside_mask` filter)
 *   and common.get_camera_from_tensor (common.py:189-229) for this use. */
int enslam_tracker_rays(int32_t n, const float *camera_tensor, const int64_t *pixel_index, int32_t H0, int32_t W0,
                        int32_t window_w, int32_t image_w, int32_t image_h, const float *depth_image, const void *color_image,
                        int32_t color_is_f64, float fx, float fy, float cx, float cy, const double *bound_host,
                        int32_t prefilter, float *pix_i, float *pix_j, float *rays_o, float *rays_d, float *gt_depth,
                        float *gt_color, uint8_t *inside, float *depth_max, int32_t *draw_counter, int32_t n_draws,
                        void *stream);

/* enslam_render_loss_fwd with the TRACKER's loss in place of the mapper's (batches of up to enslam_tracker_tail_max_rays()
 * rays; more: ENSLAM_EUNSUPPORTED): behind the decoder kernel two launches of 16 rays per workgroup (one without
 * handle_dynamic) -- compositing with
 *   tmp = |gt_depth - depth| / sqrt(var + 1e-10),
 * then, every workgroup taking the median of tmp over the inside rays for itself (the lower median of torch.median; by counting
 * ranks in LDS up to 1024 rays, a bitonic network above),
 *   keep = inside & (handle_dynamic ? tmp < 10 * median : 1),
 *   loss += sum_{keep & gt_depth > 0} tmp + w_color * sum_{keep & gt_depth > 0} |gt_color - color|   (gt_color NULL: depth term)
 * with d(loss)/d(raw) for a unit loss gradient in d_raw_unit and the work list of active tiles, the variance treated as a
 * constant.  inside NULL: every ray.  tmp_scratch: float64 [n_rays].  *loss must be 0 on entry.
 *   Replaces Tracker.optimize_cam_in_batch lines 176-195 around Renderer.render_batch_ray. */
int enslam_tracker_tail_max_rays(void);
int enslam_render_tracker_loss_fwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                                   const double *z_vals, const enslam_scene *scene, double *depth, double *var, float *rgb,
                                   float *raw_out, float *act_ws, int32_t act_light, const float *gt_depth,
                                   const float *gt_color, float w_color, const uint8_t *inside, int32_t handle_dynamic,
                                   double *tmp_scratch, double *loss, float *d_raw_unit, int32_t *work_list,
                                   int32_t *work_count, void *stream);

/* Sample distances along rays (mark_scene / mark_flags non-NULL: also does enslam_mark_blocks' work for stage mark_stage
 * on the samples it has just placed -- one launch less per render call).
 *   Replaces Renderer.render_batch_ray lines 83-171
 * (src/utils/Renderer.py): near/far from gt_depth and the AABB exit, n_lin linear samples,
 * n_surf near-surface samples (gt_depth>0: [0.95d,1.05d]; else [0.001,max d]), ascending merge.
 *   t_lin   float32 [n_lin]  = torch.linspace(0,1,n_lin)
 *   t_surf  float64 [n_surf] = torch.linspace(0,1,n_surf).double()   (ignored if n_surf == 0)
 *   t_rand  float32 [N,n_lin] or NULL (perturb == 0)
 *   scratch float32 [2]: {max(gt_depth), fl32(max(gt_depth)*1.2f)} (batch-global, Renderer.py:110,145).
 *           depth_max_given == 0: computed here over this call's gt_depth and written to scratch;
 *           depth_max_given != 0: read from scratch (a ray-sharded caller supplies the max of the WHOLE batch,
 *           so that every shard samples exactly what the unsharded reference would).
 *   z_vals  float64 [N, n_lin + n_surf] (n_surf forced to 0 when gt_depth is NULL)
 * Bit-exact with the reference's float64/float32 evaluation order. */
int enslam_sample_rays(int32_t n_rays, int32_t n_lin, int32_t n_surf, const float *rays_o, const float *rays_d,
                       const float *gt_depth, const double *bound_host, const float *t_lin, const double *t_surf,
                       int32_t lindisp, const float *t_rand, float *scratch, int32_t depth_max_given, double *z_vals,
                       int32_t mark_stage, const enslam_scene *mark_scene, uint8_t *const *mark_flags, void *stream);
/* enslam_sample_rays whose block marking works at mark_block_voxels in {64, 32, 16, 8} voxels per flag (mark_flags sized
 * accordingly) and, for the finer ones, optionally leaves the 64-voxel form in mark_flags64 as well (NULL: not wanted): the flags
 * of a ray-sharded step's gradient bucket (enslam_bucket_pack_g) and what the finish launch keeps, from ONE launch. */
int enslam_sample_rays_g(int32_t n_rays, int32_t n_lin, int32_t n_surf, const float *rays_o, const float *rays_d,
                         const float *gt_depth, const double *bound, const float *t_lin, const double *t_surf,
                         int32_t lindisp, const float *t_rand, float *scratch, int32_t depth_max_given, double *z_vals,
                         int32_t mark_stage, const enslam_scene *mark_scene, uint8_t *const *mark_flags,
                         int32_t mark_block_voxels, uint8_t *const *mark_flags64, void *stream);

/* Forward of Renderer.render_batch_ray lines 173-181 + eval_points (Renderer.py:24-62) +
 * NICE.forward (decoder.py:312-342) + raw2outputs_nerf_color (common.py:256-297, occupancy):
 * points, bound mask, trilinear gather, decoders, alpha compositing.  S must be 16, 32, 48 or 64 (64: tile-per-wave
 * forward only, i.e. up to 32768 rays per call; ENSLAM_EUNSUPPORTED beyond).
 * raw_out (may be NULL) receives the per-sample (r,g,b,occ) needed by enslam_render_bwd. */
int enslam_render_fwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                      const double *z_vals, const enslam_scene *scene, double *depth, double *var, float *rgb,
                      float *raw_out, float *act_ws, int32_t act_light, void *stream);

/* float32 count of the activation workspace `act_ws` for a batch (0 for the coarse stage, which always
 * recomputes).  When enslam_render_fwd is given a workspace of this size it also writes, per 16-sample tile and
 * decoder, the operands the backward needs (embedding, grid features, hidden activations, ReLU masks), and
 * enslam_decoder_bwd / enslam_render_bwd given the same buffer read them instead of recomputing the decoder
 * forward.  NULL in both places selects recomputation (no extra memory, slower backward).
 * act_light != 0 (same value in the size query, the forward and the backward): the backward will be asked for no
 * decoder-parameter gradient (tracker iterations, mapper stages with fixed decoders), so only the sample
 * coordinates, the ReLU masks and the trilinear cell records are kept: 1.8 KB instead of 22 KB per tile and decoder. */
size_t enslam_activation_floats(int32_t stage, int32_t n_rays, int32_t n_samples, int32_t act_light);
/* float32 count of the scratch buffer `dgrid_ws` the backward needs when it runs from `act_ws` AND ray gradients
 * are requested (decoder kernel -> ray-gradient kernel hand-off: feature gradient and embedding position gradient per
 * tile and decoder).  dgrid_ws may be NULL when g_rays_o is NULL.  act_ws needs every grid below 2^29 voxels. */
size_t enslam_grid_handoff_floats(int32_t stage, int32_t n_rays, int32_t n_samples);

/* Forward of Renderer.eval_points (Renderer.py:24-62) on explicit points p float64 [P,3].
 * apply_mask = 0 gives the bare decoder call NICE.forward (decoder.py:312-342, as Mesher.py:308 uses it). */
int enslam_eval_points(int32_t stage, int64_t n_points, const double *points, const enslam_scene *scene,
                       int32_t apply_mask, float *raw_out, void *stream);

/* Hand-derived backward of enslam_render_fwd (replaces autograd of the reference ops).
 *   g_depth float64 [N], g_var float64 [N] or NULL, g_rgb float32 [N,3] or NULL
 *   grad_grids[k].data : voxel-major accumulators (caller-zeroed) or NULL to skip that grid
 *   grad_packed[k]     : packed-layout accumulators (caller-zeroed, enslam_packed_grad_floats) or NULL
 *   g_rays_o/g_rays_d  : float32 [N,3] accumulators (caller-zeroed) or NULL
 *   d_raw              : float32 [N*S,4] workspace
 * Gradients follow the reference's autograd exactly: none through the out-of-bound occupancy
 * overwrite (Renderer.py:58), none to grid_middle / positions through the fine decoder's
 * concatenated middle feature (decoder.py:184-186), none to z_vals. */
int enslam_render_bwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                      const double *z_vals, const enslam_scene *scene, const float *raw, const double *depth,
                      const double *g_depth, const double *g_var, const float *g_rgb,
                      const enslam_grid *grad_grids, float *const *grad_packed, float *g_rays_o, float *g_rays_d,
                      float *d_raw, const float *act_ws, int32_t act_light, float *dgrid_ws, void *stream);

/* The mapper's RGB-D loss (Mapper.py:553-562:  sum_{gt_depth>0} |gt_depth - depth| + w_color * sum |gt_color - color|)
 * fused into the compositing launches of the render (two dependent launches fewer per optimisation step).
 * enslam_render_loss_fwd   : enslam_render_fwd that also ADDS the loss of this batch into loss[0] (float64, zero on
 *   entry; one atomic per 16 rays, so the last bits depend on the order).  gt_color NULL: depth term only.  Needs raw_out;
 *   ray counts above the tile-mode limit (full-image renders) return ENSLAM_EUNSUPPORTED.  d_raw_unit (optional,
 *   float32 [N*S,4]): d(loss)/d(raw) for g_loss = 1, so that the backward can start at enslam_decoder_bwd_scaled
 *   (d_raw = d_raw_unit, d_raw_scale = g_loss) without a compositing launch.
 * enslam_composite_loss_bwd: enslam_composite_bwd with d(depth), d(rgb) derived from that loss and the device scalar
 *   g_loss = d(total)/d(loss) instead of read from memory; follow with enslam_decoder_bwd. */
int enslam_render_loss_fwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                           const double *z_vals, const enslam_scene *scene, double *depth, double *var, float *rgb,
                           float *raw_out, float *act_ws, int32_t act_light, const float *gt_depth,
                           const float *gt_color, float w_color, double *loss, float *d_raw_unit, int32_t *work_list,
                           int32_t *work_count, void *stream);
int enslam_composite_loss_bwd(int32_t n_rays, int32_t n_samples, const float *raw, const double *z_vals,
                              const double *depth, const float *rgb, const float *gt_depth, const float *gt_color,
                              float w_color, const double *g_loss, float *d_raw, int32_t *work_list, int32_t *work_count,
                              void *stream);

/* The two halves of enslam_render_bwd, callable on their own.
 * enslam_composite_bwd: backward of raw2outputs_nerf_color (common.py:284-296): d(depth,var,rgb) -> d_raw [N*S,4].
 * enslam_decoder_bwd  : everything upstream of raw (decoders, gather, points), consuming d_raw.  With act_ws and
 *   (from the size query above) it reads the saved activations and cell records and leaves what the ray gradients
 *   need in dgrid_ws: follow it with enslam_ray_grad_bwd (enslam_render_bwd does).  With act_ws given and dgrid_ws NULL
 *   the kernel computes the ray gradients itself at the end of each round (costs the kernel ~10 %; worth it when nothing
 *   else would need a launch after it, e.g. tracker iterations on a fixed map).  With act_ws NULL everything, ray
 *   gradients included, is recomputed in the one kernel.  In both of these cases enslam_ray_grad_bwd must not be called.
 * enslam_ray_grad_bwd : ray gradients of the saved-activation path (fp64 sample geometry, corner re-gather,
 *   coordinate gradient, per-ray reduction) from dgrid_ws, added into g_rays_o / g_rays_d. */
int enslam_composite_bwd(int32_t n_rays, int32_t n_samples, const float *raw, const double *z_vals,
                         const double *depth, const double *g_depth, const double *g_var, const float *g_rgb,
                         float *d_raw, void *stream);
int enslam_decoder_bwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                       const double *z_vals, const enslam_scene *scene, const float *d_raw, const float *act_ws,
                       int32_t act_light, float *dgrid_ws, const enslam_grid *grad_grids, float *const *grad_packed,
                       float *g_rays_o, float *g_rays_d, void *stream);
/* enslam_decoder_bwd with every d_raw value multiplied by the device scalar *d_raw_scale (NULL: 1). */
int enslam_decoder_bwd_scaled(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                              const double *z_vals, const enslam_scene *scene, const float *d_raw,
                              const double *d_raw_scale, const float *act_ws, int32_t act_light, float *dgrid_ws,
                              const enslam_grid *grad_grids, float *const *grad_packed, float *g_rays_o,
                              float *g_rays_d, const int32_t *work_list, const int32_t *work_count, void *stream);
/* enslam_decoder_bwd_scaled whose decoder-parameter gradients leave the kernel as per-workgroup PARTIAL IMAGES instead of
 * float atomics into grad_packed: grad_partial[k] (float32 [enslam_bwd_partial_floats(k)], need not be cleared; NULL: atomics as
 * before) receives a 16-float header (word 0, int32: number of rows) and one row of enslam_packed_grad_floats(k) floats per
 * workgroup of decoder k's role; enslam_step_finish_partials sums the rows while it unpacks.  grad_packed[k] must still be
 * given (it selects which decoders get parameter gradients) but is not written for those decoders.  All workgroups of the
 * persistent backward finish within microseconds of each other; their 17.6 MB of atomics onto the same 53 k addresses ran at
 * the chip-wide float-atomic rate (10-13 us at the tail of a 137 us kernel).  Replaces nothing in the reference (autograd
 * sums the weight gradients inside its mm kernels). */
size_t enslam_bwd_partial_floats(int kind);
int enslam_decoder_bwd_partials(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                                const double *z_vals, const enslam_scene *scene, const float *d_raw,
                                const double *d_raw_scale, const float *act_ws, int32_t act_light, float *dgrid_ws,
                                const enslam_grid *grad_grids, float *const *grad_packed, float *const *grad_partial,
                                float *g_rays_o, float *g_rays_d, const int32_t *work_list, const int32_t *work_count,
                                void *stream);
/* Work list (optional everywhere, NULL/NULL = every tile): the 16-sample tiles (ray * n_samples/16 + tile) whose d_raw is
 * not all zero -- behind a converged surface the transmittance underflows to 0 and the far tiles of most rays carry no
 * gradient.  The kernel that produces d_raw appends them ray by ray (work_list int32 [n_rays * n_samples/16], work_count
 * int32 [1], ZERO on entry): enslam_render_loss_fwd (with d_raw_unit), enslam_composite_loss_bwd,
 * enslam_composite_bwd_list.  enslam_decoder_bwd_scaled (saved-activation roles only) and enslam_step_finish_rays then
 * walk the list instead of all tiles; both must be given the same list. */
int enslam_composite_bwd_list(int32_t n_rays, int32_t n_samples, const float *raw, const double *z_vals,
                              const double *depth, const double *g_depth, const double *g_var, const float *g_rgb,
                              float *d_raw, int32_t *work_list, int32_t *work_count, void *stream);
int enslam_ray_grad_bwd(int32_t stage, int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                        const double *z_vals, const enslam_scene *scene, float *dgrid_ws, float *g_rays_o,
                        float *g_rays_d, void *stream);

/* raw2outputs_nerf_color (common.py:256-297, occupancy=True) on its own: raw float32 [N,S,4], z_vals float64 [N,S]
 * -> depth/var float64 [N], rgb float32 [N,3], weights float32 [N,S] (may be NULL).  1 <= S <= 64. */
int enslam_composite_fwd(int32_t n_rays, int32_t n_samples, const float *raw, const double *z_vals, double *depth,
                         double *var, float *rgb, float *weights, void *stream);

/* Mapper's RGB-D loss, Mapper.py:553-562:  sum_{gt_depth>0} |gt_depth - depth| + w_color * sum |gt_color - color|
 * (colour term only when color/gt_color are given, i.e. in the colour stage).  One kernel each way, deterministic
 * single-block reduction (n <= ray_batch_size).  loss: float64 [1]; g_loss: float64 [1] upstream gradient (device);
 * g_depth float64 [n], g_color float32 [n,3] (NULL when there is no colour term). */
int enslam_rgbd_loss_fwd(int32_t n, const double *depth, const float *color, const float *gt_depth,
                         const float *gt_color, float w_color, double *loss, void *stream);
int enslam_rgbd_loss_bwd(int32_t n, const double *depth, const float *color, const float *gt_depth,
                         const float *gt_color, float w_color, const double *g_loss, double *g_depth,
                         float *g_color, void *stream);

/* Parity helper: base voxel index and fractions the gather uses for points p float64 [P,3]
 * (normalize_3d_coordinate, common.py:342-357, then ATen grid_sampler unnormalize/clip/floor). */
int enslam_voxel_index(int64_t n_points, const double *points, const double *bound_host, int32_t D, int32_t H,
                       int32_t W, int32_t *ix, int32_t *iy, int32_t *iz, float *fx, float *fy, float *fz,
                       void *stream);

/* Parity helper: points p = o + d*z (float64 [N*S,3]) and the strict in-bound mask (uint8 [N*S]). */
int enslam_ray_points(int32_t n_rays, int32_t n_samples, const float *rays_o, const float *rays_d,
                      const double *z_vals, const double *bound_host, double *points, uint8_t *mask, void *stream);

/* Parity helper: the device sin / cos the Fourier embedding and its backward use (decoder.py:26-30: torch.sin of
 * p @ B in float32) on n float32 arguments; either output may be NULL. */
int enslam_fourier_sincos(int64_t n, const float *x, float *sin_out, float *cos_out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Step plans: one differentiable render call (Renderer.render_batch_ray, or render_batch_ray_rgbd_loss, and its backward)
 * as TWO calls of this library instead of the six to eight entry points above, with every buffer of the call a fixed
 * offset inside three caller-allocated blobs.  A Python host pays ~200 us per direction for marshalling ~180 pointers,
 * allocating ~25 tensors and laying out accumulators; with a plan it checks a cache key, allocates three blobs and makes one
 * call.  Replaces nothing in the reference (host glue); the launches are exactly those of the entry points above.
 *
 * enslam_step_plan describes what does not change between the steps of a loop (filled by the host once per combination of
 * stage / batch size / grid set / gradient pattern); enslam_step_layout is derived from it (enslam_plan_layout).
 *   grid_mode[k]  0: kind k unused by the stage
 *                 1: values given voxel-major ([V][32]: a channels_last_3d tensor or a cached copy), no gradient
 *                 2: values voxel-major, gradient wanted: added into the gradient blob's [V][32] region of kind k (zero-filled
 *                    by the forward), which the host hands to autograd as a channels_last_3d tensor
 *                 3: values channel-major ([32][V], the reference's strides), gradient wanted: touched 64-voxel blocks
 *                    converted per step (scratch blob), gradient transposed back into a dense [32][V] region of the gradient blob
 *   par_grad[k]   decoder k's parameters get gradients (all or none); they appear in the gradient blob's parameter region at the
 *                 float offsets pgrad_off[k][..] in the order W0,b0,..,W4,b4,Wc0,bc0,..,Wc4,bc4,Wo,bo,B
 *   loss_kind     0: outputs depth / var / rgb, backward from their gradients; 1: the mapper's RGB-D loss fused into the
 *                 compositing launches (enslam_render_loss_fwd), backward from d(total)/d(loss)
 * Blobs (device memory, 256-byte aligned, sizes from the layout): scratch (lives from forward to backward), grad (lives as long
 * as any gradient the host made from it), out (depth | var | rgb | loss).  Nothing needs clearing by the caller. */
typedef struct enslam_step_plan {
    int32_t stage, n_rays, n_lin, n_surf, lindisp, act_light, need_rays, use_work_list, loss_kind, use_color;
    float w_color;
    int32_t grid_mode[4], par_grad[4];
    int32_t grid_D[4], grid_H[4], grid_W[4];
    double bound[6], coarse_bound[6];
    enslam_mlp_params params[4];       /* the callers' parameter tensors (sources of the per-step packing) */
    int64_t pgrad_off[4][23];          /* float offsets inside the gradient blob's parameter region */
    int64_t pgrad_floats;              /* size of that region */
    const float *t_lin;                /* device: [n_lin] */
    const double *t_surf;              /* device: [n_surf] */
} enslam_step_plan;

typedef struct enslam_step_layout {    /* byte offsets (-1: absent) and sizes */
    int64_t scratch_bytes, grad_bytes, out_bytes;
    int64_t s_zero_bytes;              /* scratch [0, s_zero_bytes) is cleared by the forward: block flags, packed decoders */
    int64_t s_flags[4], s_packed[4], s_z, s_dmax, s_raw, s_act, s_work, s_draw, s_dgw, s_vm[4], s_gacc[4];
    int64_t g_flat, g_flat_floats;     /* grad blob: range the forward's prepare roles clear */
    int64_t g_packed[4], g_ro, g_rd, g_counter, g_nat[4], g_params, g_dense[4];
    int64_t o_depth, o_var, o_rgb, o_loss;
    int32_t n_samples, finish_needed, inline_rays, merged;
} enslam_step_layout;

int64_t enslam_plan_struct_bytes(int32_t which);       /* sizeof(enslam_step_plan) (0) / sizeof(enslam_step_layout) (1): binding self-check */
int enslam_plan_layout(const enslam_step_plan *plan, enslam_step_layout *layout);
/* grid_values[k]: the grid's values (voxel-major for modes 1 / 2, channel-major for mode 3).  gt_color: loss_kind 1 with colour term.
 * depth_max: device float32 [2] {max, max * 1.2f} of the whole batch or NULL (taken over this call's rays). */
int enslam_plan_forward(const enslam_step_plan *plan, const enslam_step_layout *layout, void *scratch, void *grad, void *out,
                        const float *rays_o, const float *rays_d, const float *gt_depth, const float *gt_color,
                        const float *depth_max, const float *const *grid_values, void *stream);
/* g_depth / g_var (float64 [N]) / g_rgb (float32 [N,3]): gradients of the outputs, any may be NULL (loss_kind 0);
 * g_loss: device float64 [1] (loss_kind 1). */
int enslam_plan_backward(const enslam_step_plan *plan, const enslam_step_layout *layout, void *scratch, void *grad, void *out,
                         const float *rays_o, const float *rays_d, const float *gt_depth, const float *gt_color,
                         const float *const *grid_values, const double *g_depth, const double *g_var, const float *g_rgb,
                         const double *g_loss, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ENSLAM_HIP_H */
