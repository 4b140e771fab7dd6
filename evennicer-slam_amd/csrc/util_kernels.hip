// Small kernels around the fused render kernels: depth-guided sampling (bit-exact float64 order),
// grid layout conversion, decoder re-layout, parity helpers.
#include "kernels.hpp"

namespace {

// ------------------------------------------------------------------ batch max of gt_depth
// Renderer.py:110,145 need max(gt_depth) over the whole batch (max(gd*1.2) == fl32(max(gd)*1.2f):
// rounding is monotone).  One block; N <= ray_batch_size (100000).
__global__ __launch_bounds__(1024) void depth_max_kernel(int n, const float* __restrict__ gd, float* __restrict__ out) {
    __shared__ float red[16];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, gd[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 16) {
        m = red[threadIdx.x];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (threadIdx.x == 0) { out[0] = m; out[1] = m * 1.2f; }
    }
}

// ------------------------------------------------------------------ z sampling, one wave per ray
// Lane k < n_lin holds linear sample k, lanes n_lin .. n_lin+n_surf-1 the near-surface samples, the rest +inf;
// a 64-lane bitonic network (shuffles) sorts them ascending like torch.sort (values only).
// Follows the dtype promotion of Renderer.py:95-171 term by term:
//   far_bb  float64: min_axis(max_side((bound - o)/d)) + 0.01
//   near    float32: gt_depth*0.01f  (0.01f when no depth)
//   z_lin   float64: (double)(near * (1.f - t)) + far * (double)t
//   surface float64: (double)(0.95f*d) * (1 - ts) + (double)(1.05f*d) * ts   | 0.001*(1-ts) + (double)dmax*ts
constexpr int MAX_S = 64;
ENS_DEV void sample_body(const SampleArgs& A, const MarkArgs& mk, const int ray, const int lane) {
    const int n_rays = A.n_rays, n_lin = A.n_lin, n_surf = A.n_surf, lindisp = A.lindisp, dmax_inline = A.dmax_inline;
    const float* __restrict__ ro = A.ro;
    const float* __restrict__ rd = A.rd;
    const float* __restrict__ gd = A.gd;
    const float* __restrict__ t_lin = A.t_lin;
    const double* __restrict__ t_surf = A.t_surf;
    const float* __restrict__ t_rand = A.t_rand;
    const float* __restrict__ dmax = A.dmax;
    double* __restrict__ zout = A.zout;
    const double lo0 = A.lo[0], hi0 = A.hi[0], lo1 = A.lo[1], hi1 = A.hi[1], lo2 = A.lo[2], hi2 = A.hi[2];
    float dmax0 = 0.f, dmax1 = 0.f;                     // max(gt_depth) over the batch and fl32(max * 1.2f)
    if (gd != nullptr) {
        if (dmax_inline) {                              // small batch: every wave reduces the (L2-resident) depths itself
            float m = -INFINITY;
            for (int i = lane; i < n_rays; i += 64) m = fmaxf(m, gd[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            dmax0 = m; dmax1 = m * 1.2f;
        } else {
            dmax0 = dmax[0]; dmax1 = dmax[1];
        }
    }
    double far_bb = INFINITY;
    {   // the six face distances are ray-uniform: lanes 0..5 divide once each (float64 divisions are the long pole of
        // this kernel), the others pick the results up
        const int ax = lane < 6 ? (lane >> 1) : 0;
        const double face = (lane & 1) ? (ax == 0 ? hi0 : (ax == 1 ? hi1 : hi2)) : (ax == 0 ? lo0 : (ax == 1 ? lo1 : lo2));
        const double tq = (face - (double)ro[ray * 3 + ax]) / (double)rd[ray * 3 + ax];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double t0 = __shfl(tq, 2 * a), t1 = __shfl(tq, 2 * a + 1);
            const double tm = t0 > t1 ? t0 : t1;        // torch.max over the two faces
            far_bb = tm < far_bb ? tm : far_bb;        // torch.min over axes
        }
    }
    far_bb += 0.01;
    const bool guided = gd != nullptr;
    const float g = guided ? gd[ray] : 0.f;
    float near32 = 0.01f;
    double far = far_bb;
    if (guided) {
        near32 = g * 0.01f;
        const double cap = (double)dmax1;               // fl32(max(gd)*1.2f)
        far = far_bb < 0.0 ? 0.0 : far_bb;              // clamp(min=0, max=cap)
        far = far > cap ? cap : far;
    }
    const int S = n_lin + ((guided && n_surf > 0) ? n_surf : 0);
    double z = INFINITY;
    if (lane < n_lin) {
        const float t = t_lin[lane];
        const float omt = 1.f - t;
        if (!lindisp) {
            z = (double)(near32 * omt) + far * (double)t;
        } else {
            const float inv_near = guided ? 1.f / near32 : 100.0f;
            z = 1.0 / ((double)(inv_near * omt) + (1.0 / far) * (double)t);
        }
    }
    if (t_rand != nullptr) {                             // Renderer.py:160-167 (linear samples only)
        const double zn = __shfl_down(z, 1), zp = __shfl_up(z, 1);
        const double upper = lane + 1 < n_lin ? 0.5 * (zn + z) : z;
        const double lower = lane > 0 ? 0.5 * (z + zp) : z;
        if (lane < n_lin) z = lower + (upper - lower) * (double)t_rand[(int64_t)ray * n_lin + lane];
    }
    if (S > n_lin) {
        if (lane >= n_lin && lane < S) {
            const float a32 = 0.95f * g, b32 = 1.05f * g;
            const double ts = t_surf[lane - n_lin];
            z = g > 0.f ? (double)a32 * (1.0 - ts) + (double)b32 * ts : 0.001 * (1.0 - ts) + (double)dmax0 * ts;
        }
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {              // bitonic sort, ascending over the 64 lanes
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const double other = __shfl_xor(z, j);
                const bool asc = (lane & k) == 0, low = (lane & j) == 0;
                const double mn = other < z ? other : z, mx = other < z ? z : other;
                z = (low == asc) ? mn : mx;
            }
        }
    }
    if (lane < S) {
        zout[(int64_t)ray * S + lane] = z;
        if (mk.kmask) {                                  // flag the 64-voxel blocks this sample's 8 corners sit in
            double pw[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) pw[a] = (double)ro[ray * 3 + a] + (double)rd[ray * 3 + a] * z;
            float pn[3] = {0.f, 0.f, 0.f};               // normalised coordinates over Renderer.bound, shared by grids 1..3
            if (mk.kmask & 14) {
#pragma unroll
                for (int a = 0; a < 3; ++a) pn[a] = axis_norm(pw[a], mk.sc.lo[a], mk.sc.hi[a]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!((mk.kmask >> k) & 1) || mk.flags[k] == nullptr) continue;
                const DevGrid& g = mk.sc.grid[k];
                int ix, iy, iz;
                if (k == 0) {
                    ix = axis_cell(axis_norm(pw[0], mk.sc.clo[0], mk.sc.chi[0]), g.W);
                    iy = axis_cell(axis_norm(pw[1], mk.sc.clo[1], mk.sc.chi[1]), g.H);
                    iz = axis_cell(axis_norm(pw[2], mk.sc.clo[2], mk.sc.chi[2]), g.D);
                } else {
                    ix = axis_cell(pn[0], g.W); iy = axis_cell(pn[1], g.H); iz = axis_cell(pn[2], g.D);
                }
                const int dx = (ix + 1 < g.W) ? 1 : 0;         // the 8 corners, clamped like corner(): four x-pairs of
#pragma unroll                                               // adjacent linear indices -> mostly ONE block per pair
                for (int c = 0; c < 4; ++c) {
                    const int y = min(iy + (c & 1), g.H - 1), zc = min(iz + (c >> 1), g.D - 1);
                    const int64_t idx = ((int64_t)zc * g.H + y) * g.W + ix;
                    uint8_t* f = mk.flags[k];
                    const int sh = mk.shift;
                    f[idx >> sh] = 1;
                    if (((idx + dx) >> sh) != (idx >> sh)) f[(idx + dx) >> sh] = 1;
                    if (mk.flags64[k] != nullptr) {              // (finer flags for the gradient bucket: the 64-voxel form too)
                        uint8_t* f6 = mk.flags64[k];
                        f6[idx >> 6] = 1;
                        if (((idx + dx) >> 6) != (idx >> 6)) f6[(idx + dx) >> 6] = 1;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(64) void sample_kernel(SampleArgs A, MarkArgs mk) { sample_body(A, mk, (int)blockIdx.x, (int)threadIdx.x); }

// ------------------------------------------------------------------ mapper RGB-D loss (Mapper.py:553-562)
__global__ __launch_bounds__(1024) void rgbd_loss_fwd_kernel(int n, const double* __restrict__ depth,
                                                             const float* __restrict__ color,
                                                             const float* __restrict__ gd, const float* __restrict__ gc,
                                                             float w, double* __restrict__ loss) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float g = gd[i];
        if (g > 0.f) acc += fabs((double)g - depth[i]);
        if (color != nullptr) {
            float c = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) c += fabsf(gc[i * 3 + a] - color[i * 3 + a]);
            acc += (double)(w * c);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        acc = red[threadIdx.x];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (threadIdx.x == 0) loss[0] = acc;
    }
}
ENS_DEV float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// ------------------------------------------------------------------ tracker glue (Tracker.py:141-197)
// Tracker's RGB-D loss (:179-195, handle_dynamic off):  sum_{gd>0} |gd - d| / sqrt(u + 1e-10)  +  w * sum_{gd>0} |gc - c|
__global__ __launch_bounds__(1024) void tracker_loss_fwd_kernel(int n, const double* __restrict__ depth,
                                                                const double* __restrict__ unc,
                                                                const float* __restrict__ color,
                                                                const float* __restrict__ gd, const float* __restrict__ gc,
                                                                float w, double* __restrict__ loss) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float g = gd[i];
        if (!(g > 0.f)) continue;
        acc += fabs((double)g - depth[i]) / sqrt(unc[i] + 1e-10);
        if (color != nullptr) {
            float c = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) c += fabsf(gc[i * 3 + a] - color[i * 3 + a]);
            acc += (double)w * (double)c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        acc = red[threadIdx.x];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (threadIdx.x == 0) loss[0] = acc;
    }
}
__global__ __launch_bounds__(256) void tracker_loss_bwd_kernel(int n, const double* __restrict__ depth,
                                                               const double* __restrict__ unc,
                                                               const float* __restrict__ color,
                                                               const float* __restrict__ gd, const float* __restrict__ gc,
                                                               float w, const double* __restrict__ g_loss,
                                                               double* __restrict__ g_depth, float* __restrict__ g_color) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double g = g_loss[0];
    const float t = gd[i];
    const bool on = t > 0.f;
    const double diff = (double)t - depth[i];
    g_depth[i] = on ? (diff > 0.0 ? -g : (diff < 0.0 ? g : 0.0)) / sqrt(unc[i] + 1e-10) : 0.0;
    if (color != nullptr && g_color != nullptr) {
        const float gw = (float)g * w;
#pragma unroll
        for (int a = 0; a < 3; ++a) g_color[i * 3 + a] = on ? -gw * sgnf(gc[i * 3 + a] - color[i * 3 + a]) : 0.f;
    }
}

// camera tensor (unnormalised quaternion qr,qi,qj,qk + translation) -> rays through pixels (i, j):
// common.py:189-229 (quad2rotation, get_camera_from_tensor) + :74-89 (get_rays_from_uv) in one launch.
struct PoseR { float r[3][3]; float s; };
ENS_DEV PoseR pose_rotation(const float* __restrict__ ct) {
    const float qr = ct[0], qi = ct[1], qj = ct[2], qk = ct[3];
    PoseR o;
    o.s = 2.0f / (((qr * qr + qi * qi) + qj * qj) + qk * qk);
    const float s = o.s;
    o.r[0][0] = 1.f - s * (qj * qj + qk * qk); o.r[0][1] = s * (qi * qj - qk * qr); o.r[0][2] = s * (qi * qk + qj * qr);
    o.r[1][0] = s * (qi * qj + qk * qr); o.r[1][1] = 1.f - s * (qi * qi + qk * qk); o.r[1][2] = s * (qj * qk - qi * qr);
    o.r[2][0] = s * (qi * qk - qj * qr); o.r[2][1] = s * (qj * qk + qi * qr); o.r[2][2] = 1.f - s * (qi * qi + qj * qj);
    return o;
}
__global__ __launch_bounds__(256) void pose_rays_fwd_kernel(int n, const float* __restrict__ ct,
                                                            const float* __restrict__ pi, const float* __restrict__ pj,
                                                            float fx, float fy, float cx, float cy,
                                                            float* __restrict__ ro, float* __restrict__ rd) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const PoseR R = pose_rotation(ct);
    const float d0 = (pi[k] - cx) / fx, d1 = -(pj[k] - cy) / fy, d2 = -1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        rd[k * 3 + a] = (d0 * R.r[a][0] + d1 * R.r[a][1]) + d2 * R.r[a][2];
        ro[k * 3 + a] = ct[4 + a];
    }
}
// d loss / d camera tensor from the ray gradients: G[a][b] = sum_n g_rd[n][a] dir[n][b], g_T = sum_n g_ro[n], then the
// chain through R = I + s P(q), s = 2/|q|^2.  One workgroup, float64 accumulation, deterministic.
__global__ __launch_bounds__(1024) void pose_rays_bwd_kernel(int n, const float* __restrict__ ct,
                                                             const float* __restrict__ pi, const float* __restrict__ pj,
                                                             float fx, float fy, float cx, float cy,
                                                             const float* __restrict__ g_ro, const float* __restrict__ g_rd,
                                                             float* __restrict__ g_ct) {
    __shared__ double red[16][12];
    __shared__ double tot[12];
    double acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 0.0;
    const int nthr = (int)blockDim.x, nw = nthr >> 6;
    for (int k = threadIdx.x; k < n; k += nthr) {
        const double d[3] = {(double)((pi[k] - cx) / fx), (double)(-(pj[k] - cy) / fy), -1.0};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double g = g_rd ? (double)g_rd[k * 3 + a] : 0.0;
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[a * 3 + b] += g * d[b];
            acc[9 + a] += g_ro ? (double)g_ro[k * 3 + a] : 0.0;
        }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[e] += __shfl_xor(acc[e], o);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < 12; ++e) red[threadIdx.x >> 6][e] = acc[e];
    }
    __syncthreads();
    if (threadIdx.x < 12) {                              // (twelve lanes sum the waves' partials side by side)
        double v = 0.0;
        for (int w = 0; w < nw; ++w) v += red[w][threadIdx.x];
        tot[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double G[3][3], gT[3];
        for (int e = 0; e < 12; ++e) {
            if (e < 9) G[e / 3][e % 3] = tot[e]; else gT[e - 9] = tot[e];
        }
        const double qr = ct[0], qi = ct[1], qj = ct[2], qk = ct[3];
        const double nn = qr * qr + qi * qi + qj * qj + qk * qk, s = 2.0 / nn;
        const double P[3][3] = {{-(qj * qj + qk * qk), qi * qj - qk * qr, qi * qk + qj * qr},
                                {qi * qj + qk * qr, -(qi * qi + qk * qk), qj * qk - qi * qr},
                                {qi * qk - qj * qr, qj * qk + qi * qr, -(qi * qi + qj * qj)}};
        double dLds = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) dLds += G[a][b] * P[a][b];
        const double dr = -qk * G[0][1] + qj * G[0][2] + qk * G[1][0] - qi * G[1][2] - qj * G[2][0] + qi * G[2][1];
        const double di = qj * G[0][1] + qk * G[0][2] + qj * G[1][0] - 2 * qi * G[1][1] - qr * G[1][2] + qk * G[2][0] +
                          qr * G[2][1] - 2 * qi * G[2][2];
        const double dj = -2 * qj * G[0][0] + qi * G[0][1] + qr * G[0][2] + qi * G[1][0] + qk * G[1][2] - qr * G[2][0] +
                          qk * G[2][1] - 2 * qj * G[2][2];
        const double dk = -2 * qk * G[0][0] - qr * G[0][1] + qi * G[0][2] + qr * G[1][0] - 2 * qk * G[1][1] + qj * G[1][2] +
                          qi * G[2][0] + qj * G[2][1];
        const double ds = -s * s;                       // ds/dq_x = ds * q_x
        g_ct[0] = (float)(s * dr + dLds * ds * qr);
        g_ct[1] = (float)(s * di + dLds * ds * qi);
        g_ct[2] = (float)(s * dj + dLds * ds * qj);
        g_ct[3] = (float)(s * dk + dLds * ds * qk);
        g_ct[4] = (float)gT[0]; g_ct[5] = (float)gT[1]; g_ct[6] = (float)gT[2];
    }
}
__global__ __launch_bounds__(256) void rgbd_loss_bwd_kernel(int n, const double* __restrict__ depth,
                                                            const float* __restrict__ color,
                                                            const float* __restrict__ gd, const float* __restrict__ gc,
                                                            float w, const double* __restrict__ g_loss,
                                                            double* __restrict__ g_depth, float* __restrict__ g_color) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double g = g_loss[0];
    const float t = gd[i];
    const double diff = (double)t - depth[i];
    g_depth[i] = t > 0.f ? (diff > 0.0 ? -g : (diff < 0.0 ? g : 0.0)) : 0.0;      // d|gt-d|/dd = -sign(gt-d)
    if (color != nullptr && g_color != nullptr) {
        const float gw = (float)g * w;
#pragma unroll
        for (int a = 0; a < 3; ++a) g_color[i * 3 + a] = -gw * sgnf(gc[i * 3 + a] - color[i * 3 + a]);
    }
}

// ------------------------------------------------------------------ parity helpers
__global__ void ray_points_kernel(int n_rays, int S, const float* __restrict__ ro, const float* __restrict__ rd,
                                  const double* __restrict__ z, double lo0, double hi0, double lo1, double hi1,
                                  double lo2, double hi2, double* __restrict__ pts, uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_rays * S) return;
    const int ray = (int)(i / S);
    const double lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    bool in = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double pw = (double)ro[ray * 3 + a] + (double)rd[ray * 3 + a] * z[i];
        pts[i * 3 + a] = pw;
        in = in && (pw < hi[a]) && (pw > lo[a]);
    }
    mask[i] = in ? 1 : 0;
}

__global__ void voxel_index_kernel(int64_t n, const double* __restrict__ pts, double lo0, double hi0, double lo1,
                                   double hi1, double lo2, double hi2, DevGrid g, int* ix, int* iy, int* iz,
                                   float* fx, float* fy, float* fz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    const double pw[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
    const Vox v = make_vox(pw, lo, hi, g);
    ix[i] = v.ix; iy[i] = v.iy; iz[i] = v.iz;
    fx[i] = v.fx; fy[i] = v.fy; fz[i] = v.fz;
}

// the embedding's sin (forward: ens_sinf) and cos (backward: ens_cosf), exactly as the render kernels evaluate them
__global__ void sincos_kernel(int64_t n, const float* __restrict__ x, float* __restrict__ s, float* __restrict__ c) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (s != nullptr) s[i] = ens_sinf(x[i]);
    if (c != nullptr) c[i] = ens_cosf(x[i]);
}

// ------------------------------------------------------------------ [32][V] <-> [V][32]
// 64 voxels per block through a padded LDS tile; both sides move 256-byte rows.
ENS_DEV void to_vm_block(const float* __restrict__ src, float* __restrict__ dst, int64_t V, int64_t blk);
ENS_DEV void from_vm_block(const float* __restrict__ src, float* __restrict__ dst, int64_t V, int64_t blk);

ENS_DEV void convert_body(const ConvJob& job, int to_vm, int b) {
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) g = (i < job.n && b >= job.block_begin[i]) ? i : g;
    const int64_t blk = b - job.block_begin[g];
    const uint8_t* need = job.need[g];
    if (need != nullptr) {                                   // sparse path (block-uniform decisions)
        const bool needed = need[blk] != 0;
        if (to_vm) {
            uint8_t* valid = job.valid[g];
            if (!needed || (valid != nullptr && valid[blk] != 0)) return;
            to_vm_block(job.src[g], job.dst[g], job.V[g], blk);
            if (valid != nullptr && threadIdx.x == 0) valid[blk] = 1;
            return;
        }
        // Gradient direction with a PERSISTENT destination (job.valid = the flags of the launch that wrote this very buffer
        // last, or null): an untouched block that was untouched then as well still holds its zeros -- nothing to write.
        // Under hipGraph replay the dense gradient of a grid is the same memory every step, so the 48 MB zero-fill of the
        // untouched blocks shrinks to the blocks that were touched one step earlier; the flags move to job.valid on the way.
        uint8_t* prev = job.valid[g];
        const bool was = prev != nullptr && prev[blk] != 0;
        if (prev != nullptr) {
            if (!needed && !was) return;
            __syncthreads();                                 // (block-uniform) every thread has read prev[blk] and need[blk]
            if (threadIdx.x == 0) {
                if (was != needed) prev[blk] = needed ? 1 : 0;
                // the flag has served its step (marked by the sampler, read by the prepare launch and here): cleared, so
                // that a captured step can keep its flags in memory that no fill node re-zeroes; readers use `prev` from now on
                if (needed) const_cast<uint8_t*>(need)[blk] = 0;
            }
        }
        if (!needed) {                                       // untouched block of a gradient: zeros, nothing read
            const int64_t v0 = blk * 64, V = job.V[g];
            float* dst = job.dst[g];
            if ((V & 3) == 0 && v0 + 64 <= V && ((uintptr_t)dst & 15) == 0) {
                // 16 bytes per lane: a wave instruction clears four 256-byte row segments (two instructions per wave
                // and block instead of eight)
                const int row = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
#pragma unroll
                for (int cc = row; cc < 32; cc += 16)
                    *reinterpret_cast<f32x4*>(dst + (int64_t)cc * V + v0 + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
                return;
            }
            const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
            for (int cc = ty; cc < 32; cc += 4)
                if (v0 + tx < V) dst[(int64_t)cc * V + v0 + tx] = 0.f;
            return;
        }
    }
    if (to_vm) to_vm_block(job.src[g], job.dst[g], job.V[g], blk);
    else from_vm_block(job.src[g], job.dst[g], job.V[g], blk);
}
__global__ __launch_bounds__(256) void convert_kernel(ConvJob job, int to_vm) { convert_body(job, to_vm, (int)blockIdx.x); }

// zero the flagged 64-voxel blocks (8 KB each) of voxel-major buffers
ENS_DEV void zero_body(const ConvJob& job, float* __restrict__ flat, int64_t n_flat, int b) {
    if (b >= job.block_begin[job.n]) {                  // trailing blocks: a flat float range (2048 floats each)
        const int64_t e0 = ((int64_t)b - job.block_begin[job.n]) * 2048;
        for (int64_t e = e0 + threadIdx.x; e < e0 + 2048 && e < n_flat; e += 256) flat[e] = 0.f;
        return;
    }
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) g = (i < job.n && b >= job.block_begin[i]) ? i : g;
    const int64_t blk = b - job.block_begin[g];
    if (job.need[g] != nullptr && job.need[g][blk] == 0) return;
    const int64_t v0 = blk * 64, V = job.V[g];
    f32x4* dst = reinterpret_cast<f32x4*>(job.dst[g] + v0 * 32);
    const int64_t n4 = (V - v0 < 64 ? V - v0 : 64) * 8;
    for (int e = threadIdx.x; e < n4; e += 256) dst[e] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__global__ __launch_bounds__(256) void zero_blocks_kernel(ConvJob job, float* __restrict__ flat, int64_t n_flat) {
    zero_body(job, flat, n_flat, (int)blockIdx.x);
}

// ------------------------------------------------------------------ masked Adam on voxel-major grids
// torch.optim.Adam (defaults: no weight decay, no amsgrad) restricted to the voxels of the frustum mask, the way
// Mapper.optimize_map holds val_grad = val[mask] (Mapper.py:328-361) and steps it (:573-575):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// One workgroup per 64-voxel block (8 KB per array); a lane owns 4 channels of 8 voxels.  Gradients are cleared on
// the way (also where the mask is 0: the reference drops those), so the accumulators are all-zero again afterwards.
__global__ __launch_bounds__(256) void adam_masked_kernel(AdamJob job) {
    int gi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) gi = (i < job.n && (int)blockIdx.x >= job.block_begin[i]) ? i : gi;
    const int64_t blk = blockIdx.x - job.block_begin[gi];
    const int64_t V = job.V[gi], v0 = blk * 64;
    const int step = job.step[gi][0];
    const double lr = job.lr[gi][0];
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (step > 0) {                                      // the scalars torch.optim.Adam forms in Python float64
        const double bc1 = 1.0 - pow(job.beta1, (double)step);
        const double bc2 = 1.0 - pow(job.beta2, (double)step);
        step_size = (float)(lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    const float b1 = (float)job.beta1, b2 = (float)job.beta2, eps = (float)job.eps;
    const float omb1 = (float)(1.0 - job.beta1), omb2 = (float)(1.0 - job.beta2);
    f32x4* __restrict__ P = reinterpret_cast<f32x4*>(job.p[gi]);
    f32x4* __restrict__ G = reinterpret_cast<f32x4*>(job.g[gi]);
    f32x4* __restrict__ M = reinterpret_cast<f32x4*>(job.m[gi]);
    f32x4* __restrict__ W = reinterpret_cast<f32x4*>(job.v[gi]);
    const uint8_t* mask = job.mask[gi];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = threadIdx.x + 256 * j;                  // float4 index inside the block: voxel e>>3, channels 4*(e&7)..
        const int64_t vox = v0 + (e >> 3);
        if (vox >= V) continue;
        const int64_t o = vox * 8 + (e & 7);
        const bool on = step > 0 && (mask == nullptr || mask[vox] != 0);
        if (on) {
            const f32x4 g = G[o];
            f32x4 m = M[o], w = W[o], p = P[o];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                m[r] = m[r] * b1 + omb1 * g[r];
                w[r] = w[r] * b2 + omb2 * (g[r] * g[r]);
                const float denom = sqrtf(w[r]) / bc2_sqrt + eps;
                p[r] = p[r] - step_size * (m[r] / denom);
            }
            M[o] = m; W[o] = w; P[o] = p;
        }
        G[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// The same update for a list of small tensors (the decoder parameters the mapper optimises, Mapper.py:363-369):
// one launch instead of torch's ~50 foreach kernels per step.  1024 elements per workgroup.
__global__ __launch_bounds__(256) void adam_tensors_kernel(AdamTensorsJob job) {
    int lo = 0, hi = job.n;                              // tensor whose block range holds blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= job.block_begin[mid]) lo = mid; else hi = mid;
    }
    const int t = lo;
    const int step = job.step[0] + (job.self_inc ? 1 : 0);
    if (job.self_inc) {                                  // (one workgroup: every thread has read the old count before it is replaced)
        __syncthreads();
        if (threadIdx.x == 0) const_cast<int*>(job.step)[0] = step;
    }
    if (step <= 0) return;
    const double bc1 = 1.0 - pow(job.beta1, (double)step), bc2 = 1.0 - pow(job.beta2, (double)step);
    const float step_size = (float)(job.lr[0] / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float b1 = (float)job.beta1, b2 = (float)job.beta2, eps = (float)job.eps;
    const float omb1 = (float)(1.0 - job.beta1), omb2 = (float)(1.0 - job.beta2);
    float* __restrict__ P = job.p[t];
    const float* __restrict__ G = job.g[t];
    float* __restrict__ M = job.m[t];
    float* __restrict__ W = job.v[t];
    const int n = job.numel[t], e0 = ((int)blockIdx.x - job.block_begin[t]) * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = e0 + j * 256 + threadIdx.x;
        if (e >= n) continue;
        const float g = G[e];
        const float m = M[e] * b1 + omb1 * g;
        const float w = W[e] * b2 + omb2 * (g * g);
        M[e] = m; W[e] = w;
        P[e] = P[e] - step_size * (m / (sqrtf(w) / bc2_sqrt + eps));
    }
}

// flag the blocks holding the 8 (clamped) corners of every sample in every grid the stage reads
__global__ __launch_bounds__(256) void mark_blocks_kernel(int64_t n_samples_total, int S, const float* __restrict__ ro,
                                                          const float* __restrict__ rd, const double* __restrict__ z,
                                                          DevScene sc, int kmask, uint8_t* f0, uint8_t* f1, uint8_t* f2,
                                                          uint8_t* f3, int shift) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_samples_total) return;
    const int64_t ray = i / S;
    double pw[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) pw[a] = (double)ro[ray * 3 + a] + (double)rd[ray * 3 + a] * z[i];
    uint8_t* fl[4] = {f0, f1, f2, f3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!((kmask >> k) & 1) || fl[k] == nullptr) continue;
        const Vox v = make_vox(pw, k == 0 ? sc.clo : sc.lo, k == 0 ? sc.chi : sc.hi, sc.grid[k]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            int64_t idx; float w;
            corner(v, sc.grid[k], c, idx, w);
            fl[k][idx >> shift] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void to_voxel_major_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int64_t V) {
    to_vm_block(src, dst, V, blockIdx.x);
}
__global__ __launch_bounds__(256) void from_voxel_major_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                int64_t V) {
    from_vm_block(src, dst, V, blockIdx.x);
}

ENS_DEV void to_vm_block(const float* __restrict__ src, float* __restrict__ dst, int64_t V, int64_t blk) {
    __shared__ float tile[32][65];
    const int64_t v0 = blk * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int c = ty; c < 32; c += 4) {
        const int64_t v = v0 + tx;
        tile[c][tx] = v < V ? src[(int64_t)c * V + v] : 0.f;
    }
    __syncthreads();
    const int c = threadIdx.x & 31, vv = threadIdx.x >> 5;
#pragma unroll
    for (int j = vv; j < 64; j += 8) {
        const int64_t v = v0 + j;
        if (v < V) dst[v * 32 + c] = tile[c][j];
    }
}

ENS_DEV void from_vm_block(const float* __restrict__ src, float* __restrict__ dst, int64_t V, int64_t blk) {
    __shared__ float tile[32][65];
    const int64_t v0 = blk * 64;
    const int c = threadIdx.x & 31, vv = threadIdx.x >> 5;
#pragma unroll
    for (int j = vv; j < 64; j += 8) {
        const int64_t v = v0 + j;
        tile[c][j] = v < V ? src[v * 32 + c] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int cc = ty; cc < 32; cc += 4) {
        const int64_t v = v0 + tx;
        if (v < V) dst[(int64_t)cc * V + v] = tile[cc][tx];
    }
}

// ------------------------------------------------------------------ decoder re-layout
ENS_DEV void pack_body(const PackJob& job, float* __restrict__ packed, int unpack, int seg, int y, int ny) {
    const PackSeg s = job.seg[seg];
    const int n = s.rows * s.cols;
    for (int e = y * 256 + threadIdx.x; e < n; e += 256 * ny) {
        const int r = e / s.cols, c = e - r * s.cols;
        float* src = s.src + ((s.transpose & 1) ? (int64_t)c * s.src_ld + r : (int64_t)r * s.src_ld + c);
        const int cs = (s.transpose & 2) ? ((((c >> 2) ^ ((r & 15) >> 1)) << 2) | (c & 3)) : c;   // lds_util.hpp: swizzled image
        int at = r * s.dst_ld + cs;
        if (s.transpose & 4) {                                                                   // lds_util.hpp: tile-major image
            const int pp = r & 15, v = (pp >> 1) & 3, u = (pp & 1) + 2 * (pp >> 3);
            at = (r >> 4) * 16 * s.dst_ld + (c >> 4) * 256 + 64 * v + 16 * u + 4 * (((c >> 2) & 3) ^ v) + (c & 3);
        }
        float* dst = (job.packed[s.dec] ? job.packed[s.dec] : packed) + s.off + at;
        if (unpack) *src = *dst; else *dst = *src;
    }
}
__global__ __launch_bounds__(256) void pack_kernel(PackJob job, float* __restrict__ packed, int unpack) {
    pack_body(job, packed, unpack, blockIdx.x, blockIdx.y, gridDim.y);
}

// Unpack direction of a decoder whose gradient arrives as per-workgroup partial images (flush_image, render_bwd.hip:
// [16-float header, word 0 = rows | rows x part_stride floats]): the workgroup takes the 64-element groups y, y + ny, ... of
// segment `seg`; wave w sums the rows w, w + 4, ... of its group (a row's 64 elements are 256 contiguous bytes; eight
// independent loads in flight per lane), the four partial sums meet in LDS and wave 0 writes the parameter-shaped gradient.
// One group per workgroup when ny = 64 (the largest segment has 4096 elements): ~20 rows per wave instead of the ~85
// dependent-latency steps per lane of a one-thread-per-element sum (finish launch 27 -> 44 us with that form).
ENS_DEV void unpack_partial_body(const PackJob& job, int seg, int y, int ny) {
    __shared__ float red[4][64];
    const PackSeg s = job.seg[seg];
    const float* pp0 = job.part[s.dec];
    if (pp0 == nullptr) { pack_body(job, nullptr, 1, seg, y, ny); return; }     // (block-uniform)
    const int n = s.rows * s.cols, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = reinterpret_cast<const int*>(pp0)[0];
    const int64_t stride = job.part_stride[s.dec];
    for (int g = y; g * 64 < n; g += ny) {
        const int e = g * 64 + lane;
        float acc = 0.f;
        float* src = nullptr;
        if (e < n) {
            const int r = e / s.cols, c = e - r * s.cols;
            src = s.src + ((s.transpose & 1) ? (int64_t)c * s.src_ld + r : (int64_t)r * s.src_ld + c);
            const float* pp = pp0 + 16 + s.off + r * s.dst_ld + c;          // gradient images are plain row-major (build_job, fs = 0)
            int r0 = w;
            float a[8];
            for (; r0 + 28 < rows; r0 += 32) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = pp[(int64_t)(r0 + 4 * j) * stride];
                acc += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
            }
            for (; r0 < rows; r0 += 4) acc += pp[(int64_t)r0 * stride];
        }
        red[w][lane] = acc;
        __syncthreads();
        if (w == 0 && e < n) *src = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        __syncthreads();
    }
}

// The independent small jobs around a render call in ONE launch (each was a 4-6 us kernel of its own): workgroup
// ranges [decoder (un)packing | grid layout conversion | accumulator clearing].  Before the forward: pack the decoders,
// convert the touched blocks to voxel-major, clear the gradient accumulators.  After the backward: transposed-back
// grid gradients and unpacked decoder gradients.
__global__ __launch_bounds__(256) void step_kernel(PackJob pj, int unpack, ConvJob cj, int to_vm, ConvJob zj,
                                                   float* __restrict__ flat, int64_t n_flat, int nb_ray, int nb_pack, int nb_conv,
                                                   int nb_zero, RayGradArgs rg, int pack_ny, uint8_t* __restrict__ mv_need,
                                                   uint8_t* __restrict__ mv_prev, int n_move) {
    // Workgroups are dispatched in blockIdx order: the ray-gradient units (latency-bound: float64 geometry, a dependent
    // corner re-gather, a coordinate-gradient reduction; 16 us as a launch of its own) come FIRST so that they run under
    // the bandwidth-bound roles (zero-fill / transposed-back gradients) instead of behind them.
    const int b = blockIdx.x;       // (dealing the partial-image sums and the ray units alternately: finish 28 -> 31 us)
    if (b < nb_ray) {                       // ray gradients: one wave per (tile, decoder slot), four per workgroup
        const int64_t unit = (int64_t)b * 4 + (threadIdx.x >> 6);
        if (unit < (int64_t)rg.n_rays * rg.ntl * rg.n_slots) ray_grad_unit(rg, unit, (int)(threadIdx.x & 63));
    }
    else if (b < nb_ray + nb_pack) {
        if (pack_ny == 4) pack_body(pj, nullptr, unpack, (b - nb_ray) >> 2, (b - nb_ray) & 3, 4);
        else unpack_partial_body(pj, (b - nb_ray) / pack_ny, (b - nb_ray) % pack_ny, pack_ny);     // partial images to sum
    }
    else if (b < nb_ray + nb_pack + nb_conv) convert_body(cj, to_vm, b - nb_ray - nb_pack);
    else if (b < nb_ray + nb_pack + nb_conv + nb_zero) zero_body(zj, flat, n_flat, b - nb_ray - nb_pack - nb_conv);
    else {
        // block flags of gradients that live in the kernels' own layout (nothing to transpose back): the flags this step's
        // sampler marked move to `prev` -- what the next replay's prepare launch clears by and what the gradient bucket reads --
        // and are cleared for the next replay's sampler (the flag half of convert_body's persistent branch)
        const int e = (b - nb_ray - nb_pack - nb_conv - nb_zero) * 256 + (int)threadIdx.x;
        if (e < n_move) {
            const uint8_t f = mv_need[e];
            if (mv_prev[e] != f) mv_prev[e] = f;
            if (f) mv_need[e] = 0;
        }
    }
}

// sampler + prepare roles in one launch (ens_launch_sample_prepare): the rays come FIRST in blockIdx order -- a ray's wave is a
// dependent chain of float64 divisions and a 21-step shuffle sort (the whole 10 us of the launch), the other roles run under it
__global__ __launch_bounds__(256) void sample_prepare_kernel(SampleArgs A, MarkArgs mk, PackJob pj, ConvJob zj, float* __restrict__ flat,
                                                             int64_t n_flat, int nb_samp, int nb_pack) {
    const int b = blockIdx.x;
    if (b < nb_samp) {
        const int ray = b * 4 + (int)(threadIdx.x >> 6);
        if (ray < A.n_rays) sample_body(A, mk, ray, (int)(threadIdx.x & 63));
    }
    else if (b < nb_samp + nb_pack) pack_body(pj, nullptr, 0, (b - nb_samp) >> 2, (b - nb_samp) & 3, 4);
    else zero_body(zj, flat, n_flat, b - nb_samp - nb_pack);
}

}  // namespace

int ens_launch_pack(const PackJob& job, float* packed, bool unpack, hipStream_t st) {
    if (job.n <= 0) return 0;
    pack_kernel<<<dim3(job.n, 4), dim3(256), 0, st>>>(job, packed, unpack ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_transpose(const float* src, float* dst, int64_t V, bool to_vm, hipStream_t st) {
    if (V <= 0) return 0;
    const dim3 grid((unsigned)((V + 63) / 64)), block(256);
    if (to_vm) to_voxel_major_kernel<<<grid, block, 0, st>>>(src, dst, V);
    else from_voxel_major_kernel<<<grid, block, 0, st>>>(src, dst, V);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_convert(const ConvJob& job, bool to_vm, hipStream_t st) {
    if (job.n <= 0 || job.block_begin[job.n] <= 0) return 0;
    convert_kernel<<<dim3(job.block_begin[job.n]), dim3(256), 0, st>>>(job, to_vm ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_step(const PackJob& pj, bool unpack, const ConvJob& cj, bool to_vm, const ConvJob& zj, float* flat,
                    int64_t n_flat, const RayGradArgs* rg, hipStream_t st, uint8_t* mv_need, uint8_t* mv_prev, int64_t n_move) {
    bool any_part = false;
    for (int i = 0; i < 4; ++i) any_part = any_part || (unpack && pj.part[i] != nullptr);
    const int pack_ny = any_part ? 64 : 4;
    const int nb_pack = pj.n * pack_ny, nb_conv = cj.n > 0 ? cj.block_begin[cj.n] : 0;
    const int64_t nb_zero = (zj.n > 0 ? zj.block_begin[zj.n] : 0) + (flat != nullptr && n_flat > 0 ? (n_flat + 2047) / 2048 : 0);
    RayGradArgs r;
    if (rg != nullptr) r = *rg; else { r.n_rays = 0; r.ntl = 0; r.n_slots = 0; }
    const int64_t nb_ray = ((int64_t)r.n_rays * r.ntl * r.n_slots + 3) / 4;
    if (n_move < 0 || n_move > 0x7fffff00 || (n_move > 0 && (!mv_need || !mv_prev))) return -1;
    const int64_t nb_move = (n_move + 255) / 256;
    const int64_t nb = (int64_t)nb_pack + nb_conv + nb_zero + nb_ray + nb_move;
    if (nb <= 0) return 0;
    if (nb > 0x7fffffff) return -1;
    if (nb_ray > 0x7fffffff) return -1;
    step_kernel<<<dim3((unsigned)nb), dim3(256), 0, st>>>(pj, unpack ? 1 : 0, cj, to_vm ? 1 : 0, zj, flat, flat ? n_flat : 0,
                                                          (int)nb_ray, nb_pack, nb_conv, (int)nb_zero, r, pack_ny, mv_need, mv_prev,
                                                          (int)n_move);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_zero_blocks(const ConvJob& job, float* flat, int64_t n_flat, hipStream_t st) {
    const int64_t nb = job.block_begin[job.n] + (flat != nullptr && n_flat > 0 ? (n_flat + 2047) / 2048 : 0);
    if (nb <= 0) return 0;
    zero_blocks_kernel<<<dim3((unsigned)nb), dim3(256), 0, st>>>(job, flat, flat ? n_flat : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_adam(const AdamJob& job, hipStream_t st) {
    if (job.n <= 0 || job.block_begin[job.n] <= 0) return 0;
    adam_masked_kernel<<<dim3(job.block_begin[job.n]), dim3(256), 0, st>>>(job);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// gradient bucket pack / unpack (ray-sharded step).  One workgroup per 64-voxel block of every grid (unflagged blocks
// leave at once), then one per 1024 elements of the small tensors.  A block slot holds the block in the gradient's own
// layout: [C][64] for channel-major grids (256-byte row segments), [64][C] for voxel-major ones (one contiguous run).
// Voxels past the end of the grid (partial last block) travel as zeros.
// ------------------------------------------------------------------------------------------------
template <bool UNPACK>
__global__ __launch_bounds__(256) void bucket_kernel(BucketJob job) {
    const int nb = job.blk_begin[job.n_grids];
    const int b = (int)blockIdx.x;
    const int BS = job.bs;                                   // voxels per block: 64 (the renderer's flags) or 32 / 16 / 8
    if (b < nb) {
        if (!job.flags[b]) return;
        int g = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (k < job.n_grids && b >= job.blk_begin[k]) g = k;
        const int C = job.C;
        const int64_t V = job.V[g], v0 = (int64_t)(b - job.blk_begin[g]) * BS;
        float* __restrict__ G = job.grid[g];
        float* __restrict__ S = job.bucket + (int64_t)(job.pos[b] - 1) * C * BS;
        if (job.layout[g] == 0) {
            const int v = threadIdx.x % BS, cstep = 256 / BS;
            const bool in = v0 + v < V;
            for (int c = threadIdx.x / BS; c < C; c += cstep) {
                if (UNPACK) { if (in) G[(int64_t)c * V + v0 + v] = S[c * BS + v]; }
                else S[c * BS + v] = in ? G[(int64_t)c * V + v0 + v] : 0.f;
            }
        } else {
            const int64_t n_in = (V - v0 < BS ? V - v0 : BS) * C;
            for (int e = threadIdx.x; e < C * BS; e += 256) {
                if (UNPACK) { if (e < n_in) G[v0 * C + e] = S[e]; }
                else S[e] = e < n_in ? G[v0 * C + e] : 0.f;
            }
        }
        return;
    }
    const int sb = b - nb;
    int lo = 0, hi = job.n_small;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sb >= job.small_blk_begin[mid]) lo = mid; else hi = mid;
    }
    float* __restrict__ T = job.small[lo];
    float* __restrict__ S = job.bucket + job.small_off[lo];
    const int n = job.numel[lo], e0 = (sb - job.small_blk_begin[lo]) * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = e0 + j * 256 + threadIdx.x;
        if (e >= n) continue;
        if (UNPACK) T[e] = S[e]; else S[e] = T[e];
    }
}
int ens_launch_bucket(const BucketJob& job, bool unpack, hipStream_t st) {
    const int64_t blocks = (int64_t)job.blk_begin[job.n_grids] + (job.n_small > 0 ? job.small_blk_begin[job.n_small] : 0);
    if (blocks <= 0) return 0;
    if (unpack) bucket_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(job);
    else bucket_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(job);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int ens_launch_adam_tensors(const AdamTensorsJob& job, hipStream_t st) {
    if (job.n <= 0 || job.block_begin[job.n] <= 0) return 0;
    adam_tensors_kernel<<<dim3(job.block_begin[job.n]), dim3(256), 0, st>>>(job);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_mark_blocks(int stage, int n_rays, int S, const float* ro, const float* rd, const double* z,
                           const DevScene& sc, uint8_t* const* flags, hipStream_t st, int block_voxels) {
    int shift = 6;
    switch (block_voxels) { case 64: shift = 6; break; case 32: shift = 5; break; case 16: shift = 4; break; case 8: shift = 3; break; default: return -1; }
    const int64_t n = (int64_t)n_rays * S;
    if (n <= 0) return 0;
    const int kmask = stage == 0 ? 1 : (stage == 1 ? 2 : (stage == 2 ? 6 : 14));
    mark_blocks_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, S, ro, rd, z, sc, kmask, flags[0], flags[1],
                                                                               flags[2], flags[3], shift);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

static bool fill_sample_args(int n_rays, int n_lin, int n_surf, const float* ro, const float* rd, const float* gd,
                             const double* b, const float* t_lin, const double* t_surf, int lindisp, const float* t_rand,
                             float* scratch, int dmax_given, double* z, const MarkArgs* mark, SampleArgs& A, MarkArgs& mk) {
    if (n_lin + n_surf > MAX_S || n_lin < 1) return false;
    A.n_rays = n_rays; A.n_lin = n_lin; A.n_surf = gd ? n_surf : 0; A.lindisp = lindisp;
    A.dmax_inline = (gd != nullptr && !dmax_given && n_rays <= 4096) ? 1 : 0;
    A.ro = ro; A.rd = rd; A.gd = gd; A.t_lin = t_lin; A.t_surf = t_surf; A.t_rand = t_rand; A.dmax = scratch; A.zout = z;
    for (int a = 0; a < 3; ++a) { A.lo[a] = b[2 * a]; A.hi[a] = b[2 * a + 1]; }
    if (mark != nullptr) mk = *mark; else { mk.kmask = 0; mk.shift = 6; for (int k = 0; k < 4; ++k) { mk.flags[k] = nullptr; mk.flags64[k] = nullptr; } }
    return true;
}

int ens_launch_sample(int n_rays, int n_lin, int n_surf, const float* ro, const float* rd, const float* gd,
                      const double* b, const float* t_lin, const double* t_surf, int lindisp, const float* t_rand,
                      float* scratch, int dmax_given, double* z, const MarkArgs* mark, hipStream_t st) {
    if (n_rays <= 0) return 0;
    SampleArgs A;
    MarkArgs mk;
    if (!fill_sample_args(n_rays, n_lin, n_surf, ro, rd, gd, b, t_lin, t_surf, lindisp, t_rand, scratch, dmax_given, z, mark, A, mk)) return -1;
    if (gd != nullptr && !dmax_given && !A.dmax_inline) depth_max_kernel<<<1, 1024, 0, st>>>(n_rays, gd, scratch);
    sample_kernel<<<dim3(n_rays), dim3(64), 0, st>>>(A, mk);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The sampler and the forward's prepare launch as ONE launch (grids that arrive in the kernels' own layout need no conversion, so
// nothing in the prepare roles waits for the sampler's block marks): workgroups [4 rays each | decoder packing | accumulator clearing]
int ens_launch_sample_prepare(int n_rays, int n_lin, int n_surf, const float* ro, const float* rd, const float* gd,
                              const double* b, const float* t_lin, const double* t_surf, int lindisp, const float* t_rand,
                              float* scratch, int dmax_given, double* z, const MarkArgs* mark, const PackJob& pj, const ConvJob& zj,
                              float* flat, int64_t n_flat, hipStream_t st) {
    SampleArgs A;
    MarkArgs mk;
    if (!fill_sample_args(n_rays, n_lin, n_surf, ro, rd, gd, b, t_lin, t_surf, lindisp, t_rand, scratch, dmax_given, z, mark, A, mk)) return -1;
    if (n_rays < 0) return -1;
    const int nb_samp = (n_rays + 3) / 4, nb_pack = pj.n * 4;
    const int64_t nb_zero = (zj.n > 0 ? zj.block_begin[zj.n] : 0) + (flat != nullptr && n_flat > 0 ? (n_flat + 2047) / 2048 : 0);
    const int64_t nb = (int64_t)nb_samp + nb_pack + nb_zero;
    if (nb <= 0) return 0;
    if (nb > 0x7fffffff) return -1;
    if (n_rays > 0 && gd != nullptr && !dmax_given && !A.dmax_inline) depth_max_kernel<<<1, 1024, 0, st>>>(n_rays, gd, scratch);
    sample_prepare_kernel<<<dim3((unsigned)nb), dim3(256), 0, st>>>(A, mk, pj, zj, flat, flat ? n_flat : 0, nb_samp, nb_pack);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_rgbd_loss(int n, const double* depth, const float* color, const float* gd, const float* gc, float w,
                         const double* g_loss, double* loss, double* g_depth, float* g_color, hipStream_t st) {
    if (g_loss == nullptr) {
        rgbd_loss_fwd_kernel<<<1, 1024, 0, st>>>(n, depth, color, gd, gc, w, loss);
    } else if (n > 0) {
        rgbd_loss_bwd_kernel<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(n, depth, color, gd, gc, w, g_loss, g_depth, g_color);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_tracker_loss(int n, const double* depth, const double* unc, const float* color, const float* gd,
                            const float* gc, float w, const double* g_loss, double* loss, double* g_depth, float* g_color,
                            hipStream_t st) {
    if (g_loss == nullptr) {
        tracker_loss_fwd_kernel<<<1, 1024, 0, st>>>(n, depth, unc, color, gd, gc, w, loss);
    } else if (n > 0) {
        tracker_loss_bwd_kernel<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(n, depth, unc, color, gd, gc, w, g_loss, g_depth, g_color);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// common.get_sample_uv / select_uv (common.py:125-141) behind the single torch.randint draw: pixel k of the window
// [H0, H0+hh) x [W0, W0+ww) -> its column / row as floats (the reference indexes a linspace meshgrid of integers) and its depth
// and colour samples.  Replaces 2 linspace + 2 index-arithmetic + 4 indexing launches per sampled frame.
template <typename C>
__global__ __launch_bounds__(256) void gather_pixels_kernel(int n, const int64_t* __restrict__ idx, int H0, int W0, int ww, int Wimg,
                                                            const float* __restrict__ depth, const C* __restrict__ color,
                                                            float* __restrict__ oi, float* __restrict__ oj, float* __restrict__ od,
                                                            C* __restrict__ oc) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int64_t k = idx[t];
    const int row = (int)(k / ww), col = (int)(k - (int64_t)row * ww);
    const int64_t at = (int64_t)(H0 + row) * Wimg + (W0 + col);
    oi[t] = (float)(W0 + col);
    oj[t] = (float)(H0 + row);
    od[t] = depth[at];
#pragma unroll
    for (int a = 0; a < 3; ++a) oc[(int64_t)t * 3 + a] = color[at * 3 + a];
}
int ens_launch_gather_pixels(int n, const int64_t* idx, int H0, int W0, int ww, int Wimg, const float* depth, const void* color,
                             int color_f64, float* oi, float* oj, float* od, void* oc, hipStream_t st) {
    if (n <= 0) return 0;
    const dim3 grid((n + 255) / 256), block(256);
    if (color_f64) gather_pixels_kernel<double><<<grid, block, 0, st>>>(n, idx, H0, W0, ww, Wimg, depth, (const double*)color, oi, oj, od, (double*)oc);
    else gather_pixels_kernel<float><<<grid, block, 0, st>>>(n, idx, H0, W0, ww, Wimg, depth, (const float*)color, oi, oj, od, (float*)oc);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The head of the tracker's camera iteration in one single-workgroup launch (Tracker.py:160-174): the pixels of the batch
// (get_sample_uv behind the single randint draw: gather_pixels_kernel), their rays from the camera tensor
// (pose_rays_fwd_kernel), the in-bound prefilter  t = min_axis max_side((bound - o) / d) >= gt_depth  in float64 as a mask
// and the sampler's batch maxima of gt_depth over the rays that mask keeps -- gather + pose + ~12 tiny torch launches before.
template <typename C>
__global__ __launch_bounds__(1024) void tracker_rays_kernel(int n, const float* __restrict__ ct, const int64_t* __restrict__ idx,
                                                            int H0, int W0, int ww, int Wimg, int Himg, const float* __restrict__ depth,
                                                            const C* __restrict__ color, float fx, float fy, float cx, float cy,
                                                            double lo0, double hi0, double lo1, double hi1, double lo2, double hi2,
                                                            float* __restrict__ oi, float* __restrict__ oj, float* __restrict__ ro,
                                                            float* __restrict__ rd, float* __restrict__ gd, float* __restrict__ gc,
                                                            uint8_t* __restrict__ inside, float* __restrict__ dmax,
                                                            int* __restrict__ iter, int iter_count) {
    __shared__ float red[16];
    if (iter != nullptr) {                               // pre-drawn indices of a whole frame: this call takes row iter[0] % iter_count
        const int k = iter[0];
        idx += (int64_t)(k % iter_count) * n;
        __syncthreads();                                 // (one workgroup: everyone has read the count before it moves on)
        if (threadIdx.x == 0) iter[0] = k + 1;
    }
    const PoseR R = pose_rotation(ct);
    const double lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    float m = 0.f;                                   // max over the kept rays of gt_depth (0 when none: torch.where(inside, gd, 0).max())
    for (int t = threadIdx.x; t < n; t += 1024) {
        int64_t k = idx[t];
        const int64_t kmax = (int64_t)(Himg - H0) * ww - 1;            // (an index outside the window's rows never leaves the image)
        k = k < 0 ? 0 : (k > kmax ? kmax : k);
        const int row = (int)(k / ww), col = (int)(k - (int64_t)row * ww);
        const int64_t at = (int64_t)(H0 + row) * Wimg + (W0 + col);
        const float pi = (float)(W0 + col), pj = (float)(H0 + row), g = depth[at];
        oi[t] = pi; oj[t] = pj; gd[t] = g;
#pragma unroll
        for (int a = 0; a < 3; ++a) gc[(int64_t)t * 3 + a] = (float)color[at * 3 + a];
        const float d0 = (pi - cx) / fx, d1 = -(pj - cy) / fy, d2 = -1.f;
        double tmin = INFINITY;
        bool nan = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float dv = (d0 * R.r[a][0] + d1 * R.r[a][1]) + d2 * R.r[a][2], ov = ct[4 + a];
            rd[t * 3 + a] = dv;
            ro[t * 3 + a] = ov;
            const double t0 = (lo[a] - (double)ov) / (double)dv, t1 = (hi[a] - (double)ov) / (double)dv;
            const double tm = t0 > t1 ? t0 : t1;
            nan = nan || t0 != t0 || t1 != t1;           // (torch.max / torch.min propagate NaN: the comparison below is then False)
            tmin = tm < tmin ? tm : tmin;
        }
        bool in = true;
        if (inside != nullptr) {
            in = !nan && tmin >= (double)g;
            inside[t] = in ? 1 : 0;
        }
        if (in) m = fmaxf(m, g);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 16) {
        m = red[threadIdx.x];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (threadIdx.x == 0) { dmax[0] = m; dmax[1] = m * 1.2f; }
    }
}
int ens_launch_tracker_rays(int n, const float* ct, const int64_t* idx, int H0, int W0, int ww, int Wimg, int Himg, const float* depth,
                            const void* color, int color_f64, float fx, float fy, float cx, float cy, const double* b,
                            float* oi, float* oj, float* ro, float* rd, float* gd, float* gc, uint8_t* inside, float* dmax,
                            hipStream_t st, int* iter, int iter_count) {
    if (n <= 0) return 0;
    if ((iter != nullptr && iter_count < 1) || ww < 1 || W0 < 0 || H0 < 0 || W0 + ww > Wimg || H0 >= Himg) return -1;
    if (color_f64) tracker_rays_kernel<double><<<1, 1024, 0, st>>>(n, ct, idx, H0, W0, ww, Wimg, Himg, depth, (const double*)color, fx, fy, cx, cy,
                                                                   b[0], b[1], b[2], b[3], b[4], b[5], oi, oj, ro, rd, gd, gc, inside, dmax, iter, iter_count);
    else tracker_rays_kernel<float><<<1, 1024, 0, st>>>(n, ct, idx, H0, W0, ww, Wimg, Himg, depth, (const float*)color, fx, fy, cx, cy,
                                                        b[0], b[1], b[2], b[3], b[4], b[5], oi, oj, ro, rd, gd, gc, inside, dmax, iter, iter_count);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_pose_rays(int n, const float* ct, const float* pi, const float* pj, float fx, float fy, float cx, float cy,
                         const float* g_ro, const float* g_rd, float* ro, float* rd, float* g_ct, hipStream_t st) {
    if (g_ct == nullptr) {
        if (n > 0) pose_rays_fwd_kernel<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(n, ct, pi, pj, fx, fy, cx, cy, ro, rd);
    } else {
        // (as many waves as the batch has 64-ray slices, at most 16: fewer partials to reduce, same summation order per wave count)
        const int thr = n <= 64 ? 64 : (n <= 256 ? 256 : 1024);
        pose_rays_bwd_kernel<<<1, thr, 0, st>>>(n, ct, pi, pj, fx, fy, cx, cy, g_ro, g_rd, g_ct);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_ray_points(int n_rays, int S, const float* ro, const float* rd, const double* z, const double* b,
                          double* pts, uint8_t* mask, hipStream_t st) {
    const int64_t n = (int64_t)n_rays * S;
    if (n <= 0) return 0;
    ray_points_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n_rays, S, ro, rd, z, b[0], b[1], b[2],
                                                                                b[3], b[4], b[5], pts, mask);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_sincos(int64_t n, const float* x, float* s, float* c, hipStream_t st) {
    if (n <= 0) return 0;
    sincos_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, x, s, c);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ens_launch_voxel_index(int64_t n, const double* pts, const double* b, int D, int H, int W, int* ix, int* iy,
                           int* iz, float* fx, float* fy, float* fz, hipStream_t st) {
    if (n <= 0) return 0;
    DevGrid g{nullptr, D, H, W};
    voxel_index_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, pts, b[0], b[1], b[2], b[3], b[4],
                                                                                 b[5], g, ix, iy, iz, fx, fy, fz);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
