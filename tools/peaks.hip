// Peak micro-benchmarks on the box (the datasheet peaks used by the roofline are cross-checked by measurement):
//   fp32 MFMA (v_mfma_f32_16x16x4_f32, the instruction every kernel of the path uses), HBM read, HBM copy.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/exp/peaks tools/peaks.hip ; run: build/exp/peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ src, float* out, size_t n4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) s += src[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
template <typename F> float time_ms(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    float* out; CK(hipMalloc(&out, 256 * 4096 * sizeof(float)));
    for (int wg_per_cu : {1, 2, 4}) {
        const int blocks = p.multiProcessorCount * wg_per_cu, iters = 20000;
        const float ms = time_ms([&] { mfma_kernel<<<blocks, 256>>>(out, iters); }, 5);
        const double flops = (double)blocks * 4 /*waves*/ * iters * 8 * 2048.0;
        printf("fp32 MFMA 16x16x4, %d waves/SIMD: %.1f TFLOP/s\n", wg_per_cu, flops / ms / 1e9);
    }
    const size_t bytes = (size_t)4 << 30, n4 = bytes / 16;
    f32x4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    for (int mult : {4, 8, 16}) {
        const int blocks = p.multiProcessorCount * mult;
        const float r = time_ms([&] { read_kernel<<<blocks, 256>>>(a, out, n4); }, 5);
        const float c = time_ms([&] { copy_kernel<<<blocks, 256>>>(a, b, n4); }, 5);
        printf("HBM, %d workgroups/CU over 4 GiB: read %.2f TB/s, copy (read+write) %.2f TB/s\n", mult, bytes / r / 1e9, 2.0 * bytes / c / 1e9);
    }
    return 0;
}
