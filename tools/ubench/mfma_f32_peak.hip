// Microbenchmark: achievable fp32 matrix-core rate (v_mfma_f32_32x32x2_f32 / 16x16x4_f32) with N independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float seed) {
    float a = seed + threadIdx.x * 0.001f, b = seed * 0.5f + threadIdx.x;
    floatx16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float seed) {
    float a = seed + threadIdx.x * 0.001f, b = seed * 0.5f + threadIdx.x;
    floatx4 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *out; hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        const int grid = 256 * wg_per_cu;
        double ms;
        ms = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
        printf("32x32x2 f32 NACC=4 wg/cu=%d: %.1f TF (%.3f ms)\n", wg_per_cu, 2.0 * 32 * 32 * 2 * 4 * iters * grid * 4 / (ms * 1e-3) / 1e12, ms);
        ms = timeit([&] { hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
        printf("32x32x2 f32 NACC=2 wg/cu=%d: %.1f TF\n", wg_per_cu, 2.0 * 32 * 32 * 2 * 2 * iters * grid * 4 / (ms * 1e-3) / 1e12);
        ms = timeit([&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
        printf("32x32x2 f32 NACC=1 wg/cu=%d: %.1f TF\n", wg_per_cu, 2.0 * 32 * 32 * 2 * 1 * iters * grid * 4 / (ms * 1e-3) / 1e12);
        ms = timeit([&] { hipLaunchKernelGGL(k16<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
        printf("16x16x4 f32 NACC=8 wg/cu=%d: %.1f TF\n", wg_per_cu, 2.0 * 16 * 16 * 4 * 8 * iters * grid * 4 / (ms * 1e-3) / 1e12);
        ms = timeit([&] { hipLaunchKernelGGL(k16<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
        printf("16x16x4 f32 NACC=2 wg/cu=%d: %.1f TF\n", wg_per_cu, 2.0 * 16 * 16 * 4 * 2 * iters * grid * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
