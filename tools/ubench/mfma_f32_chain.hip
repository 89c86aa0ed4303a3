// Microbenchmark: v_mfma_f32_32x32x2_f32 in 1 / 2 / 4 independent accumulator chains per wave, one or two waves per SIMD, all CUs or 200 workgroups:
// what does a DEPENDENT fp32 MFMA cost issue to issue?  (conv32_kernel's K loop is one chain per wave at less than one wave per SIMD.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(const float *__restrict__ src, float *out, int iters) {
    const int tid = threadIdx.x;
    float a = src[tid & 1023], b = src[(tid * 7 + 1) & 1023];
    floatx16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
    float *src, *out; (void)hipMalloc(&src, 4096); (void)hipMalloc(&out, 1 << 24);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 201 - 100) * 0.01f;
    (void)hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    const int total = 4000 * 4;  // MFMAs per wave
    for (int wgs : {200, 256, 512}) {
        double m1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, src, out, total / 4); });
        double m2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, src, out, total / 8); });
        double m4 = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, src, out, total / 16); });
        printf("workgroups %3d (x 4 waves): ns per MFMA per wave: 1 chain %.1f  2 chains %.1f  4 chains %.1f   (64 cycles at 2.4 GHz = 26.7 ns)\n", wgs, m1 * 1e6 / total, m2 * 1e6 / total,
               m4 * 1e6 / total);
    }
    return 0;
}
