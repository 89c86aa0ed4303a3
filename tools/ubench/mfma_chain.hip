// Microbenchmark: one dependent accumulator chain of v_mfma_f32_32x32x16_f16 per wave (the split-fp16 detector kernel's inner
// loop: hi*hi, hi*lo, lo*hi into ONE accumulator), 1 or 2 waves per SIMD, with the "lo" operands in the normal or the subnormal
// fp16 range - does a subnormal operand or the dependency slow the matrix pipe?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const half8 *__restrict__ src, float *out, int iters, float lo_scale) {
    const int tid = threadIdx.x;
    half8 ah = src[tid & 4095], bh = src[(tid * 7 + 1) & 4095], al, bl;
    for (int i = 0; i < 8; ++i) { al[i] = (_Float16)((float)src[(tid + 99) & 4095][i] * lo_scale); bl[i] = (_Float16)((float)src[(tid + 77) & 4095][i] * lo_scale); }
    floatx16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[n], 0, 0, 0);
        }
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
    half8 *src; float *out; (void)hipMalloc(&src, 4096 * 16); (void)hipMalloc(&out, 1 << 24);
    _Float16 *h = (_Float16 *)malloc(4096 * 16);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) * 0.001f);
    (void)hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int wg = 1; wg <= 2; ++wg)
        for (float sc : {1.0f, 2.44e-4f, 0.f}) {
            double ms1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(256 * wg), dim3(256), 0, 0, src, out, iters * 2, sc); });
            double ms2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(256 * wg), dim3(256), 0, 0, src, out, iters, sc); });
            const double mf = (double)iters * 2 * 3 * wg;  // MFMAs per SIMD
            printf("waves/SIMD=%d lo_scale=%-8g  1 chain: %.1f clk/MFMA@2.4GHz (%.0f TF)   2 chains: %.1f clk (%.0f TF)\n", wg, sc, ms1 * 1e-3 * 2.4e9 / mf,
                   mf * 1024 * 32768 / (ms1 * 1e-3) / 1e12, ms2 * 1e-3 * 2.4e9 / mf, mf * 1024 * 32768 / (ms2 * 1e-3) / 1e12);
        }
    return 0;
}
