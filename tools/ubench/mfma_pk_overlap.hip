// Does VALU work issue under a running MFMA of the same wave (one wave per SIMD)?  Loops of  1 x v_mfma_f32_32x32x16_f16 + N x {v_pk_fma_f32 |
// v_fma_f32 | ds_read2_b32}, cycles per iteration by s_memtime.   hipcc -O3 --offload-arch=gfx950 mfma_pk_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

template <int MODE, int N, bool MFMA>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int iters) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i * 0.5f); }
    floatx16 acc[4];
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    floatx2 p[6], w = {1.0001f, 0.9999f};
    for (int i = 0; i < 6; ++i) p[i] = floatx2{(float)threadIdx.x, (float)i};
    float f[6];
    for (int i = 0; i < 6; ++i) f[i] = threadIdx.x + i;
    const unsigned la = (unsigned)(threadIdx.x * 4);
    floatx2 d = {0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MFMA) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < N; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i % 6]) : "v"(w));
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i % 6]) : "v"(w[0]));
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(d) : "v"(la));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float s = d[0] + d[1];
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    for (int i = 0; i < 6; ++i) s += p[i][0] + p[i][1] + f[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int N, bool MFMA>
void run(const char *name, float *out, long long *cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, N, MFMA>), dim3(1024), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, N, MFMA>), dim3(1024), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f cycles per (MFMA + fillers)\n", name, (double)c / iters / 4);
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 8);
    run<0, 0, true>("MFMA alone", out, cyc);
    run<0, 6, false>("6 v_pk_fma_f32 alone", out, cyc);
    run<0, 6, true>("MFMA + 6 v_pk_fma_f32", out, cyc);
    run<0, 3, true>("MFMA + 3 v_pk_fma_f32", out, cyc);
    run<1, 6, false>("6 v_fma_f32 alone", out, cyc);
    run<1, 6, true>("MFMA + 6 v_fma_f32", out, cyc);
    run<1, 12, true>("MFMA + 12 v_fma_f32", out, cyc);
    run<2, 6, false>("6 ds_read2_b32 alone", out, cyc);
    run<2, 6, true>("MFMA + 6 ds_read2_b32", out, cyc);
    return 0;
}
