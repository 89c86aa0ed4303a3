// Microbenchmark: what does the shader clock do under the recogniser's instruction mix?  One wave per SIMD on every CU runs
// `iters` x 28 v_mfma_f32_32x32x16_f16 (7 accumulators x 4), optionally with one ds_read_b128 per MFMA (random fp16 data in LDS) and
// optionally 4 global_load_dwordx4 per 28 MFMAs; effective clock = (MFMAs per wave x 32 cycles) / elapsed.  Short (about 20 us)
// and long (about 2 ms) launches, isolated and back-to-back, show whether power management or instruction issue sets the rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <bool LDSR, bool VMEM, bool RAND>
__global__ __launch_bounds__(256) void k(const half8 *__restrict__ src, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8 *l = reinterpret_cast<half8 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) l[i] = src[(RAND ? i : 0) + blockIdx.x * 0];
    __syncthreads();
    half8 b[7], a[4];
    for (int j = 0; j < 7; ++j) b[j] = l[(lane * 9 + j * 64) & 4095];
    for (int kq = 0; kq < 4; ++kq) a[kq] = src[(RAND ? (tid + kq * 256) : 0) & 4095];
    floatx16 acc[7];
    for (int n = 0; n < 7; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    const half8 *g = src + (blockIdx.x & 1) * 2048 + tid;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
            for (int n = 0; n < 7; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kq], b[n], acc[n], 0, 0, 0);
                if (LDSR) b[n] = l[(lane * 9 + n * 64 + (it * 4 + kq) * 17) & 4095];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (VMEM) a[kq] = g[((it * 4 + kq) * 256) & 4095 & ~2047 | ((it * 4 + kq) * 256 & 1023)];
        }
    }
    float s = 0.f;
    for (int n = 0; n < 7; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    half8 *src; float *out; hipMalloc(&src, 8192 * 16); hipMalloc(&out, 1 << 24);
    _Float16 *h = (_Float16 *)malloc(8192 * 16);
    for (int i = 0; i < 8192 * 8; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) * 0.001f);
    hipMemcpy(src, h, 8192 * 16, hipMemcpyHostToDevice);
    const int lds = 65536;
    auto report = [&](const char *name, double ms, int iters) {
        const double cyc = (double)iters * 28 * 32;
        printf("%-34s iters=%5d  %8.1f us  effective %.2f GHz  (%.0f TF)\n", name, iters, ms * 1e3, cyc / (ms * 1e-3) / 1e9,
               2.0 * 32 * 32 * 16 * 28 * iters * 1024 / (ms * 1e-3) / 1e12);
    };
    for (int iters : {36, 3600}) {
        const int reps = iters == 36 ? 50 : 3;
        report("mfma const operands", timeit([&] { hipLaunchKernelGGL((k<false, false, false>), dim3(256), dim3(256), lds, 0, src, out, iters); }, reps), iters);
        report("mfma random operands", timeit([&] { hipLaunchKernelGGL((k<false, false, true>), dim3(256), dim3(256), lds, 0, src, out, iters); }, reps), iters);
        report("mfma + ds_read_b128 (random)", timeit([&] { hipLaunchKernelGGL((k<true, false, true>), dim3(256), dim3(256), lds, 0, src, out, iters); }, reps), iters);
        report("mfma + ds_read + global loads", timeit([&] { hipLaunchKernelGGL((k<true, true, true>), dim3(256), dim3(256), lds, 0, src, out, iters); }, reps), iters);
        report("mfma + global loads", timeit([&] { hipLaunchKernelGGL((k<false, true, true>), dim3(256), dim3(256), lds, 0, src, out, iters); }, reps), iters);
    }
    return 0;
}
