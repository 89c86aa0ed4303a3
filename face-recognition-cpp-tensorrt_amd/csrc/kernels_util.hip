// Small utility kernels of libfrt.so (gfx950).
#include "frt_kernels.h"

namespace {

// Busy-waits for `ticks` of the constant 100 MHz device clock (s_memrealtime): one wave, no memory traffic.  Used by the pipeline's
// stream-overlap self-check: two of these on two streams take `ticks` together when the streams sit on different hardware queues and
// 2 x `ticks` when they share one (a hardware queue runs its kernels in order).
__global__ __launch_bounds__(64) void spin_kernel(long ticks, long *sink) {
    const long t0 = (long)__builtin_amdgcn_s_memrealtime();
    long t = t0;
    while (t - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        t = (long)__builtin_amdgcn_s_memrealtime();
    }
    if (sink && threadIdx.x == 0 && ticks < 0) *sink = t;  // never true: keeps the loop observable
}

// Sustained matrix-core rate probe (the in-library form of tools/ubench/clock_probe.hip): one 4-wave workgroup per CU, every wave runs
// `iters` x 28 v_mfma_f32_32x32x16_f16 (7 accumulators x 4 k-steps: conv_patch_kernel's K loop shape) on RANDOM fp16 operands, optionally with
// one ds_read_b128 per MFMA (LDSR) and four 16-byte global loads per 28 MFMAs (VMEM) - the K loop's instruction mix.  The part is power-managed:
// what it sustains on this mix is well below the nominal 2.5 PFLOP/s, and bench.py reports the dominant kernel against both (roofline.sustained_peak).
template <bool LDSR, bool VMEM>
__global__ __launch_bounds__(256) void mfma_probe_kernel(const half8 *__restrict__ src, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char probe_smem[];
    half8 *l = reinterpret_cast<half8 *>(probe_smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) l[i] = src[i];
    __syncthreads();
    half8 b[7], a[4];
#pragma unroll
    for (int j = 0; j < 7; ++j) b[j] = l[(lane * 9 + j * 64) & 4095];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) a[kq] = src[(tid + kq * 256) & 4095];
    floatx16 acc[7];
#pragma unroll
    for (int n = 0; n < 7; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
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
            if (VMEM) a[kq] = g[(((it * 4 + kq) * 256) & 4095 & ~2047) | ((it * 4 + kq) * 256 & 1023)];
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < 7; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

}  // namespace

// mix 0: MFMAs only, 1: + one ds_read_b128 per MFMA, 2: + the global loads as well.  src: 8192 half8 of random fp16 values, out: n_wg * 256 floats.
// Returns the flop one launch performs.
double launch_mfma_probe(int mix, int n_wg, int iters, const void *src, float *out, hipStream_t s) {
    const half8 *p = static_cast<const half8 *>(src);
    const size_t lds = 65536;
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mfma_probe_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mfma_probe_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mfma_probe_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (mix == 0) hipLaunchKernelGGL((mfma_probe_kernel<false, false>), dim3(n_wg), dim3(256), lds, s, p, out, iters);
    else if (mix == 1) hipLaunchKernelGGL((mfma_probe_kernel<true, false>), dim3(n_wg), dim3(256), lds, s, p, out, iters);
    else hipLaunchKernelGGL((mfma_probe_kernel<true, true>), dim3(n_wg), dim3(256), lds, s, p, out, iters);
    return 2.0 * 32 * 32 * 16 * 28.0 * iters * 4.0 * n_wg;
}

void launch_spin(double microseconds, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, (long)(microseconds * 100.0), (long *)nullptr);
}
