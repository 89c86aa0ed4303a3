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

}  // namespace

void launch_spin(double microseconds, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, (long)(microseconds * 100.0), (long *)nullptr);
}
