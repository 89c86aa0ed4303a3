// how many workgroups of T threads with L bytes of dynamic LDS run at once on a CU?  each workgroup spins ~20 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long cycles, int *sink) {
    extern __shared__ char sm[];
    sm[threadIdx.x] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (sm[threadIdx.x] == 77) *sink = 1;
}
int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 128;
    int *sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds : {1024, 16384, 32768, 40960, 51200, 53248, 65536, 81920}) {
        hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        printf("T=%d lds=%6d:", T, lds);
        for (int k = 1; k <= 6; ++k) {
            hipLaunchKernelGGL(spin, dim3(256 * k), dim3(T), lds, 0, 2000LL, sink);  // warm
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(256 * k), dim3(T), lds, 0, 2000LL, sink);   // 2000 ticks of the 100 MHz wall clock = 20 us
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %dx256: %5.1f us", k, ms * 1e3);
        }
        printf("\n");
    }
    return 0;
}
