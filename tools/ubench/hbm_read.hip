// Microbenchmark: streaming-read bandwidth of a 1 GiB buffer (16-byte loads, grid-stride), various grids.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const floatx4 *__restrict__ p, long n, float *out) {
    floatx4 a = {0, 0, 0, 0};
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const floatx4 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
        a += v0 + v1 + v2 + v3;
    }
    for (; i < n; i += stride) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == 12345.f) out[0] = 1.f;
}
int main() {
    const long bytes = 1L << 30, n = bytes / 16;
    floatx4 *d; float *o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4);
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
        hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, d, n, o);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, d, n, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %6d: %.2f TB/s (%.1f us per GiB)\n", grid, 5.0 * bytes / (ms * 1e-3) / 1e12, ms * 200);
    }
    return 0;
}
