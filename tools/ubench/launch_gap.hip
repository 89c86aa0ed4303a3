// Microbenchmark: cost of a chain of dependent tiny kernels in one stream, eager launches vs one hipGraph replay.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(float *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
int main() {
    float *d;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    hipStream_t s;
    hipStreamCreate(&s);
    const int N = 150, reps = 50;
    for (int grid : {1, 256, 2048}) {
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, 1 << 18);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, 1 << 18);
        hipStreamSynchronize(s);
        auto t1 = std::chrono::high_resolution_clock::now();
        const double eager = std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * N);
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, 1 << 18);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t1 = std::chrono::high_resolution_clock::now();
        const double graph = std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * N);
        printf("grid %5d: eager %.2f us/kernel, graph %.2f us/kernel\n", grid, eager, graph);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    return 0;
}
