// How fast does the part START workgroups?  A kernel that does nothing but one store per workgroup, launched with many small workgroups:
// duration / workgroups = the dispatch cost per workgroup, the floor under every "one small tile per workgroup" kernel of the detector.
// Also with a body that lives ~ 5 us (the conv_dw block's workgroup lifetime), to see how many workgroups are resident together.
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_rate dispatch_rate.hip && ./dispatch_rate
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void empty_kernel(int *out) {
    if (threadIdx.x == 0 && out) out[blockIdx.x & 1023] = 1;
}
template <int VG>
__global__ void spin_kernel(int *out, unsigned long long ticks) {  // ticks of the 100 MHz constant clock; VG: registers kept live (occupancy)
    float v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < VG; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += v[i];
    if (s == 12345.678f && out) out[0] = 1;
}

template <class F>
static float time_us(F &&launch, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    int *d = nullptr;
    hipMalloc(&d, 4096);
    const int grids[] = {256, 800, 1600, 3200, 6400, 12800, 51200};
    for (int threads : {64, 256}) {
        for (size_t lds : {(size_t)0, (size_t)9216, (size_t)40960}) {
            std::printf("empty kernel, %3d threads, %5zu B LDS:", threads, lds);
            for (int g : grids) {
                const float us = time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(threads), lds, 0, d); });
                std::printf("  %5d wg %6.1f us (%4.1f ns/wg)", g, us, 1e3f * us / g);
            }
            std::printf("\n");
        }
    }
    for (int g : {256, 1600, 6400}) {
        const float u88 = time_us([&] { hipLaunchKernelGGL(spin_kernel<64>, dim3(g), dim3(256), 9216, 0, d, 500ull); });
        const float u16 = time_us([&] { hipLaunchKernelGGL(spin_kernel<8>, dim3(g), dim3(256), 9216, 0, d, 500ull); });
        std::printf("5 us workgroups of 256 threads, %5d wg: %7.1f us with ~80 live registers, %7.1f us with ~16 (256 CUs: %.1f workgroups per CU)\n", g, u88, u16, g / 256.0);
    }
    return 0;
}
