// Does the plane stride of an NCHW fp32 tensor matter to HBM / MALL bandwidth when a wave reads the SAME pixels of many channel planes at once
// (the access pattern of the detector's pointwise / conv_dw kernels: 128- or 256-byte row segments, one per channel, `plane` bytes apart)?
// Reads B x C planes of HW floats with wave = 32 (or 64) consecutive pixels x all C channels, planes `stride` floats apart; prints TB/s per stride.
//   hipcc -O3 --offload-arch=gfx950 -o plane_stride plane_stride.hip && ./plane_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PIX>  // pixels per wave: 32 (lane (r, hi): 8 channels per 16-group, like pw_mfma_kernel) or 64 (lane = pixel, all channels)
__global__ __launch_bounds__(256) void rd(const float *in, float *out, int B, int C, int HW, long stride, int n_groups) {
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int gw = blockIdx.x * 4 + (threadIdx.x >> 6); gw < n_groups; gw += gridDim.x * 4) {
        const long g = (long)gw * PIX + (PIX == 32 ? (lane & 31) : lane);
        const int b = (int)(g / HW), p = (int)(g - (long)b * HW);
        const float *x = in + (long)b * C * stride + p;
        if (PIX == 32) {
            const int hi = lane >> 5;
            float v[32];
#pragma unroll 1
            for (int c0 = 0; c0 < C; c0 += 64) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[q * 8 + e] = x[(long)(c0 + 16 * q + 8 * hi + e) * stride];
#pragma unroll
                for (int i = 0; i < 32; ++i) s += v[i];
            }
        } else {
            float v[32];
#pragma unroll 1
            for (int c0 = 0; c0 < C; c0 += 32) {
#pragma unroll
                for (int e = 0; e < 32; ++e) v[e] = x[(long)(c0 + e) * stride];
#pragma unroll
                for (int i = 0; i < 32; ++i) s += v[i];
            }
        }
    }
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int B = 32;
    float *in, *out;
    const size_t cap = (size_t)B * 64 * (102400 + 4096) * 4 + (1 << 20);
    CK(hipMalloc(&in, cap)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(in, 0, cap));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { int C, HW; } cases[] = {{64, 6400}, {128, 1600}, {32, 25600}, {16, 102400}, {256, 400}};
    for (auto cs : cases) {
        for (int pad : {0, 16, 64, 80, 256, 1024}) {
            const long stride = cs.HW + pad;
            for (int pix : {32, 64}) {
                const int n_groups = (int)((long)B * cs.HW / pix);
                auto launch = [&] {
                    if (pix == 32) hipLaunchKernelGGL(rd<32>, dim3(512), dim3(256), 0, 0, in, out, B, cs.C, cs.HW, stride, n_groups);
                    else hipLaunchKernelGGL(rd<64>, dim3(512), dim3(256), 0, 0, in, out, B, cs.C, cs.HW, stride, n_groups);
                };
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e0, 0));
                const int reps = 20;
                for (int i = 0; i < reps; ++i) launch();
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)B * cs.C * cs.HW * 4;
                printf("C=%3d HW=%6d (%5.1f MB) pad %4d floats  wave=%2d px : %7.2f us  %.2f TB/s\n", cs.C, cs.HW, bytes / 1e6, pad, pix, ms * 1000 / reps, bytes / (ms / reps * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
