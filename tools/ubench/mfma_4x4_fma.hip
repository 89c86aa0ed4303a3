// Is v_mfma_f32_4x4x1_16B_f32 a per-lane FMA engine?  16 blocks of 4x4 outer products: D[i][j] += A[i] * B[j] per block; with A = the same four
// weights in every block (lane l holds w[l & 3]) and B = the lane's own value, lane l's four accumulator registers become
// acc[i] += w[i] * x  - four fused multiply-adds of the lane's value with four weights, on the MATRIX pipe, the weights in a vector register.
// Checks (a) bit equality with an fmaf chain over 27 steps for random data, (b) the rate with 1 / 2 / 4 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void check(const float *__restrict__ w /*[27][4]*/, const float *__restrict__ x /*[64][27]*/, float *out_m, float *out_v) {
    const int l = threadIdx.x;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 27; ++t) {
        const float a = w[t * 4 + (l & 3)], b = x[l * 27 + t];
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
        for (int i = 0; i < 4; ++i) v[i] = fmaf(w[t * 4 + i], b, v[i]);
    }
    for (int i = 0; i < 4; ++i) { out_m[l * 4 + i] = acc[i]; out_v[l * 4 + i] = v[i]; }
}
template <int NCH>
__global__ __launch_bounds__(192) void rate(const float *__restrict__ src, float *out, int iters) {
    const int l = threadIdx.x;
    float a = src[l & 1023], b = src[(l * 7 + 1) & 1023];
    floatx4 acc[NCH];
    for (int n = 0; n < NCH; ++n) acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int n = 0; n < NCH; ++n) acc[n] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NCH; ++n) for (int i = 0; i < 4; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + l] = s;
}
template <typename F>
double timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
    float hw[108], hx[64 * 27], hm[256], hv[256];
    srand(7);
    for (float &f : hw) f = (rand() % 20001 - 10000) * 1e-4f;
    for (float &f : hx) f = (float)(rand() % 256) - 117.f;
    float *w, *x, *om, *ov, *src, *out;
    (void)hipMalloc(&w, sizeof hw); (void)hipMalloc(&x, sizeof hx); (void)hipMalloc(&om, 1024); (void)hipMalloc(&ov, 1024); (void)hipMalloc(&src, 4096); (void)hipMalloc(&out, 1 << 24);
    (void)hipMemcpy(w, hw, sizeof hw, hipMemcpyHostToDevice); (void)hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, w, x, om, ov);
    (void)hipMemcpy(hm, om, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(hv, ov, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += memcmp(&hm[i], &hv[i], 4) != 0;
    printf("4x4x1 MFMA chain vs fmaf chain, 27 steps, 256 values: %d differ (first: %.9g / %.9g)\n", bad, hm[0], hv[0]);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 201 - 100) * 0.01f;
    (void)hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    const int total = 8000 * 8;  // MFMAs per wave
    for (int wgs : {256, 1280}) {
        double m1 = timeit([&] { hipLaunchKernelGGL(rate<1>, dim3(wgs), dim3(192), 0, 0, src, out, total / 8); });
        double m2 = timeit([&] { hipLaunchKernelGGL(rate<2>, dim3(wgs), dim3(192), 0, 0, src, out, total / 16); });
        double m4 = timeit([&] { hipLaunchKernelGGL(rate<4>, dim3(wgs), dim3(192), 0, 0, src, out, total / 32); });
        printf("workgroups %4d x 3 waves: ns per 4x4x1 MFMA per wave: 1 chain %.2f  2 chains %.2f  4 chains %.2f   (8 cycles at 2.4 GHz = 3.33 ns; a v_pk_fma_f32 does half the FMAs in 4 cycles)\n", wgs,
               m1 * 1e6 / total, m2 * 1e6 / total, m4 * 1e6 / total);
    }
    return 0;
}
