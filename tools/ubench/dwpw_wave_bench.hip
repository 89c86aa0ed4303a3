// Stand-alone harness for dwpw_wave_kernel (kernels_det_wave.hip): the 128 -> 128 conv_dw block at 40x40 on random data, checked against a
// plain fp32 / fp64 kernel, timed with HIP events.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFRT_TUNING -I<csrc> dwpw_wave_bench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels_det_wave.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void ref_kernel(const float *in, const float *wd, const float *bd, const float *wp, const float *bp, float *out, int B, int C, int H, int W, int Cout) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * Cout * H * W;
    if (g >= total) return;
    const int x = g % W, y = (g / W) % H, co = (g / ((long)W * H)) % Cout, b = g / ((long)W * H * Cout);
    double acc = 0;
    for (int c = 0; c < C; ++c) {
        float o = bd[c];
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 3; ++j) {
                const int iy = y - 1 + k, ix = x - 1 + j;
                const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? in[((long)(b * C + c) * H + iy) * W + ix] : 0.f;
                o = fmaf(v, wd[c * 9 + 3 * k + j], o);
            }
        o = fmaxf(o, 0.f);
        acc += (double)o * (double)wp[(long)c * Cout + co];
    }
    out[g] = fmaxf((float)acc + bp[co], 0.f);
}

static uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static float h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, C = argc > 2 ? atoi(argv[2]) : 128, H = argc > 3 ? atoi(argv[3]) : 40, W = H, Cout = C;
    const long n = (long)B * C * H * W;
    std::vector<float> in(n), wd(C * 9), bd(C), wp((size_t)C * Cout), bp(Cout);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &v : in) v = fmaxf(rnd(), 0.f) * 2.f;
    for (auto &v : wd) v = rnd() * 0.4f;
    for (auto &v : bd) v = rnd() * 0.2f;
    for (auto &v : wp) v = rnd() * 0.15f;
    for (auto &v : bp) v = rnd() * 0.2f;
    std::vector<float> wdp((size_t)C * 10, 0.f);
    for (int c = 0; c < C; ++c) {
        for (int t = 0; t < 9; ++t) wdp[(size_t)(c / 2) * 20 + 2 * t + (c & 1)] = wd[c * 9 + t];
        wdp[(size_t)(c / 2) * 20 + 18 + (c & 1)] = bd[c];
    }
    const int ng = C / 16, ncb = Cout / 32;
    std::vector<uint16_t> wpf((size_t)C * Cout * 2);
    for (int g = 0; g < ng; ++g)
        for (int cb = 0; cb < ncb; ++cb)
            for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 8; ++j) {
                    const float x = wp[(size_t)(16 * g + 8 * (ln >> 5) + j) * Cout + cb * 32 + (ln & 31)];
                    const uint16_t hi = f2h(x);
                    wpf[((((size_t)g * ncb + cb) * 2 + 0) * 64 + ln) * 8 + j] = hi;
                    wpf[((((size_t)g * ncb + cb) * 2 + 1) * 64 + ln) * 8 + j] = f2h(x - h2f(hi));
                }
    float *d_in, *d_out, *d_ref, *d_wd, *d_bd, *d_wp, *d_bp, *d_wdp;
    half_t *d_wpf;
    CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_out, n * 4)); CK(hipMalloc(&d_ref, n * 4));
    CK(hipMalloc(&d_wd, wd.size() * 4)); CK(hipMalloc(&d_bd, bd.size() * 4)); CK(hipMalloc(&d_wp, wp.size() * 4)); CK(hipMalloc(&d_bp, bp.size() * 4));
    CK(hipMalloc(&d_wdp, wdp.size() * 4)); CK(hipMalloc(&d_wpf, wpf.size() * 2));
    CK(hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wd, wd.data(), wd.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_bd, bd.data(), bd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_bp, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wdp, wdp.data(), wdp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wpf, wpf.data(), wpf.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0xff, n * 4));
    DwPwArgs a{};
    a.in = d_in; a.out = d_out; a.wd = d_wd; a.bd = d_bd; a.wp = d_wp; a.bp = d_bp; a.B = B; a.Cin = C; a.H = H; a.W = W; a.Cout = Cout; a.Ho = H; a.Wo = W;
    a.stride = 1; a.relu = 1; a.wdp = d_wdp; a.wpf = d_wpf;
    float *d_zero; CK(hipMalloc(&d_zero, dwpw_wave_zero_bytes())); CK(hipMemset(d_zero, 0, dwpw_wave_zero_bytes()));
    a.zeros = d_zero;
    long long *d_st; CK(hipMalloc(&d_st, 64 * 8)); CK(hipMemset(d_st, 0, 64 * 8));
    a.tmp = reinterpret_cast<float *>(d_st);
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_in, d_wd, d_bd, d_wp, d_bp, d_ref, B, C, H, W, Cout);
    if (!launch_dwpw_wave(a, 0)) { printf("shape not covered\n"); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> out(n), ref(n);
    CK(hipMemcpy(out.data(), d_out, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ref.data(), d_ref, n * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxr = 0; long bad = 0, first = -1;
    for (long i = 0; i < n; ++i) {
        const double d = fabs((double)out[i] - ref[i]);
        if (!(d <= 1e-4 * (1 + fabs(ref[i])))) { if (first < 0) first = i; ++bad; }
        if (d > maxd) maxd = d;
        if (fabs(ref[i]) > maxr) maxr = fabs(ref[i]);
    }
    printf("B=%d max |d| %.3g (max |ref| %.3g), %ld of %ld outside 1e-4", B, maxd, maxr, bad, n);
    if (first >= 0) printf("; first at b=%ld co=%ld y=%ld x=%ld: got %g want %g", first / ((long)Cout * H * W), (first / (H * W)) % Cout, (first / W) % H, first % W, out[first], ref[first]);
    printf("\n");
    if (bad) {
        long byl[64] = {0}, byc[256] = {0};
        for (long i = 0; i < n; ++i) {
            const double d = fabs((double)out[i] - ref[i]);
            if (!(d <= 1e-4 * (1 + fabs(ref[i])))) { ++byl[(i % (H * W)) % 64]; ++byc[(i / (H * W)) % Cout]; }
        }
        printf("bad by lane (pixel %% 64):");
        for (int i = 0; i < 64; ++i) printf(" %ld", byl[i]);
        printf("\nbad by output channel:");
        for (int i = 0; i < Cout; ++i) printf(" %ld", byc[i]);
        printf("\n");
    }
#ifdef WAVE_STAMP
    {
        launch_dwpw_wave(a, 0); CK(hipDeviceSynchronize());
        launch_dwpw_wave(a, 0); CK(hipDeviceSynchronize());
        long long st[64]; CK(hipMemcpy(st, d_st, sizeof st, hipMemcpyDeviceToHost));
        printf("stamps (100 MHz ticks x 10 ns) from kernel start: zero-fill done %lld, prologue issued %lld, chunk 0 landed %lld, K loop done %lld, end %lld\n",
               st[1] - st[0], st[2] - st[0], st[3] - st[0], st[4] - st[0], st[5] - st[0]);
        printf("top of step s (cycles from kernel start):");
        for (int c = 0; c < 32; ++c) printf(" %lld", st[7 + c] - st[0]);
        printf("\n");
    }
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch_dwpw_wave(a, 0);
    CK(hipEventRecord(e0, 0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) launch_dwpw_wave(a, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dwpw_wave_kernel: %.2f us per launch (B=%d: %.1f MB in + out -> %.2f TB/s)\n", ms * 1000 / reps, B, 2.0 * n * 4 / 1e6, 2.0 * n * 4 / (ms / reps * 1e-3) / 1e12);
    return 0;
}
