// Stand-alone harness for the detector's split-fp16 3x3 conv kernel (csrc/kernels_det_conv3h.hip) against the round-4 kernel parked in
// tools/experiments/det_conv3h_r04.hip (bit for bit on the 64-channel shapes) and against a plain fp64 direct convolution (tolerance), with
// HIP-event timings per launch.  Shapes = the five launches of one detector pass at 640x640: merge1 (64->64 @ 80x80), merge2 (@ 40x40), the
// fused SSH 64->48 on three levels, the SSH 16->32 and 16->16 convs on three levels.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I face-recognition-cpp-tensorrt_amd/csrc -o det_conv3h_bench tools/ubench/det_conv3h_bench.hip
//   ./det_conv3h_bench [frames = 32]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "frt_kernels.h"
bool det_mfma_enabled() { return true; }
namespace oldk {
#include "../experiments/det_conv3h_r04.hip"
}
namespace newk {
#include "kernels_det_conv3h.hip"
}
namespace wk {
#include "../experiments/det_conv3w_weights_in_registers.hip"
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

__global__ void ref_conv(const float *in, const float *w, const float *b, float *out, int B, int Cin, int H, int W, int Cout, int relu) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * Cout * H * W;
    if (g >= total) return;
    const int x = g % W, y = (g / W) % H, co = (g / ((long)W * H)) % Cout, bb = g / ((long)W * H * Cout);
    double acc = 0;
    for (int c = 0; c < Cin; ++c)
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 3; ++j) {
                const int iy = y - 1 + k, ix = x - 1 + j;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                acc += (double)in[((long)(bb * Cin + c) * H + iy) * W + ix] * (double)w[((long)c * 9 + 3 * k + j) * Cout + co];
            }
    float v = (float)(acc + (double)b[co]);
    out[g] = relu ? fmaxf(v, 0.f) : v;
}

struct Prob {
    int cin, cout, split, n;      // n levels
    int hw[3];
    const char *name;
};

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const Prob probs[] = {{64, 64, 64, 1, {80, 0, 0}, "merge1 64->64 @80"}, {64, 64, 64, 1, {40, 0, 0}, "merge2 64->64 @40"},
                          {64, 48, 32, 3, {80, 40, 20}, "ssh 64->32+16 x3"}, {16, 32, 16, 3, {80, 40, 20}, "ssh 16->16+16 x3"},
                          {16, 16, 16, 3, {80, 40, 20}, "ssh 16->16 x3"}};
    srand(3);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (const Prob &pr : probs) {
        Conv3Args a[3];
        std::vector<float *> outs_new, outs_old, outs_ref, outs2_new, outs2_old;
        std::vector<long> nout, nout2;
        std::vector<std::vector<float>> hw_, hb_;
        std::vector<float *> d_in(3), d_w(3), d_b(3);
        for (int k = 0; k < pr.n; ++k) {
            const int H = pr.hw[k], W = H;
            const long nin = (long)B * pr.cin * H * W;
            std::vector<float> in(nin), w((size_t)pr.cin * 9 * pr.cout), b(pr.cout);
            for (auto &v : in) v = fmaxf(rnd(), -0.2f) * 3.f;
            for (auto &v : w) v = rnd() * 0.1f;
            for (auto &v : b) v = rnd() * 0.3f;
            const int nch = pr.cin / 16;
            std::vector<uint16_t> wh((size_t)nch * 9 * 64 * 32, 0);
            for (int c = 0; c < nch; ++c)
                for (int t = 0; t < 9; ++t)
                    for (int co = 0; co < pr.cout; ++co)
                        for (int kk = 0; kk < 16; ++kk) {
                            const float x = w[((size_t)(c * 16 + kk) * 9 + t) * pr.cout + co];
                            const uint16_t hi = f2h(x);
                            const size_t row = (((size_t)c * 9 + t) * 64 + co) * 32;
                            wh[row + kk] = hi;
                            wh[row + 16 + kk] = f2h(x - h2f(hi));
                        }
            half_t *d_wh;
            CK(hipMalloc(&d_in[k], nin * 4)); CK(hipMalloc(&d_w[k], w.size() * 4)); CK(hipMalloc(&d_b[k], b.size() * 4)); CK(hipMalloc(&d_wh, wh.size() * 2));
            CK(hipMemcpy(d_in[k], in.data(), nin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w[k], w.data(), w.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_b[k], b.data(), b.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
            const long n1 = (long)B * pr.split * H * W, n2 = (long)B * (pr.cout - pr.split) * H * W;
            float *o1n, *o1o, *o2n = nullptr, *o2o = nullptr, *oref;
            CK(hipMalloc(&o1n, n1 * 4)); CK(hipMalloc(&o1o, n1 * 4)); CK(hipMalloc(&oref, (n1 + n2) * 4));
            CK(hipMemset(o1n, 0xff, n1 * 4)); CK(hipMemset(o1o, 0xff, n1 * 4));
            if (n2) { CK(hipMalloc(&o2n, n2 * 4)); CK(hipMalloc(&o2o, n2 * 4)); CK(hipMemset(o2n, 0xff, n2 * 4)); CK(hipMemset(o2o, 0xff, n2 * 4)); }
            Conv3Args c{};
            c.in = d_in[k]; c.out = o1n; c.w = d_w[k]; c.b = d_b[k]; c.B = B; c.Cin = pr.cin; c.H = H; c.W = W; c.Cout = pr.cout; c.Ho = H; c.Wo = W; c.stride = 1; c.relu = 1;
            c.out_ctotal = pr.split; c.out_coff = 0; c.wh = d_wh;
            if (n2) { c.out2 = o2n; c.split = pr.split; c.out2_ctotal = pr.cout - pr.split; c.out2_coff = 0; }
            a[k] = c;
            outs_new.push_back(o1n); outs_old.push_back(o1o); outs2_new.push_back(o2n); outs2_old.push_back(o2o); outs_ref.push_back(oref);
            nout.push_back(n1); nout2.push_back(n2);
            const long nt = n1 + n2;
            hipLaunchKernelGGL(ref_conv, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, 0, d_in[k], d_w[k], d_b[k], oref, B, pr.cin, H, W, pr.cout, 1);
        }
        const bool use_w = wk::launch_conv3x3_splitw(a, pr.n, 0);   // weights-in-registers kernel where it applies (64 input channels)
        if (!use_w && !newk::launch_conv3x3_split(a, pr.n, 0)) { printf("%s: new kernel does not cover the shape\n", pr.name); continue; }
        Conv3Args ao[3];
        for (int k = 0; k < pr.n; ++k) { ao[k] = a[k]; ao[k].out = outs_old[k]; if (ao[k].out2) ao[k].out2 = outs2_old[k]; }
        const bool have_old = pr.cin == 64 && oldk::launch_conv3x3_split(ao, pr.n, 0);
        CK(hipDeviceSynchronize());
        long diff_bits = 0, bad = 0, tot = 0;
        double maxd = 0;
        for (int k = 0; k < pr.n; ++k) {
            const int H = pr.hw[k], W = H;
            const long hw = (long)H * W;
            std::vector<float> n1(nout[k]), o1(nout[k]), n2(nout2[k]), o2(nout2[k]), ref(nout[k] + nout2[k]);
            CK(hipMemcpy(n1.data(), outs_new[k], nout[k] * 4, hipMemcpyDeviceToHost));
            if (have_old) CK(hipMemcpy(o1.data(), outs_old[k], nout[k] * 4, hipMemcpyDeviceToHost));
            if (nout2[k]) { CK(hipMemcpy(n2.data(), outs2_new[k], nout2[k] * 4, hipMemcpyDeviceToHost)); if (have_old) CK(hipMemcpy(o2.data(), outs2_old[k], nout2[k] * 4, hipMemcpyDeviceToHost)); }
            CK(hipMemcpy(ref.data(), outs_ref[k], ref.size() * 4, hipMemcpyDeviceToHost));
            if (have_old) {
                diff_bits += memcmp(n1.data(), o1.data(), nout[k] * 4) != 0;
                if (nout2[k]) diff_bits += memcmp(n2.data(), o2.data(), nout2[k] * 4) != 0;
            }
            for (int b = 0; b < B; ++b)
                for (int co = 0; co < pr.cout; ++co)
                    for (long p = 0; p < hw; ++p) {
                        const float got = co < pr.split ? n1[((long)b * pr.split + co) * hw + p] : n2[((long)b * (pr.cout - pr.split) + co - pr.split) * hw + p];
                        const float want = ref[((long)b * pr.cout + co) * hw + p];
                        const double d = fabs((double)got - want);
                        if (!(d <= 2e-5 * (1 + fabs(want)))) ++bad;
                        if (d > maxd) maxd = d;
                        ++tot;
                    }
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](auto f) {
            for (int i = 0; i < 5; ++i) f();
            CK(hipEventRecord(e0, 0));
            const int reps = 40;
            for (int i = 0; i < reps; ++i) f();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000 / reps;
        };
        const float t_new = timeit([&] { if (use_w) wk::launch_conv3x3_splitw(a, pr.n, 0); else newk::launch_conv3x3_split(a, pr.n, 0); });
        const float t_h = use_w ? timeit([&] { newk::launch_conv3x3_split(a, pr.n, 0); }) : 0.f;
        const float t_old = have_old ? timeit([&] { oldk::launch_conv3x3_split(ao, pr.n, 0); }) : 0.f;
        printf("%-20s B=%d  %s %7.2f us  conv3h %7.2f us  r04 %7.2f us   vs fp64: max |d| %.3g, %ld of %ld outside 2e-5   vs r04: %s\n", pr.name, B, use_w ? "conv3w" : "conv3h", t_new, t_h, t_old, maxd, bad, tot,
               have_old ? (diff_bits ? "DIFFERENT BITS" : "bit-identical") : "n/a");
        for (int k = 0; k < pr.n; ++k) { hipFree(d_in[k]); hipFree(d_w[k]); hipFree(d_b[k]); hipFree(outs_new[k]); hipFree(outs_old[k]); hipFree(outs_ref[k]); if (outs2_new[k]) { hipFree(outs2_new[k]); hipFree(outs2_old[k]); } }
    }
    return 0;
}
