// PARKED EXPERIMENT (round 3) - not part of libfrt.so.  Winograd F(2x2, 3x3) for the 26 256 -> 256 convolutions at 14x14, built, parity-correct on
// the first run (128- and 64-face passes against the fp32 oracle: 1 - cos 5.2e-6 where the direct kernels give 2.9e-6, IR-SE 2.0e-6 / 1.7e-6),
// and SLOWER than the direct strip kernel: 68.7 us per launch against 41.6 (profiles/r03/r03z_wino_first.txt, r03z_wino_ablations.txt).  Where the
// time goes (ablations, us per launch): everything 68.7; without the input transform 62.4; without the output fold 61.0; without the MFMAs and
// their fragment reads 52.0; without the weight loads 65.8; with transform, fold and MFMAs all off 35.7 - the skeleton alone (1 MB of
// transformed weights per workgroup through one CU's memory path ~ 13 us, two patch loads, 32 barriers, an epilogue of scattered 8-byte
// stores) costs what the direct kernel's whole K loop costs, and at 103 KB of LDS there is one workgroup per CU and one wave per SIMD, so
// nothing overlaps the VALU work of transform and fold with another wave's MFMAs.  Halving the MFMAs (512 instead of 1 008 per wave) buys ~ 13 us
// of the direct kernel's 27 us K loop; transform + fold + the 16/9 weight bytes cost more.  See DESIGN 3.15.
// To rebuild: add kernels_arc_wino to csrc/Makefile's NAMES, declare conv_wino_applies / launch_conv_wino and ConvMfmaArgs::wu in
// frt_kernels.h, call them from launch_conv_mfma, and give frt_api.cpp the host-side weight transform at the end of this file.
//
// ArcFace IR-50: the 256 -> 256 3x3 stride-1 convolutions at 14x14 (26 of the 48 convs; model_irse.py:58-66 in the third stage) as
// Winograd F(2x2, 3x3): 16 multiplies per 2x2 output tile and channel pair instead of 36 - under the package power limit the pipelined
// step is bound by matrix-core cycles, and this is the one form that removes them.
//
//   Y = A^T [ sum_c (G w G^T)_c . (B^T d B)_c ] A      per 2x2 output tile, 4x4 input window d, 3x3 filter w
//
// * weights U = G w G^T come pre-transformed from the host (fp32 arithmetic, rounded to fp16 once), one K-slab per position p = (a, b),
//   fragment-ordered like the strip kernels' weights: [16 positions][Cout/32][Cin/16][64 lanes][8 halfs];
// * a workgroup = one image x 128 output channels (4 waves = 4 cout blocks), the image's 49 tiles in 64 tile slots (two MFMA column
//   tiles); the raw zero-padded 16x16 input patch of HALF the input channels sits in LDS (68 KB), the channels go in two halves;
// * positions are the OUTER loop: V_p = (B^T d B)_p for all 64 slots x 128 channels is computed once per position by all threads
//   (4 patch reads, 3 packed fp16 adds per 8 channels) into a double-buffered 17 KB LDS slice, every wave accumulates
//   M_p = U_p . V_p over the channel half (16 MFMAs) and folds it into its four output accumulators with the +-1 coefficients of
//   A^T M A - so a wave holds 2 x 4 output accumulators, not 2 x 16 position accumulators;
// * epilogue straight from the accumulators (lane = tile, 16 couts): PReLU, or BN + shortcut + next BN, as in the strip kernels.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "frt_kernels.h"

namespace {

constexpr int PPITCH = 272;            // bytes per pixel of the raw patch (128 channels + 16 pad)
constexpr int VPITCH = 272;            // bytes per tile slot of a transformed slice
constexpr int PATCH_B = 256 * PPITCH;  // 16 x 16 padded pixels
constexpr int V_B = 64 * VPITCH;

template <class F, int... I>
__device__ __forceinline__ void wfor_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wfor(F &&f) {
    wfor_impl(f, std::make_integer_sequence<int, N>{});
}

// B^T rows of F(2x2, 3x3): [1 0 -1 0], [0 1 1 0], [0 -1 1 0], [0 1 0 -1] -> (first index, second index, sign of the second)
__device__ __forceinline__ constexpr int bt_i1(int a) { return a == 0 ? 0 : (a == 2 ? 2 : 1); }
__device__ __forceinline__ constexpr int bt_i2(int a) { return a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3)); }
__device__ __forceinline__ constexpr int bt_s2(int a) { return a == 1 ? 1 : -1; }
// A^T rows: [1 1 1 0], [0 1 -1 -1]
__device__ __forceinline__ constexpr int at0(int a) { return a <= 2 ? 1 : 0; }
__device__ __forceinline__ constexpr int at1(int a) { return a == 0 ? 0 : (a == 1 ? 1 : -1); }

__global__ __launch_bounds__(256, 1) void conv_wino14_kernel(ConvMfmaArgs p, const half_t *U, int abl) {  // abl (tuning builds, wrong results): 1 no transform, 2 no fold, 4 no MFMA, 8 no weight loads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *patch = smem;
    char *vbuf = smem + PATCH_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int f = blockIdx.x >> 1, cb = (blockIdx.x & 1) * 4 + wave;  // image, 32-cout block
    constexpr int HW = 14, C = 256;
    const half_t *xf = p.x + (long)f * HW * HW * C;

    // transform items of this thread: slot = 16 i + tid / 16 (i = 0..3), 8-channel group g = tid % 16
    const int g16 = tid & 15;
    int pix0[4];  // byte offset of the window's top-left padded pixel, or -1 for a dead slot
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = 16 * i + (tid >> 4);
        const int ty = slot / 7, tx = slot - ty * 7;
        pix0[i] = slot < 49 ? ((2 * ty) * 16 + 2 * tx) * PPITCH + g16 * 16 : -1;
    }

    floatx16 Y[2][4], M[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[n][o][e] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) M[n][e] = 0.f;
    }

    const half_t *ub = U + ((long)cb * 16) * 512 + lane * 8;  // + (p * 8 cout blocks * 16 k-steps + k16) * 512
    auto ufrag = [&](int pos, int k16) { return *reinterpret_cast<const half8 *>(ub + ((long)pos * 8 * 16 + k16) * 512); };

    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();  // every wave is done with the first half's patch and slices
        // ---- raw patch of channel half h: 256 padded pixels x 16 pieces of 16 bytes
        {
            half8 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int id = i * 256 + tid, pixel = id >> 4, piece = id & 15;
                const int iy = (pixel >> 4) - 1, ix = (pixel & 15) - 1;
                const bool in = iy >= 0 && iy < HW && ix >= 0 && ix < HW;
                v[i] = in ? *reinterpret_cast<const half8 *>(xf + ((long)iy * HW + ix) * C + h * 128 + piece * 8) : half8{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int id = i * 256 + tid, pixel = id >> 4, piece = id & 15;
                *reinterpret_cast<half8 *>(patch + pixel * PPITCH + piece * 16) = v[i];
            }
        }
        __syncthreads();

        // position p = 4 a + b: a (the row combination) is a run-time loop, b (the column combination) is unrolled
        auto transform = [&](int a, auto bc) {
            constexpr int b = decltype(bc)::value;
            const int i1 = a == 0 ? 0 : (a == 2 ? 2 : 1), i2 = a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
            const half_t sa = a == 1 ? (half_t)1.f : (half_t)-1.f;
            const half8 sa8 = {sa, sa, sa, sa, sa, sa, sa, sa};
            const int r1 = i1 * 16 * PPITCH, r2 = i2 * 16 * PPITCH;
            constexpr int c1 = bt_i1(b) * PPITCH, c2 = bt_i2(b) * PPITCH;
            char *dst = vbuf + (b & 1) * V_B + (tid >> 4) * VPITCH + g16 * 16;  // (4 a + b) & 1 == b & 1
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (pix0[i] < 0) continue;
                const char *s = patch + pix0[i];
                const half8 d11 = *reinterpret_cast<const half8 *>(s + r1 + c1), d12 = *reinterpret_cast<const half8 *>(s + r1 + c2);
                const half8 d21 = *reinterpret_cast<const half8 *>(s + r2 + c1), d22 = *reinterpret_cast<const half8 *>(s + r2 + c2);
                const half8 t1 = bt_s2(b) > 0 ? d11 + d12 : d11 - d12;
                const half8 t2 = bt_s2(b) > 0 ? d21 + d22 : d21 - d22;
                *reinterpret_cast<half8 *>(dst + i * 16 * VPITCH) = t1 + sa8 * t2;
            }
        };

        half8 A[2][8];  // weight fragments of a position are requested one whole position ahead
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) A[0][kk] = ufrag(0, h * 8 + kk);
        transform(0, std::integral_constant<int, 0>{});
        __syncthreads();
#pragma unroll 1
        for (int a = 0; a < 4; ++a) {
            const float ci0 = a <= 2 ? 1.f : 0.f, ci1 = a == 0 ? 0.f : (a == 1 ? 1.f : -1.f);  // A^T[0][a], A^T[1][a]
            wfor<4>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                const int P = 4 * a + b;
                if (P + 1 < 16 && !(abl & 8)) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) A[(b + 1) & 1][kk] = ufrag(P + 1, h * 8 + kk);
                }
                if (!(abl & 1)) {
                    if constexpr (b < 3) transform(a, std::integral_constant<int, b + 1>{});
                    else if (a < 3) transform(a + 1, std::integral_constant<int, 0>{});
                }
                const char *vb = vbuf + (b & 1) * V_B + r * VPITCH + hi * 16;
                if (!(abl & 4)) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            M[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[b & 1][kk], *reinterpret_cast<const half8 *>(vb + n * 32 * VPITCH + kk * 32), M[n], 0, 0, 0);
                }
                // fold M_p into the four outputs of the tile: Y_ij += At[i][a] At[j][b] M_ab
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (abl & 2) break;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        constexpr int cj0 = at0(b), cj1 = at1(b);
                        const int cj = j ? cj1 : cj0;
                        if (cj == 0) continue;
                        const float w0 = ci0 * (float)cj, w1 = ci1 * (float)cj;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            Y[n][j][e] = fmaf(w0, M[n][e], Y[n][j][e]);
                            Y[n][2 + j][e] = fmaf(w1, M[n][e], Y[n][2 + j][e]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) M[n][e] = 0.f;
                }
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }

    // ---- epilogue: lane = tile slot n * 32 + r, couts cb * 32 + 8 g + 4 hi + e
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int cch = cb * 32 + 8 * g + 4 * hi;
        const floatx4 q0 = *reinterpret_cast<const floatx4 *>(p.p0 + cch);
        floatx4 q1 = {0, 0, 0, 0}, q2 = q1, q3 = q1;
        if (p.mode != EPI_PRELU) q1 = *reinterpret_cast<const floatx4 *>(p.p1 + cch);
        const bool two = p.mode == EPI_BN_ADD_BN && p.out1;
        if (two) {
            q2 = *reinterpret_cast<const floatx4 *>(p.p2 + cch);
            q3 = *reinterpret_cast<const floatx4 *>(p.p3 + cch);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int slot = n * 32 + r;
            if (slot >= 49) continue;
            const int ty = slot / 7, tx = slot - ty * 7;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const long m = ((long)f * HW + 2 * ty + (o >> 1)) * HW + 2 * tx + (o & 1);
                float v[4] = {Y[n][o][4 * g], Y[n][o][4 * g + 1], Y[n][o][4 * g + 2], Y[n][o][4 * g + 3]};
                if (p.mode == EPI_PRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * q0[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * q0[e] + q1[e];
                }
                if (p.mode == EPI_BN_ADD_BN) {
                    const half4 sc = *reinterpret_cast<const half4 *>(p.sc + m * C + cch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)sc[e];
                }
                half4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (half_t)v[e];
                *reinterpret_cast<half4 *>(p.out0 + m * C + cch) = ov;
                if (two) {
                    half4 z;
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[e] = (half_t)(v[e] * q2[e] + q3[e]);
                    *reinterpret_cast<half4 *>(p.out1 + m * C + cch) = z;
                }
            }
        }
    }
}

}  // namespace

// 256 -> 256, 14x14, 3x3 / stride 1 / pad 1, transformed weights present, PReLU or BN (+ shortcut tensor of the output's geometry + next BN)
bool conv_wino_applies(const ConvMfmaArgs &a, const half_t *U) {
    if (!U || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Cin != 256 || a.Cout != 256 || a.H != 14 || a.W != 14 || a.splits != 1) return false;
    if (a.mode != EPI_PRELU && a.mode != EPI_BN && a.mode != EPI_BN_ADD_BN) return false;
    if (a.mode == EPI_BN_ADD_BN && !(a.sc && !a.scx && a.sc_stride == 1 && a.sc_h == 14 && a.sc_w == 14)) return false;
    return true;
}

void launch_conv_wino(const ConvMfmaArgs &a, const half_t *U, hipStream_t s) {
    const size_t lds = PATCH_B + 2 * V_B;
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino14_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static const int abl = frt_tuning_env("FRT_CONV_WINO_ABL") ? atoi(frt_tuning_env("FRT_CONV_WINO_ABL")) : 0;
    hipLaunchKernelGGL(conv_wino14_kernel, dim3(a.B * 2), dim3(256), lds, s, a, U, abl);
}

// ---------------------------------------------------------------- host side (was in frt_api.cpp)
#if 0
// Winograd F(2x2, 3x3) weights U = G w G^T (fp32 arithmetic, rounded to fp16 once), one K-slab per position p = 4a + b, in MFMA A-fragment
// order [16][Cout/32][Cin/16][lane = (k half, cout row)][8] (kernels_arc_wino.hip)
std::vector<uint16_t> conv_w_wino_frag(const frt::Blob &b, const std::string &name, int cout, int cin) {
    if (cin % 16 || cout % 32) return {};
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    const float *src = b.get(name, (size_t)cout * cin * 9).data;
    std::vector<uint16_t> w((size_t)16 * cout * cin);
    const int ncb = cout / 32, nk = cin / 16;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float *k = src + ((size_t)co * cin + ci) * 9;
            float t[4][3];
            for (int a = 0; a < 4; ++a)
                for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * k[0 * 3 + j] + G[a][1] * k[1 * 3 + j] + G[a][2] * k[2 * 3 + j];
            const int blk = co >> 5, r = co & 31, k16 = ci >> 4, hi = (ci & 15) >> 3, e = ci & 7;
            for (int a = 0; a < 4; ++a)
                for (int bb = 0; bb < 4; ++bb) {
                    const float u = t[a][0] * G[bb][0] + t[a][1] * G[bb][1] + t[a][2] * G[bb][2];
                    w[(((((size_t)(a * 4 + bb) * ncb + blk) * nk + k16) * 64) + hi * 32 + r) * 8 + e] = frt::f32_to_f16(u);
                }
        }
    return w;
}
#endif
