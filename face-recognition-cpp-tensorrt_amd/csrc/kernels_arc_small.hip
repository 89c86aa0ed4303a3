// ArcFace IR-50 / IR-SE-50 at SMALL batches: the 3x3 convolutions of one synchronous /inference call (1 - 8 faces: src/app.cpp:304-310 calls
// forward() with the faces of ONE frame), fp16 NHWC, fp32 accumulation on v_mfma_f32_32x32x16_f16 (model_irse.py:48-66).
//
// Why the strip kernels (kernels_arc.hip, kernels_arc_s2.hip) are slow here: their workgroup owns 128 output channels of a strip, so it
// streams the weights of 128 couts - 590 KB for a 256 -> 256 layer - however few pixels it has; with 4 faces a 14x14 layer is 56 such
// workgroups, each pulling 590 KB through ONE CU's vector memory path (~ 90 GB/s) for 1 us of MFMAs: 8.6 - 12.4 us per launch, ~ 150 such
// launches in a call.  Here the unit of work is the smallest the matrix core allows - ONE 32-cout block x ONE 32-pixel tile per workgroup
// (pixels flattened over faces and rows, so 4 faces of 14x14 are 24.5 tiles, not 28) - and the K loop (Cin x 9 taps) is split over the
// four waves of the workgroup (two cout blocks per workgroup once a layer has more units than the chip has CUs): a 256 -> 256 layer for 4
// faces is 200 workgroups of 147 KB of weights each, every wave runs 36 MFMAs on
// operands it loads straight from global memory / L2 into registers (no LDS staging: a tile's patch is read once, by one workgroup), and
// the four partial accumulators meet in LDS in a fixed order (deterministic).  The 1x1 stride-2 shortcut convolution of a unit's first
// block rides along as extra K steps into a second accumulator (its own BatchNorm), exactly as in the stride-2 strip kernel.
// Weights come in the fragment order the strip kernels already use (one contiguous KB per MFMA and wave).
#include <cstdlib>
#include <type_traits>

#include "frt_kernels.h"

namespace {

constexpr int EROW = 36;  // floats per pixel row of the reduction tile (32 + 4 pad)

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// NW waves split the K loop; the workgroup owns NC 32-cout blocks of one 32-pixel tile (a pixel fragment then feeds NC MFMAs); D = depth of
// the operand register ring in (chunk, tap) pairs (one pair = the operands of 4 MFMAs per cout block).
template <bool SCF, int NW, int NC, int D>
__global__ __launch_bounds__(NW * 64) void conv_small_kernel(ConvMfmaArgs p, const half_t *wfrag, unsigned long long tap_pack, int M) {
    __shared__ __attribute__((aligned(16))) float red[SCF ? 2 : 1][NW][32][EROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int tile = blockIdx.x, cb0 = blockIdx.y * NC;
    const int H = p.H, W = p.W, Cin = p.Cin, HoWo = p.Ho * p.Wo;
    const int nch = Cin >> 6, np = nch * 9, nsc = SCF ? p.Csc >> 6 : 0;
    const int total = np + nsc;
    const int cnt = total > wave ? (total - wave + NW - 1) / NW : 0;  // this wave's pairs: wave, wave + NW, ...

    // the pixel this lane feeds into the B fragments
    const int m = tile * 32 + r;
    const bool valid = m < M;
    const int f = valid ? m / HoWo : 0, rem = valid ? m - f * HoWo : 0;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - 1, ix0 = ox * p.stride - 1;
    const half_t *xf = p.x + (long)f * H * W * Cin + hi * 8;
    const half_t *scxf = SCF ? p.scx + ((long)(f * H + oy * 2) * W + ox * 2) * p.Csc + hi * 8 : nullptr;
    const long wstride = (long)nch * 9 * 2048, wsc_stride = (long)nsc * 2048;  // halfs per cout block
    const half_t *wb = wfrag + (long)cb0 * wstride + lane * 8;
    const half_t *wscb = SCF ? p.wscf + (long)cb0 * wsc_stride + lane * 8 : nullptr;

    half8 A[D][NC][4], B[D][4];
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef FRT_ABLATE
    const int abl = (int)(tap_pack >> 40) & 3;  // timing experiments (wrong results): 1 no B loads, 2 no A loads
#endif
    auto load = [&](int i, auto dc) {
        constexpr int d = decltype(dc)::value;
        const int j = wave + NW * i;
#ifdef FRT_ABLATE
        if (abl) {
            const half_t *ap = wb + (long)((j / 9) * 9 + j % 9) * 2048;
            const half_t *bp = xf + (long)(j % 9) * Cin + (j / 9) * 64;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) A[d][c][kk] = abl == 2 ? zero8 + (half_t)lane : *reinterpret_cast<const half8 *>(ap + c * wstride + kk * 512);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) B[d][kk] = abl == 1 ? zero8 + (half_t)lane : *reinterpret_cast<const half8 *>(bp + kk * 16);
            return;
        }
#endif
        if (SCF && j >= np) {
            const int c64 = j - np;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) A[d][c][kk] = *reinterpret_cast<const half8 *>(wscb + c * wsc_stride + (long)c64 * 2048 + kk * 512);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) B[d][kk] = valid ? *reinterpret_cast<const half8 *>(scxf + c64 * 64 + kk * 16) : zero8;
        } else {
            const int c64 = j / 9, st = j - c64 * 9;
            const int tap = (int)(tap_pack >> (4 * st)) & 15;
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) A[d][c][kk] = *reinterpret_cast<const half8 *>(wb + c * wstride + (long)(c64 * 9 + st) * 2048 + kk * 512);
            const int iy = iy0 + dy, ix = ix0 + dx;
            const bool inb = valid && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const half_t *bp = xf + ((long)iy * W + ix) * Cin + c64 * 64;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) B[d][kk] = inb ? *reinterpret_cast<const half8 *>(bp + kk * 16) : zero8;
        }
    };

    // the epilogue's operands (thread = output pixel px, couts 4q..4q+3 of every cout block) are requested now: they land under the K loop
    const int px = (tid & 255) >> 3, q = tid & 7;
    const int mo = tile * 32 + px;
    floatx4 q0[NC], q1[NC], q2[NC], q3[NC], q4[NC], q5[NC];
    half4 sc4[NC];
    if (tid < 256) {
        const int mc = mo < M ? mo : 0;
        const int fo = mc / HoWo, ro = mc - fo * HoWo, yo = ro / p.Wo, xo = ro - yo * p.Wo;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int cch = (cb0 + c) * 32 + 4 * q;
            q0[c] = *reinterpret_cast<const floatx4 *>(p.p0 + cch);
            if (p.mode != EPI_PRELU) q1[c] = *reinterpret_cast<const floatx4 *>(p.p1 + cch);
            if (p.mode == EPI_BN_ADD_BN) {
                if (p.out1) {
                    q2[c] = *reinterpret_cast<const floatx4 *>(p.p2 + cch);
                    q3[c] = *reinterpret_cast<const floatx4 *>(p.p3 + cch);
                }
                if constexpr (SCF) {
                    q4[c] = *reinterpret_cast<const floatx4 *>(p.psc0 + cch);
                    q5[c] = *reinterpret_cast<const floatx4 *>(p.psc1 + cch);
                } else {
                    sc4[c] = *reinterpret_cast<const half4 *>(p.sc + ((long)(fo * p.sc_h + yo * p.sc_stride) * p.sc_w + xo * p.sc_stride) * p.Cout + cch);
                }
            }
        }
    }

    floatx16 acc[NC], acc_sc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f, acc_sc[c][e] = 0.f;

    static_for<D>([&](auto dc) {
        if (decltype(dc)::value < cnt) load(decltype(dc)::value, dc);
    });
    for (int i0 = 0; i0 < cnt; i0 += D) {
        static_for<D>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            const int i = i0 + d;
            if (i >= cnt) return;
            if (SCF && wave + NW * i >= np) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc_sc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[d][c][kk], B[d][kk], acc_sc[c], 0, 0, 0);
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[d][c][kk], B[d][kk], acc[c], 0, 0, 0);
            }
            if (i + D < cnt) load(i + D, dc);
        });
    }

    // ---- the waves' partial sums meet in LDS (lane owns pixel r, couts 8g + 4hi + j), summed in wave order; cout block by cout block
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<floatx4 *>(&red[0][wave][r][8 * g + 4 * hi]) = floatx4{acc[c][4 * g], acc[c][4 * g + 1], acc[c][4 * g + 2], acc[c][4 * g + 3]};
            if constexpr (SCF)
                *reinterpret_cast<floatx4 *>(&red[1][wave][r][8 * g + 4 * hi]) =
                    floatx4{acc_sc[c][4 * g], acc_sc[c][4 * g + 1], acc_sc[c][4 * g + 2], acc_sc[c][4 * g + 3]};
        }
        __syncthreads();
        if (tid >= 256 || mo >= M) continue;
        const int cch = (cb0 + c) * 32 + 4 * q;
        floatx4 v = *reinterpret_cast<const floatx4 *>(&red[0][0][px][4 * q]);
#pragma unroll
        for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const floatx4 *>(&red[0][w][px][4 * q]);
        if (p.mode == EPI_PRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * q0[c][e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * q0[c][e] + q1[c][e];
        }
        if (p.mode == EPI_BN_ADD_BN) {
            if constexpr (SCF) {
                floatx4 sacc = *reinterpret_cast<const floatx4 *>(&red[1][0][px][4 * q]);
#pragma unroll
                for (int w = 1; w < NW; ++w) sacc += *reinterpret_cast<const floatx4 *>(&red[1][w][px][4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += sacc[e] * q4[c][e] + q5[c][e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)sc4[c][e];
            }
        }
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
        *reinterpret_cast<half4 *>(p.out0 + (long)mo * p.Cout + cch) = o;
        if (p.mode == EPI_BN_ADD_BN && p.out1) {
            half4 z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = (half_t)(v[e] * q2[c][e] + q3[c][e]);
            *reinterpret_cast<half4 *>(p.out1 + (long)mo * p.Cout + cch) = z;
        }
    }
}

// How much work may go this way: workgroups x (chunk, tap) pairs of a launch; beyond it the strip kernels' weight reuse wins.  Measured per
// layer at 1 / 4 / 8 / 16 / 32 faces (profiles/r03/r03p_small_layers.txt, r03p_small_ab*.txt): stride-1 layers break even at ~ 14 000 (14x14x256
// for 8 faces: 392 x 36; 28x28x128 and 56x56x64 for 4 faces are level from 7 000 on); the stride-2 layers - 33 - 44 us in the strip
// kernel however small the batch - win at 30 000 (16 faces: 25 - 27 us) and lose at 60 000 (32 faces, 14 -> 7: 47 against 43).
long small_work_limit(int stride) {
    static const long lim = [] {
        const char *e = frt_tuning_env("FRT_CONV_SMALL_WORK");
        return e ? atol(e) : -1;
    }();
    if (lim >= 0) return lim;
    return stride == 2 ? 40000 : 16000;
}

}  // namespace

// 3x3 / pad 1 / stride 1 or 2 with fragment-ordered weights, Cin % 64 == 0, Cout % 32 == 0, the unit epilogues without the SE tail
// (PReLU; BN; BN + shortcut tensor or fused 1x1 stride-2 shortcut conv + next BN), few enough output pixels.
bool conv_small_applies(const ConvMfmaArgs &a) {
    if (a.ks != 3 || a.pad != 1 || (a.stride != 1 && a.stride != 2) || a.Cin % 64 || a.Cout % 32 || a.splits != 1) return false;
    if (!(a.stride == 1 ? a.wf : a.wf2)) return false;
    if (a.Ho != a.H / a.stride || a.Wo != a.W / a.stride) return false;
    if (a.mode != EPI_PRELU && a.mode != EPI_BN && a.mode != EPI_BN_ADD_BN) return false;
    if (a.mode == EPI_BN_ADD_BN) {
        if (a.scx) {
            if (!(a.stride == 2 && a.wscf && a.psc0 && a.psc1 && a.Csc % 64 == 0)) return false;
        } else if (!a.sc) {
            return false;
        }
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long pairs = a.Cin / 64 * 9 + (a.mode == EPI_BN_ADD_BN && a.scx ? a.Csc / 64 : 0);
    return (M + 31) / 32 * (a.Cout / 32) * pairs <= small_work_limit(a.stride);
}

template <bool SCF, int NW, int NC, int D>
void launch_small_t(const ConvMfmaArgs &a, const half_t *wfrag, unsigned long long taps, int M, hipStream_t s) {
    const dim3 grid((M + 31) / 32, a.Cout / (32 * NC));
    hipLaunchKernelGGL((conv_small_kernel<SCF, NW, NC, D>), grid, dim3(NW * 64), 0, s, a, wfrag, taps, M);
}

bool launch_conv_small(const ConvMfmaArgs &a, hipStream_t s) {
    if (!conv_small_applies(a)) return false;
    const int M = a.B * a.Ho * a.Wo;
    // tap of K step `st` in the weight array: natural order, or the stride-2 strip kernel's 0,2,6,8,4,1,7,3,5 (not for its 64 -> 64 form)
    const bool s2_order = a.stride == 2 && !(a.Cin == 64 && a.Cout == 64);
    unsigned long long taps = s2_order ? 0x537148620ull : 0x876543210ull;
#ifdef FRT_ABLATE
    static const int abl = frt_tuning_env("FRT_CONV_SMALL_ABL") ? atoi(frt_tuning_env("FRT_CONV_SMALL_ABL")) : 0;
    taps |= (unsigned long long)(abl & 3) << 40;
#endif
    const half_t *wfrag = a.stride == 1 ? a.wf : a.wf2;
    const bool scf = a.mode == EPI_BN_ADD_BN && a.scx;
    // Four waves, one pixel tile, ring of 3.  (Measured and not kept: 8 waves with every load of the launch in flight at once (ring of 5),
    // and 2 / 4 PIXEL tiles per workgroup sharing the weight fragments (8 waves, rings of 3 / 2) - 1 face 412 / 404 / 720 us per pass,
    // 4 faces 500 / 547 / 800: a workgroup's time is the ~ 300 KB of operands it pulls through its CU's vector memory path, not the number
    // of round trips, and a second pixel tile doubles the gathers.  Also not
    // kept: the pixel operand fetched as whole 128-byte chunks of 8 pixels per load (8 cache lines per instruction instead of the gather's
    // 32) and transposed into MFMA fragments through a wave-private LDS tile, with 1 / 2 / 4 tiles per workgroup: bit-identical results,
    // 430 / 659 us (1 face), 504 / 716 us (4 faces), 16 - 32 faces 1.3 - 3.1 ms per pass (profiles/r03/r03x_small_lds.txt).)
    static const int nc_env = frt_tuning_env("FRT_CONV_SMALL_NC") ? atoi(frt_tuning_env("FRT_CONV_SMALL_NC")) : 0;
    const int wgs = (M + 31) / 32 * (a.Cout / 32);
    int nc = wgs > 256 && a.Cout % 64 == 0 ? 2 : 1;  // more one-block units than CUs: two cout blocks per workgroup share the pixel fragments
    if (nc_env == 1 || (nc_env == 2 && a.Cout % 64 == 0)) nc = nc_env;
    // (four cout blocks per workgroup, ring of 2: 8 / 12 / 16 / 32 faces 0.86 / 1.00 / 1.04 / 1.61 ms per pass against 0.73 / 0.86 / 0.91 / 1.19 -
    //  from ~ 10 faces on the strip kernels win, profiles/r03/r03z_small_nc4.txt)
    if (nc == 2) {
        if (scf) launch_small_t<true, 4, 2, 3>(a, wfrag, taps, M, s);
        else launch_small_t<false, 4, 2, 3>(a, wfrag, taps, M, s);
    } else {
        if (scf) launch_small_t<true, 4, 1, 3>(a, wfrag, taps, M, s);
        else launch_small_t<false, 4, 1, 3>(a, wfrag, taps, M, s);
    }
    return true;
}
