// PARKED EXPERIMENT (round 4, not in the build): bit-identical (0 of 4.9 M elements differ from the two kernels), but SLOWER - 97 + 80 us (interior +
// ring launch) against 90 + 42 us for dwpw_row4_kernel<32> + dwpw_mfma_kernel at 32 frames.  Unlike the stem (kernels_det_stem.hip: 266 -> 137 us),
// the tensor it removes is small next to the arithmetic it serialises: 1 312 fma per position in PA behind two barrier-separated load
// phases, then one wave of five runs PB; a tile takes ~ 26 us and only two fit a CU (60 KB of LDS).  To build it: add it to the Makefile,
// declare launch_det_stage1b in frt_kernels.h, give DwPwArgs.wdt to blocks with 32 input channels and call it from frt_detector::forward.
//
// Detector body.stage1[3..4] in one kernel (round 4): conv_dw 32 -> 32 at 160x160 (stride 1) + conv_dw 32 -> 64 at stride 2 -> 64 x 80x80
// (net.py:107-108 of the reference's MobileNetV1 body).  Same scheme as kernels_det_stem.hip: a workgroup owns an 8x8 tile of the 80x80
// output and walks both layers over the tile's halo'd regions through LDS, so the 32-channel tensor at 160x160 between them (105 MB written
// and re-read per 32 frames) never exists.
//   PA  conv_dw 32 -> 32 at the 17x17 positions behind the tile; the 19x19 input region comes in two halves of 16 channels
//       ([position][16] in LDS: a tap's channels are four ds_read_b128), every thread keeps its position's 32 pointwise sums across the halves
//                                                                                                    -> LDS b3[17x17][32]
//   PB  conv_dw 32 -> 64 at stride 2 at the 8x8 outputs (wave 0): lane = pixel, depthwise outputs in registers, fp16 hi / lo split, pointwise
//       product on the fp16 matrix cores (three MFMAs per product), B operand by v_permlane32_swap (kernels_det_wave.hip)  -> global
// Arithmetic: dwpw_row4_kernel<32>'s chains for the first block, dwpw_mfma_kernel<2, 2, 32, 1, 2, false, true>'s for the second - bit-identical
// (tuning build: FRT_DET_STEM_CHECK=1 python tools/stem_check_run.py compares this tensor too).
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "frt_kernels.h"

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

struct S1bArgs {
    const float *in;                 // [B][32][H][W]
    const float *wdtA, *wpA, *bpA;   // block A: depthwise tap-major channel pairs [10][16][2] (taps 0-8, bias); pointwise (transposed) [32][32]; bias [32]
    const float *wdtB, *bpB;         // block B: depthwise [10][16][2]; bias [64]
    const half_t *wpfB;              // block B pointwise, fp16 hi/lo split, fragment order [2][2][hi|lo][64][8]
    float *out;                      // [B][64][H/2][W/2]
    int B, H, W;
};

template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T *uni(const T *p) {
    return reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int NP, class P>
__device__ __forceinline__ void ld_pairs(const P w, int off, floatx2 (&dst)[NP]) {  // NP scalar pairs from w[off ...]
#pragma unroll
    for (int c = 0; c < NP; ++c) dst[c] = floatx2{w[off + 2 * c], w[off + 2 * c + 1]};
}

constexpr int RI = 19, RA = 17;   // edge of the input / block A region behind an 8x8 output tile
constexpr int NT = 320;           // five waves: the 289 positions of PA in one pass

template <bool INTERIOR>
__global__ __launch_bounds__(NT) void det_stage1b_kernel(S1bArgs a) {
    __shared__ __attribute__((aligned(16))) float xin[RI * RI][16];
    __shared__ __attribute__((aligned(16))) float b3[RA * RA][32];
    const int tid = threadIdx.x;
    const int H2 = a.H >> 1, W2 = a.W >> 1, tiles_x = W2 >> 3, tiles_y = H2 >> 3;
    int t = blockIdx.x, b, ty, tx;
    if (INTERIOR) {
        const int ix = tiles_x - 2, per = ix * (tiles_y - 2);
        b = t / per;
        t -= b * per;
        ty = t / ix;
        tx = t - ty * ix + 1;
        ty += 1;
    } else {
        const int per = 2 * tiles_x + 2 * (tiles_y - 2);
        b = t / per;
        t -= b * per;
        if (t < tiles_x) { ty = 0; tx = t; }
        else if (t < 2 * tiles_x) { ty = tiles_y - 1; tx = t - tiles_x; }
        else { t -= 2 * tiles_x; ty = 1 + (t >> 1); tx = (t & 1) ? tiles_x - 1 : 0; }
    }
    const int Y0 = ty * 8, X0 = tx * 8;
    const int ray0 = 2 * Y0 - 1, rax0 = 2 * X0 - 1;  // block A region origin
    const int riy0 = ray0 - 1, rix0 = rax0 - 1;      // input region origin
    const long HW = (long)a.H * a.W;
    const float *inb = a.in + (long)b * 32 * HW;

    // this thread's block A position
    const int pa = min(tid, RA * RA - 1), ay = pa / RA, ax = pa - ay * RA;
    const bool a_in = INTERIOR || (ray0 + ay >= 0 && ray0 + ay < a.H && rax0 + ax >= 0 && rax0 + ax < a.W);
    floatx2 acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = floatx2{0.f, 0.f};

    // the input region's loads: thread = position (two passes: idx0 = tid, idx1 = tid + NT), zeros outside the map.  The SECOND half's values are
    // requested before the first half's arithmetic and wait in registers (one exposed memory round trip per tile instead of two)
    int idxs[2];
    bool oks[2];
    const float *srcs[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int idx = min(tid + ps * NT, RI * RI - 1);
        idxs[ps] = tid + ps * NT;
        const int iy = idx / RI, ixx = idx - iy * RI;
        const int gy = riy0 + iy, gx = rix0 + ixx;
        oks[ps] = INTERIOR || (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);
        srcs[ps] = inb + (oks[ps] ? (long)gy * a.W + gx : 0);
    }
    float xv[2][16];
    auto fetch = [&](int half) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            if (idxs[ps] < RI * RI) {
#pragma unroll
                for (int c = 0; c < 16; ++c) xv[ps][c] = srcs[ps][(long)(half * 16 + c) * HW];
            }
    };
    auto stash = [&]() {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            if (idxs[ps] < RI * RI) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<floatx4 *>(&xin[idxs[ps]][4 * q]) =
                        oks[ps] ? floatx4{xv[ps][4 * q], xv[ps][4 * q + 1], xv[ps][4 * q + 2], xv[ps][4 * q + 3]} : floatx4{0.f, 0.f, 0.f, 0.f};
            }
    };
    fetch(0);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();  // every thread is through with the first half's region
        stash();
        __syncthreads();
        if (half == 0) fetch(1);
        // ---- PA on this half: depthwise (channel pairs), ReLU, pointwise partial sums
        {
            int z;
            asm volatile("s_mov_b32 %0, 0" : "=s"(z));  // (keeps the scalar weight loads inside the loop: see kernels_det_stem.hip)
            const auto wdt = uni(a.wdtA) + (z + half * 16), wp = uni(a.wpA) + (z + half * 16 * 32);
            floatx2 dd[8];
            {
                floatx2 wc[2][8];
                ld_pairs<8>(wdt, 9 * 32, dd);
                ld_pairs<8>(wdt, 0, wc[0]);
                static_for<0, 9>([&](auto tc) {
                    constexpr int tt = decltype(tc)::value;
                    if constexpr (tt + 1 < 9) ld_pairs<8>(wdt, (tt + 1) * 32, wc[(tt + 1) & 1]);
                    const float *tp = &xin[(ay + tt / 3) * RI + ax + tt % 3][0];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const floatx4 tq = *reinterpret_cast<const floatx4 *>(tp + 4 * q);
                        dd[2 * q] = __builtin_elementwise_fma(floatx2{tq[0], tq[1]}, wc[tt & 1][2 * q], dd[2 * q]);
                        dd[2 * q + 1] = __builtin_elementwise_fma(floatx2{tq[2], tq[3]}, wc[tt & 1][2 * q + 1], dd[2 * q + 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            {
                floatx2 wc[2][16];
                ld_pairs<16>(wp, 0, wc[0]);
                static_for<0, 16>([&](auto cc) {
                    constexpr int ci = decltype(cc)::value;
                    if constexpr (ci + 1 < 16) ld_pairs<16>(wp, (ci + 1) * 32, wc[(ci + 1) & 1]);
                    const float d = fmaxf(dd[ci >> 1][ci & 1], 0.f);
#pragma unroll
                    for (int c = 0; c < 16; ++c) acc[c] = __builtin_elementwise_fma(floatx2{d, d}, wc[ci & 1][c], acc[c]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
    }
    if (tid < RA * RA) {
        const auto bp = uni(a.bpA);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            floatx4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * q + e;
                o[e] = a_in ? fmaxf(acc[c >> 1][c & 1] + bp[c], 0.f) : 0.f;
            }
            *reinterpret_cast<floatx4 *>(&b3[tid][4 * q]) = o;
        }
    }
    __syncthreads();
    if (tid >= 64) return;

    // ---- PB: conv_dw 32 -> 64 at stride 2; lane = output pixel (py, px) of the tile
    const int lane = tid, py = lane >> 3, px = lane & 7, r = lane & 31, hi = lane >> 5;
    floatx2 d2[16];
    {
        const auto wdt = uni(a.wdtB);
        floatx2 wc[2][16];
        ld_pairs<16>(wdt, 9 * 32, d2);
        ld_pairs<16>(wdt, 0, wc[0]);
        static_for<0, 9>([&](auto tc) {
            constexpr int tt = decltype(tc)::value;
            if constexpr (tt + 1 < 9) ld_pairs<16>(wdt, (tt + 1) * 32, wc[(tt + 1) & 1]);
            const float *tp = &b3[(2 * py + tt / 3) * RA + 2 * px + tt % 3][0];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const floatx4 tq = *reinterpret_cast<const floatx4 *>(tp + 4 * q);
                d2[2 * q] = __builtin_elementwise_fma(floatx2{tq[0], tq[1]}, wc[tt & 1][2 * q], d2[2 * q]);
                d2[2 * q + 1] = __builtin_elementwise_fma(floatx2{tq[2], tq[3]}, wc[tt & 1][2 * q + 1], d2[2 * q + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    // fp16 hi / lo split, pixel tiles A (lanes 0-31) / B by permlane swap, three MFMAs per product - kernels_det_wave.hip's scheme and order
    floatx16 acc2[2][2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[tl][cb][e] = 0.f;
    const half_t *wf = a.wpfB + (long)lane * 8;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        unsigned hp[8], lp[8];
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
            floatx2 o = d2[8 * g + pr];
            o[0] = fmaxf(o[0], 0.f);
            o[1] = fmaxf(o[1], 0.f);
            const half2v h = __builtin_convertvector(o, half2v);
            const floatx2 back = __builtin_convertvector(h, floatx2);
            const half2v l = __builtin_convertvector(o - back, half2v);
            hp[pr] = __builtin_bit_cast(unsigned, h);
            lp[pr] = __builtin_bit_cast(unsigned, l);
        }
        unsigned ha[8], la[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint2v sh = __builtin_amdgcn_permlane32_swap(hp[i], hp[4 + i], false, false);
            const uint2v sl = __builtin_amdgcn_permlane32_swap(lp[i], lp[4 + i], false, false);
            ha[i] = sh[0];
            ha[4 + i] = sh[1];
            la[i] = sl[0];
            la[4 + i] = sl[1];
        }
        half8 ah[2], al[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            ah[cb] = *reinterpret_cast<const half8 *>(wf + ((long)(g * 2 + cb) * 2) * 512);
            al[cb] = *reinterpret_cast<const half8 *>(wf + ((long)(g * 2 + cb) * 2 + 1) * 512);
        }
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const half8 bh = __builtin_bit_cast(half8, uint4v{ha[4 * tl], ha[4 * tl + 1], ha[4 * tl + 2], ha[4 * tl + 3]});
            const half8 bl = __builtin_bit_cast(half8, uint4v{la[4 * tl], la[4 * tl + 1], la[4 * tl + 2], la[4 * tl + 3]});
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                acc2[tl][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bh, acc2[tl][cb], 0, 0, 0);
                acc2[tl][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bl, acc2[tl][cb], 0, 0, 0);
                acc2[tl][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], bh, acc2[tl][cb], 0, 0, 0);
            }
        }
    }
    const long hw2 = (long)H2 * W2;
    float bb[2][16];  // (loaded before the first store: the compiler must assume the output aliases them)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) bb[cb][e] = a.bpB[cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int pix = 32 * tl + r, oy = Y0 + (pix >> 3), ox = X0 + (pix & 7);
        float *ob = a.out + ((long)b * 64 + 4 * hi) * hw2 + (long)oy * W2 + ox;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) ob[(cb * 32 + (e & 3) + 8 * (e >> 2)) * hw2] = fmaxf(acc2[tl][cb][e] + bb[cb][e], 0.f);
    }
}

}  // namespace

// true: launched (both blocks are done)
bool launch_det_stage1b(const DwPwArgs &d1, const DwPwArgs &d2, hipStream_t s) {
    static const bool on = !(frt_tuning_env("FRT_DET_STAGE1B") && frt_tuning_env("FRT_DET_STAGE1B")[0] == '0');
    static const int min_b = frt_tuning_env("FRT_DET_STAGE1B_MINB") ? atoi(frt_tuning_env("FRT_DET_STAGE1B_MINB")) : 2;
    if (!on || !det_mfma_enabled() || d1.B < min_b || !d1.wdt || !d2.wdt || !d2.wpf) return false;
    if (!d1.wd || d1.add || d1.Cin != 32 || d1.Cout != 32 || d1.stride != 1 || !d1.relu) return false;
    if (!d2.wd || d2.add || d2.Cin != 32 || d2.Cout != 64 || d2.stride != 2 || !d2.relu || d2.H != d1.H || d2.W != d1.W || d2.in != d1.out) return false;
    if ((d2.Ho & 7) || (d2.Wo & 7) || d2.H != 2 * d2.Ho || d2.W != 2 * d2.Wo || (d2.Ho >> 3) < 3 || (d2.Wo >> 3) < 3) return false;
    S1bArgs a{d1.in, d1.wdt, d1.wp, d1.bp, d2.wdt, d2.bp, d2.wpf, d2.out, d1.B, d1.H, d1.W};
    const int tx = d2.Wo >> 3, ty = d2.Ho >> 3;
    hipLaunchKernelGGL(det_stage1b_kernel<true>, dim3((unsigned)(d1.B * (tx - 2) * (ty - 2))), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(det_stage1b_kernel<false>, dim3((unsigned)(d1.B * (2 * tx + 2 * (ty - 2)))), dim3(NT), 0, s, a);
    return true;
}
