// Detector stem in one kernel (round 4): first conv (3 -> 8, stride 2, fed by the u8 frame) + conv_dw block 8 -> 16 at stride 1 + conv_dw
// block 16 -> 32 at stride 2 (net.py:102-106 of the reference's MobileNetV1 body: stage1[0..2]), 640x640 u8 in, 32 x 160x160 fp32 out.
//
// Why: as three kernels these layers write and re-read the two largest tensors of the network - 8 and 16 channels at 320x320 fp32, 105 +
// 210 MB per 32 frames, 630 MB of the 774 MB the three move - and take 59 + 97 + 110 us, a quarter of the detector, at 2.9 TB/s.  Here a
// wave owns an 8x8 tile of the 160x160 output and walks the three layers over the tile's halo'd regions through LDS:
//   P1  first conv at the 19x19 positions of the 320x320 map behind the tile  -> LDS c1[8][19x19]   (taps straight from the u8 frame)
//   P2  conv_dw 8 -> 16 at the inner 17x17 positions                          -> LDS b1[16][17x17]
//   P3  conv_dw 16 -> 32, stride 2, at the 8x8 outputs: lane = pixel, depthwise outputs in registers, pointwise product on the fp32 matrix
//       cores with the B operand by v_permlane32_swap (kernels_det_wave.hip)  -> global
// Positions outside the 320x320 map are stored as zeros (the next layer's zero padding).  1.41x / 1.13x of the first two layers' arithmetic is
// recomputed in the halos; nothing but the frame is read and nothing but the 32-channel output written (39 + 105 MB per 32 frames).
// Arithmetic: every chain is the stand-alone kernels' chain (det_conv1_u8_kernel, dwpw_row4_kernel<16>, dwpw_mfma_kernel<4, 1, 16, 1, 2>'s fp32
// MFMA path: same operand order, same fma order) - the output is bit-identical: tuning build, FRT_DET_STEM_CHECK=1 python tools/stem_check_run.py
// runs the three kernels and this one on the same frames and compares the 32-channel tensor element for element; in the GPU suite
// tests/test_gpu_detector.py::test_batch_of_32_equals_frame_by_frame (32 frames in one call against one frame per call) and test_fused_stem_equals_the_staged_path_on_other_identity_geometries (against the three kernels).
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "frt_kernels.h"

namespace {

typedef unsigned uint2v __attribute__((ext_vector_type(2)));

// every weight of the three layers in ONE buffer (gathered once per detector: det_stem_pack)
constexpr int OFF_W1 = 0, OFF_B1 = 216, OFF_WDT1 = 224, OFF_WP1 = 304, OFF_BP1 = 432, OFF_WDT2 = 448, OFF_WP2 = 608, OFF_BP2 = 1120, STEM_FLOATS = 1152;
struct StemArgs {
    const uint8_t *frames;
    size_t row_stride, frame_stride;
    // first conv [3][9][8], bias [8] | block 1: depthwise tap-major channel pairs [10][4][2] (taps 0-8, bias), pointwise (transposed) [8][16],
    // bias [16] | block 2: depthwise [10][8][2], pointwise (transposed) [16][32], bias [32]
    const float *w1, *b1, *wdt1, *wp1, *bp1, *wdt2, *wp2, *bp2;  // (separate kernel arguments into the one buffer: with a single base pointer
                                                                 //  the compiler's code measured 256 us against 206)
    float *out;                            // [B][32][H2][W2]
    int B, H, W, H1, W1, H2, W2;
};

template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T *uni(const T *p) {  // wave-uniform, read-only: scalar loads
    return reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

constexpr int R1 = 19, R2 = 17;  // edge of the first conv's / block 1's region behind an 8x8 output tile

typedef float floatx2 __attribute__((ext_vector_type(2)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// NP scalar weight pairs, fetched one chunk ahead of their use.  The chunk loop is unrolled and every chunk ends in a sched_barrier: left
// alone the compiler clusters all ~ 220 scalar loads of a phase at its top and spills them (569 SGPR spills, 900 v_readlane / v_writelane).
template <int NP, class P>
__device__ __forceinline__ void ld_pairs(const P w, int chunk, floatx2 (&dst)[NP]) {
#pragma unroll
    for (int c = 0; c < NP; ++c) dst[c] = floatx2{w[chunk * 2 * NP + 2 * c], w[chunk * 2 * NP + 2 * c + 1]};
}

// Round 6: the two long FMA chains - the first conv's 27 x 8 and block 1's pointwise 8 x 16 - run on the MATRIX pipe, which idled until P3.
// v_mfma_f32_4x4x1_16B_f32 is 16 independent 4x4 outer products D[i][j] += A[i] B[j]; with CBSZ = 4 the four A values of ONE block (ABID) are
// broadcast to all 16 blocks, and with the lane's own value as B lane l's four accumulator registers become acc[i] += w[i] * x: four fused
// multiply-adds of the lane's value with four weights - bit for bit the fmaf chain (tools/ubench/mfma_4x4_fma.hip: 0 of 256 values differ after
// 27 steps; 4.3 ns per instruction and SIMD).  One VECTOR register carries 16 weight quads (lane l: quad l / 4, element l & 3), the quad is
// picked by the ABID immediate: the first conv's 54 quads are 4 registers, the pointwise conv's 32 are 2 - no scalar weight stream in P1 at all
// (it was 27 s_load_dwordx8 per position pass with an lgkmcnt(0) in front of every second one), a third of the kernel's vector instructions
// gone.  The chains keep their order ((ci, tap) ascending; cin ascending), so the output is unchanged bit for bit (FRT_DET_STEM_CHECK, tests).
struct StemW {
    float w1[4];  // first conv: quad Q = (ci * 9 + tap) * 2 + (cout quad) lives in w1[Q / 16], block Q % 16
    float wp[2];  // block 1 pointwise: quad Q = cin * 4 + (cout quad)
};
template <int Q>
__device__ __forceinline__ floatx4 fma4(const float (&w)[4], float x, floatx4 acc) {  // acc[i] += W[quad Q][i] * x
    return __builtin_amdgcn_mfma_f32_4x4x1f32(w[Q / 16], x, acc, 4, Q % 16, 0);
}
template <int Q>
__device__ __forceinline__ floatx4 fma4p(const float (&w)[2], float x, floatx4 acc) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(w[Q / 16], x, acc, 4, Q % 16, 0);
}

// INTERIOR: the tile's regions lie inside the 320x320 map and every tap of the first conv inside the frame (81 % of the tiles at 640x640):
// no validity masks anywhere.  LDS is position-major ([position][channel]): a tap's 8 / 16 channels are two / four ds_read_b128, and the
// channel pairs they deliver are the operands of v_pk_fma_f32 (weights: scalar pairs, host-packed [pair][tap][2] for the depthwise parts).
template <bool INTERIOR>
__device__ __forceinline__ void stem_body(const StemArgs &a, const StemW &W, float (*c1)[8], float (*b1s)[16], int b, int Y0, int X0) {
    constexpr int NT = 192;  // three waves per tile split the positions of P1 and P2 (measured at 32 frames, interior + ring: one wave 134 + 68 us, two 106 + 50, four 97 + 45, three 94 + 43: more waves per LDS byte against emptier passes); wave 0 alone runs P3
    const int lane = threadIdx.x;  // (position index in P1 / P2; < 64: the P3 lane)
    const int r2y0 = 2 * Y0 - 1, r2x0 = 2 * X0 - 1;  // block 1 region origin (H1 x W1 map)
    const int r1y0 = r2y0 - 1, r1x0 = r2x0 - 1;      // first conv region origin

    // ---- P1: first conv (det_conv1_u8_kernel's arithmetic) at the R1 x R1 positions
    {
        const uint8_t *fb = a.frames + (size_t)b * a.frame_stride;
        const float mean[3] = {104.f, 117.f, 123.f};
        if constexpr (INTERIOR) {
            // the nine raw loads of the NEXT position are in flight under this position's arithmetic (one pass = one HBM / L2 round trip otherwise:
            // six per wave, with little more than one wave per SIMD to hide them)
            uint32_t raw[9], nxt[9];
            auto ldraw = [&](int idx, uint32_t (&d)[9]) {
                const int id = min(idx, R1 * R1 - 1), cy = id / R1, cx = id - cy * R1;
                const uint8_t *row = fb + (size_t)((r1y0 + cy) * 2 - 1) * a.row_stride + (size_t)(r1x0 + cx) * 6;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    __builtin_memcpy(&d[3 * kh], row + kh * a.row_stride - 3, 4);
                    __builtin_memcpy(&d[3 * kh + 1], row + kh * a.row_stride + 1, 4);
                    d[3 * kh + 2] = row[kh * a.row_stride + 5];
                }
            };
            ldraw(lane, raw);
#pragma unroll 1
            for (int idx = lane; idx < R1 * R1; idx += NT) {
                const auto bias = uni(a.b1);
                ldraw(idx + NT, nxt);
                float v[3][9];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const uint32_t d0 = raw[3 * kh], d1 = raw[3 * kh + 1], d2 = raw[3 * kh + 2];
                    const uint32_t by[9] = {d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u, d0 >> 24, d1 & 255u, (d1 >> 8) & 255u, (d1 >> 16) & 255u, d1 >> 24, d2};
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) v[ci][kh * 3 + kw] = (float)by[kw * 3 + ci] - mean[ci];
                }
                floatx4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;  // couts 0-3 / 4-7; chunk i = (ci, tap): the chain order of det_conv1_u8_kernel
                static_for<0, 27>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    q0 = fma4<2 * i>(W.w1, v[i / 9][i % 9], q0);
                    q1 = fma4<2 * i + 1>(W.w1, v[i / 9][i % 9], q1);
                });
                floatx4 o0, o1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o0[c] = fmaxf(q0[c] + bias[c], 0.f);
                    o1[c] = fmaxf(q1[c] + bias[4 + c], 0.f);
                }
                *reinterpret_cast<floatx4 *>(&c1[idx][0]) = o0;
                *reinterpret_cast<floatx4 *>(&c1[idx][4]) = o1;
#pragma unroll
                for (int k = 0; k < 9; ++k) raw[k] = nxt[k];
            }
        } else
#pragma unroll 1
        for (int idx = lane; idx < R1 * R1; idx += NT) {
            // (an opaque zero: without it the 224 scalar weight loads are hoisted out of the loop and spilled - 569 SGPR spills)
            const auto bias = uni(a.b1);
            const int cy = idx / R1, cx = idx - cy * R1;
            const int oh = r1y0 + cy, ow = r1x0 + cx;
            const bool inside = INTERIOR || (oh >= 0 && oh < a.H1 && ow >= 0 && ow < a.W1);
            floatx4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
            if (inside) {
                float v[3][9];
                if (INTERIOR || ow * 2 + 1 < a.W) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int ih = oh * 2 - 1 + kh;
                        const bool rok = INTERIOR || (ih >= 0 && ih < a.H);
                        const uint8_t *row = fb + (size_t)(rok ? ih : 0) * a.row_stride + (size_t)ow * 6;
                        uint32_t d0, d1;
                        if (INTERIOR || ow > 0) __builtin_memcpy(&d0, row - 3, 4);
                        else { __builtin_memcpy(&d0, row, 4); d0 <<= 24; }
                        __builtin_memcpy(&d1, row + 1, 4);
                        const uint32_t d2 = row[5];
                        const uint32_t by[9] = {d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u, d0 >> 24, d1 & 255u, (d1 >> 8) & 255u, (d1 >> 16) & 255u, d1 >> 24, d2};
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const bool ok = INTERIOR || (rok && (kw > 0 || ow > 0));
#pragma unroll
                            for (int ci = 0; ci < 3; ++ci) v[ci][kh * 3 + kw] = ok ? (float)by[kw * 3 + ci] - mean[ci] : 0.f;
                        }
                    }
                } else {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                            const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                            const uint8_t *px = fb + (size_t)(ok ? ih : 0) * a.row_stride + (size_t)(ok ? iw : 0) * 3;
#pragma unroll
                            for (int ci = 0; ci < 3; ++ci) {
                                const float raw = (float)px[ci];
                                v[ci][kh * 3 + kw] = ok ? raw - mean[ci] : 0.f;
                            }
                        }
                }
                static_for<0, 27>([&](auto ic) {  // chunk i = (ci, tap): the eight output channels' weights of one input value
                    constexpr int i = decltype(ic)::value;
                    q0 = fma4<2 * i>(W.w1, v[i / 9][i % 9], q0);
                    q1 = fma4<2 * i + 1>(W.w1, v[i / 9][i % 9], q1);
                });
            }
            floatx4 o0, o1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                o0[c] = inside ? fmaxf(q0[c] + bias[c], 0.f) : 0.f;
                o1[c] = inside ? fmaxf(q1[c] + bias[4 + c], 0.f) : 0.f;
            }
            *reinterpret_cast<floatx4 *>(&c1[idx][0]) = o0;
            *reinterpret_cast<floatx4 *>(&c1[idx][4]) = o1;
        }
    }
    __syncthreads();

    // ---- P2: conv_dw 8 -> 16 (dwpw_row4_kernel<16>'s arithmetic) at the R2 x R2 positions
    {
#pragma unroll 1
        for (int idx = lane; idx < R2 * R2; idx += NT) {
            int z;
            asm volatile("s_mov_b32 %0, 0" : "=s"(z));
            const auto wdt = uni(a.wdt1) + z, bp = uni(a.bp1) + z;
            const int by = idx / R2, bx = idx - by * R2;
            const bool inside = INTERIOR || (r2y0 + by >= 0 && r2y0 + by < a.H1 && r2x0 + bx >= 0 && r2x0 + bx < a.W1);
            floatx2 dd[4];  // depthwise outputs, channel pairs; weights tap-major [tap | bias][pair][2]
            {
                constexpr int D = INTERIOR ? 3 : 1;
                floatx2 wc[D + 1][4];
                ld_pairs<4>(wdt, 9, dd);
                static_for<0, D>([&](auto jc) { ld_pairs<4>(wdt, decltype(jc)::value, wc[decltype(jc)::value]); });
                static_for<0, 9>([&](auto tc) {
                    constexpr int tt = decltype(tc)::value;
                    if constexpr (tt + D < 9) ld_pairs<4>(wdt, tt + D, wc[(tt + D) % (D + 1)]);
                    const float *tp = &c1[(by + tt / 3) * R1 + bx + tt % 3][0];
                    const floatx4 t0 = *reinterpret_cast<const floatx4 *>(tp), t1 = *reinterpret_cast<const floatx4 *>(tp + 4);
                    dd[0] = __builtin_elementwise_fma(floatx2{t0[0], t0[1]}, wc[tt % (D + 1)][0], dd[0]);
                    dd[1] = __builtin_elementwise_fma(floatx2{t0[2], t0[3]}, wc[tt % (D + 1)][1], dd[1]);
                    dd[2] = __builtin_elementwise_fma(floatx2{t1[0], t1[1]}, wc[tt % (D + 1)][2], dd[2]);
                    dd[3] = __builtin_elementwise_fma(floatx2{t1[2], t1[3]}, wc[tt % (D + 1)][3], dd[3]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            floatx4 accq[4];  // cout quads; cin ascending: the chain order of dwpw_row4_kernel<16>
#pragma unroll
            for (int q = 0; q < 4; ++q) accq[q] = floatx4{0.f, 0.f, 0.f, 0.f};
            static_for<0, 8>([&](auto cc) {
                constexpr int ci = decltype(cc)::value;
                const float d = fmaxf(dd[ci >> 1][ci & 1], 0.f);
                accq[0] = fma4p<4 * ci>(W.wp, d, accq[0]);
                accq[1] = fma4p<4 * ci + 1>(W.wp, d, accq[1]);
                accq[2] = fma4p<4 * ci + 2>(W.wp, d, accq[2]);
                accq[3] = fma4p<4 * ci + 3>(W.wp, d, accq[3]);
            });
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                floatx4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = inside ? fmaxf(accq[q][e] + bp[4 * q + e], 0.f) : 0.f;
                *reinterpret_cast<floatx4 *>(&b1s[idx][4 * q]) = o;
            }
        }
    }
    __syncthreads();

    // ---- P3: conv_dw 16 -> 32 at stride 2; lane = output pixel (py, px) of the tile
    if (lane >= 64) return;
    const int py = lane >> 3, px = lane & 7;
    const int r = lane & 31, hi = lane >> 5;
    floatx2 d2[8];
    {
        const auto wdt = uni(a.wdt2);
        floatx2 wc[2][8];
        ld_pairs<8>(wdt, 9, d2);
        ld_pairs<8>(wdt, 0, wc[0]);
        static_for<0, 9>([&](auto tc) {
            constexpr int tt = decltype(tc)::value;
            if constexpr (tt + 1 < 9) ld_pairs<8>(wdt, tt + 1, wc[(tt + 1) & 1]);
            const float *tp = &b1s[(2 * py + tt / 3) * R2 + 2 * px + tt % 3][0];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const floatx4 t = *reinterpret_cast<const floatx4 *>(tp + 4 * q);
                d2[2 * q] = __builtin_elementwise_fma(floatx2{t[0], t[1]}, wc[tt & 1][2 * q], d2[2 * q]);
                d2[2 * q + 1] = __builtin_elementwise_fma(floatx2{t[2], t[3]}, wc[tt & 1][2 * q + 1], d2[2 * q + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    // pointwise 16 -> 32 on v_mfma_f32_32x32x2f32, k-step ks = channels (2 ks, 2 ks + 1): lane (n, hi) of a pixel tile supplies channel
    // 2 ks + hi of pixel n.  Lanes 0-31 are tile A (rows 0-3), lanes 32-63 tile B: swapping the upper half of channel 2 ks with the lower half
    // of channel 2 ks + 1 leaves tile A's operand in the first register and tile B's in the second
    floatx16 accA, accB;
#pragma unroll
    for (int e = 0; e < 16; ++e) accA[e] = accB[e] = 0.f;
    unsigned ua[8], ub[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const float e0 = fmaxf(d2[ks][0], 0.f), e1 = fmaxf(d2[ks][1], 0.f);
        const uint2v sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1), false, false);
        ua[ks] = sw[0];
        ub[ks] = sw[1];
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const float aw = a.wp2[(2 * ks + hi) * 32 + r];
        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, __builtin_bit_cast(float, ua[ks]), accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, __builtin_bit_cast(float, ub[ks]), accB, 0, 0, 0);
    }
    const long hw2 = (long)a.H2 * a.W2;
    float bb[16];  // (loaded before the first store: the compiler must assume the output aliases them)
#pragma unroll
    for (int e = 0; e < 16; ++e) bb[e] = a.bp2[(e & 3) + 8 * (e >> 2) + 4 * hi];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int pix = 32 * tl + r, oy = Y0 + (pix >> 3), ox = X0 + (pix & 7);
        float *ob = a.out + ((long)b * 32 + 4 * hi) * hw2 + (long)oy * a.W2 + ox;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2);
            ob[co * hw2] = fmaxf((tl ? accB[e] : accA[e]) + bb[e], 0.f);
        }
    }
}

// Two launches: the interior tiles (everything but the outermost ring of 8x8 tiles) and the ring.  As one kernel with a run-time branch the
// register allocation was the border path's (93 scalar spills in the interior loops).
// ONE launch, interior tiles first, then the ring: blockIdx.x < n_interior selects.  (As two launches the register allocation of the interior loops
// is cleaner - 9 scalar spills against 99 - but the ring's few, slow workgroups then run alone: 94 + 43 us against 133 at 32 frames, 23 + 18
// against 33 at 4, 14 + 13 against 16.5 at one.)
__global__ __launch_bounds__(192) void det_stem_kernel(StemArgs a, int n_interior) {
    __shared__ __attribute__((aligned(16))) float c1[R1 * R1][8];
    __shared__ __attribute__((aligned(16))) float b1s[R2 * R2][16];
    StemW W;  // lane l: quad (16 k + l / 4), element l & 3 of the first conv's [27][8] = 54 quads (k < 4; the last 10 quads unused) and of
              // the pointwise conv's [8][16] = 32 quads (k < 2)
    {
        const int l = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = 16 * k + (l >> 2);
            W.w1[k] = q < 54 ? a.w1[q * 4 + (l & 3)] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) W.wp[k] = a.wp1[(16 * k + (l >> 2)) * 4 + (l & 3)];
    }
    const int tiles_x = a.W2 >> 3, tiles_y = a.H2 >> 3;
    int t = blockIdx.x, b, ty, tx;
    if (t < n_interior) {
        const int ix = tiles_x - 2, per = ix * (tiles_y - 2);
        b = t / per;
        t -= b * per;
        ty = t / ix;
        tx = t - ty * ix + 1;
        ty += 1;
        stem_body<true>(a, W, c1, b1s, b, ty * 8, tx * 8);
    } else {  // the ring: top row, bottom row, then the left / right columns of the rows in between
        t -= n_interior;
        const int per = 2 * tiles_x + 2 * (tiles_y - 2);
        b = t / per;
        t -= b * per;
        if (t < tiles_x) { ty = 0; tx = t; }
        else if (t < 2 * tiles_x) { ty = tiles_y - 1; tx = t - tiles_x; }
        else { t -= 2 * tiles_x; ty = 1 + (t >> 1); tx = (t & 1) ? tiles_x - 1 : 0; }
        stem_body<false>(a, W, c1, b1s, b, ty * 8, tx * 8);
    }
}

}  // namespace

size_t det_stem_weight_floats() { return STEM_FLOATS; }
// gathers the three layers' weights (device pointers) into the kernel's one buffer; called once when the detector is built
void det_stem_pack(const Conv3Args &c, const DwPwArgs &d1, const DwPwArgs &d2, float *dst, hipStream_t s) {
    auto cp = [&](int off, const float *src, int n) { (void)hipMemcpyAsync(dst + off, src, sizeof(float) * n, hipMemcpyDeviceToDevice, s); };
    cp(OFF_W1, c.w, 216); cp(OFF_B1, c.b, 8);
    cp(OFF_WDT1, d1.wdt, 80); cp(OFF_WP1, d1.wp, 128); cp(OFF_BP1, d1.bp, 16);
    cp(OFF_WDT2, d2.wdt, 160); cp(OFF_WP2, d2.wp, 512); cp(OFF_BP2, d2.bp, 32);
}

// true: launched (the three layers are done).  Identity letterbox only (the u8 frame IS the network input), 8x8-tileable output.
// d1.stem: the buffer det_stem_pack() filled
bool launch_det_stem(const uint8_t *frames, size_t row_stride, size_t frame_stride, const Conv3Args &c, const DwPwArgs &d1, const DwPwArgs &d2, hipStream_t s) {
    static const bool on = !(frt_tuning_env("FRT_DET_STEM") && frt_tuning_env("FRT_DET_STEM")[0] == '0');
    static const int min_b = frt_tuning_env("FRT_DET_STEM_MINB") ? atoi(frt_tuning_env("FRT_DET_STEM_MINB")) : 1;  // (us, this / the three kernels: 1 frame 16.5 / 27.5, 2: 20 / 31, 4: 33 / 42.5, 8: 48 / 72, 32: 133 / 266)
    if (!on || !det_mfma_enabled() || !d1.stem || c.B < min_b) return false;
    if (c.Cin != 3 || c.Cout != 8 || c.stride != 2 || !c.relu || c.out_ctotal != 8 || c.out_coff != 0) return false;
    if (!d1.wd || d1.add || d1.Cin != 8 || d1.Cout != 16 || d1.stride != 1 || !d1.relu || d1.H != c.Ho || d1.W != c.Wo) return false;
    if (!d2.wd || d2.add || d2.Cin != 16 || d2.Cout != 32 || d2.stride != 2 || !d2.relu || d2.H != d1.Ho || d2.W != d1.Wo) return false;
    if ((c.W & 1) || (c.H & 1) || (d2.Ho & 7) || (d2.Wo & 7) || d2.H != 2 * d2.Ho || d2.W != 2 * d2.Wo || d1.in != c.out || d2.in != d1.out) return false;
    const float *w = d1.stem;
    StemArgs a{frames, row_stride, frame_stride, w + OFF_W1, w + OFF_B1, w + OFF_WDT1, w + OFF_WP1, w + OFF_BP1, w + OFF_WDT2, w + OFF_WP2, w + OFF_BP2, d2.out, c.B, c.H, c.W, c.Ho, c.Wo, d2.Ho, d2.Wo};
    const int tx = d2.Wo >> 3, ty = d2.Ho >> 3;
    if (tx < 3 || ty < 3) return false;
    const int n_int = c.B * (tx - 2) * (ty - 2), n_ring = c.B * (2 * tx + 2 * (ty - 2));
    hipLaunchKernelGGL(det_stem_kernel, dim3((unsigned)(n_int + n_ring)), dim3(192), 0, s, a, n_int);
    return true;
}
