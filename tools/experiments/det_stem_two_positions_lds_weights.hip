// PARKED (round 5, not linked into libfrt.so): det_stem_kernel with TWO positions per lane in P1 / P2 and the first two layers' weights
// broadcast from an LDS copy (ds_read_b128, in-order returns) instead of scalar loads.  Motivation: scalar loads return out of order, so every
// use waits with lgkmcnt(0) for ALL outstanding ones - the "chunk i + 2 in flight" prefetch of the shipped kernel overlaps nothing (43 % of wave
// cycles in s_waitcnt, profiles/r04/r04x_det_pmc.txt).  Outcome: compiles to 288 - 332 VGPRs (one wave per SIMD instead of four; the SGPR-chunk form
// of the same idea: 184 VGPRs and 455 scalar spills) - the compiler keeps far more of the unrolled body live than the source suggests, and
// neither sched_barrier fences nor opaque address offsets changed that.  Never measured on the GPU: at a quarter of the occupancy it cannot win.
// What would: the P1 / P2 bodies as hand-written asm with fixed registers (the dwpw_wave_kernel treatment).
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

// INTERIOR: the tile's regions lie inside the 320x320 map and every tap of the first conv inside the frame (81 % of the tiles at 640x640):
// no validity masks anywhere.  LDS is position-major ([position][channel]): a tap's 8 / 16 channels are two / four ds_read_b128, and the
// channel pairs they deliver are the operands of v_pk_fma_f32 (weights: scalar pairs, host-packed [pair][tap][2] for the depthwise parts).
template <bool INTERIOR>
__device__ __forceinline__ void stem_body(const StemArgs &a, float (*c1)[8], float (*b1s)[16], const float *wl, int b, int Y0, int X0) {
    constexpr int NT = 192;  // three waves per tile split the positions of P1 and P2 (measured at 32 frames, interior + ring: one wave 134 + 68 us, two 106 + 50, four 97 + 45, three 94 + 43: more waves per LDS byte against emptier passes); wave 0 alone runs P3
    const int lane = threadIdx.x;  // (position index in P1 / P2; < 64: the P3 lane)
    const int r2y0 = 2 * Y0 - 1, r2x0 = 2 * X0 - 1;  // block 1 region origin (H1 x W1 map)
    const int r1y0 = r2y0 - 1, r1x0 = r2x0 - 1;      // first conv region origin

    // ---- P1: first conv (det_conv1_u8_kernel's arithmetic) at the R1 x R1 positions
    {
        const uint8_t *fb = a.frames + (size_t)b * a.frame_stride;
        const float mean[3] = {104.f, 117.f, 123.f};
        if constexpr (INTERIOR) {
            // Round 5: TWO positions per lane (idx, idx + NT) walk the weights together, and the weights arrive in six chunks of 32 - 40 floats
            // instead of 27 chunks of 8 per position.  Scalar loads return out of order, so every use of one waits for ALL outstanding ones
            // (s_waitcnt lgkmcnt(0)): "chunk i + 2 in flight under chunk i" never overlapped anything, each of a position's 27 chunks paid a
            // full scalar-cache round trip for 16 cycles of arithmetic (1 instruction per ~ 9 cycles and SIMD, 43 % of wave cycles waiting,
            // profiles/r04/r04x_det_pmc.txt).  Now a wait buys 64 - 80 v_pk_fma_f32.  Same products in the same order per output: bit-identical.
            constexpr int N1 = R1 * R1;
            const int i1 = lane + NT;
            const bool has1 = i1 < N1;
            uint32_t raw[2][9];
            auto ldraw = [&](int idx, uint32_t (&d)[9]) {
                const int id = min(idx, N1 - 1), cy = id / R1, cx = id - cy * R1;
                const uint8_t *row = fb + (size_t)((r1y0 + cy) * 2 - 1) * a.row_stride + (size_t)(r1x0 + cx) * 6;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    __builtin_memcpy(&d[3 * kh], row + kh * a.row_stride - 3, 4);
                    __builtin_memcpy(&d[3 * kh + 1], row + kh * a.row_stride + 1, 4);
                    d[3 * kh + 2] = row[kh * a.row_stride + 5];
                }
            };
            ldraw(lane, raw[0]);
            ldraw(i1, raw[1]);
            floatx2 acc[2][4];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[p][c] = floatx2{0.f, 0.f};
            // weights from the workgroup's LDS copy (wl, same offsets as the packed buffer): every lane reads the same address - a broadcast
            // ds_read_b128 - and LDS reads return in order, so chunk i + 2 really is in flight under chunk i (counted lgkmcnt)
            constexpr int D = 2;
            floatx4 wq[D + 1][2];
            // (z: an opaque zero defined between two sched_barriers - without it every one of the 54 reads is hoisted to the top: 332 registers)
            auto ldw = [&](int i, int z, floatx4 (&d)[2]) {
                d[0] = *reinterpret_cast<const floatx4 *>(wl + z + OFF_W1 + 8 * i);
                d[1] = *reinterpret_cast<const floatx4 *>(wl + z + OFF_W1 + 8 * i + 4);
            };
            {
                int z;
                asm volatile("s_mov_b32 %0, 0" : "=s"(z));
                static_for<0, D>([&](auto jc) { ldw(decltype(jc)::value, z, wq[decltype(jc)::value]); });
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 27>([&](auto ic) {  // chunk i = (ci, tap): the eight output channels' weights of one input value
                constexpr int i = decltype(ic)::value, ci = i / 9, tap = i % 9, kh = tap / 3, kw = tap % 3;
                if constexpr (i + D < 27) {
                    int z;
                    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
                    ldw(i + D, z, wq[(i + D) % (D + 1)]);
                }
                const floatx4 w0 = wq[i % (D + 1)][0], w1 = wq[i % (D + 1)][1];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    // byte kw * 3 + ci of the 9 the row's three raw words hold (converted here instead of keeping 54 floats alive)
                    constexpr int bi = kw * 3 + ci;
                    const uint32_t word = raw[p][3 * kh + (bi >> 2)];
                    const float x = (float)((bi == 8 ? word : (word >> (8 * (bi & 3)))) & 255u) - mean[ci];
                    acc[p][0] = __builtin_elementwise_fma(floatx2{x, x}, floatx2{w0[0], w0[1]}, acc[p][0]);
                    acc[p][1] = __builtin_elementwise_fma(floatx2{x, x}, floatx2{w0[2], w0[3]}, acc[p][1]);
                    acc[p][2] = __builtin_elementwise_fma(floatx2{x, x}, floatx2{w1[0], w1[1]}, acc[p][2]);
                    acc[p][3] = __builtin_elementwise_fma(floatx2{x, x}, floatx2{w1[2], w1[3]}, acc[p][3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            {
                const float *bias = wl + OFF_B1;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    floatx4 o0, o1;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float ov = fmaxf(acc[p][c >> 1][c & 1] + bias[c], 0.f);
                        if (c < 4) o0[c] = ov;
                        else o1[c - 4] = ov;
                    }
                    if (p == 0 || has1) {
                        const int idx = p ? i1 : lane;
                        *reinterpret_cast<floatx4 *>(&c1[idx][0]) = o0;
                        *reinterpret_cast<floatx4 *>(&c1[idx][4]) = o1;
                    }
                }
            }
        } else
#pragma unroll 1
        for (int idx = lane; idx < R1 * R1; idx += NT) {
            // (an opaque zero: without it the 224 scalar weight loads are hoisted out of the loop and spilled - 569 SGPR spills)
            int z;
            asm volatile("s_mov_b32 %0, 0" : "=s"(z));
            const auto w = uni(a.w1) + z;
            const auto bias = uni(a.b1) + z;
            const int cy = idx / R1, cx = idx - cy * R1;
            const int oh = r1y0 + cy, ow = r1x0 + cx;
            const bool inside = INTERIOR || (oh >= 0 && oh < a.H1 && ow >= 0 && ow < a.W1);
            floatx2 acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = floatx2{0.f, 0.f};
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
                constexpr int D = INTERIOR ? 3 : 1;  // chunks of scalar weights in flight ahead of the one in use (scalar registers permitting)
                floatx2 wc[D + 1][4];
                static_for<0, D>([&](auto jc) { ld_pairs<4>(w, decltype(jc)::value, wc[decltype(jc)::value]); });
                static_for<0, 27>([&](auto ic) {  // chunk i = (ci, tap): the eight output channels' weights of one input value
                    constexpr int i = decltype(ic)::value;
                    if constexpr (i + D < 27) ld_pairs<4>(w, i + D, wc[(i + D) % (D + 1)]);
                    const float x = v[i / 9][i % 9];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_elementwise_fma(floatx2{x, x}, wc[i % (D + 1)][c], acc[c]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            floatx4 o0, o1;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float v = inside ? fmaxf(acc[c >> 1][c & 1] + bias[c], 0.f) : 0.f;
                if (c < 4) o0[c] = v;
                else o1[c - 4] = v;
            }
            *reinterpret_cast<floatx4 *>(&c1[idx][0]) = o0;
            *reinterpret_cast<floatx4 *>(&c1[idx][4]) = o1;
        }
    }
    __syncthreads();

    // ---- P2: conv_dw 8 -> 16 (dwpw_row4_kernel<16>'s arithmetic) at the R2 x R2 positions
    if constexpr (INTERIOR) {
        // (round 5, as P1: two positions per lane, the weights in six chunks - depthwise taps 0-4 + bias, taps 5-8, pointwise rows of two input
        //  channels at a time - so that a scalar-load wait covers 40 - 64 packed fmas instead of 4 - 8)
        constexpr int N2 = R2 * R2;
        const int i1 = lane + NT;
        const bool has1 = i1 < N2;
        int pos[2];
        {
            const int id1 = min(i1, N2 - 1);
            const int by0 = lane / R2, bx0 = lane - by0 * R2, by1 = id1 / R2, bx1 = id1 - by1 * R2;
            pos[0] = by0 * R1 + bx0;
            pos[1] = by1 * R1 + bx1;
        }
        floatx2 dd[2][4];
        {
            const floatx4 b0 = *reinterpret_cast<const floatx4 *>(wl + OFF_WDT1 + 72), b1 = *reinterpret_cast<const floatx4 *>(wl + OFF_WDT1 + 76);  // chunk 9: biases
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                dd[p][0] = floatx2{b0[0], b0[1]};
                dd[p][1] = floatx2{b0[2], b0[3]};
                dd[p][2] = floatx2{b1[0], b1[1]};
                dd[p][3] = floatx2{b1[2], b1[3]};
            }
        }
        static_for<0, 9>([&](auto tc) {  // depthwise: tap tt, weights [tap][pair][2] = 8 floats per tap
            constexpr int tt = decltype(tc)::value;
            int z;
            asm volatile("s_mov_b32 %0, 0" : "=s"(z));  // (opaque: keeps this tap's reads behind the previous tap's sched_barrier)
            const floatx4 w0 = *reinterpret_cast<const floatx4 *>(wl + z + OFF_WDT1 + 8 * tt), w1 = *reinterpret_cast<const floatx4 *>(wl + z + OFF_WDT1 + 8 * tt + 4);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float *tp = &c1[pos[p] + z + (tt / 3) * R1 + tt % 3][0];
                const floatx4 u0 = *reinterpret_cast<const floatx4 *>(tp), u1 = *reinterpret_cast<const floatx4 *>(tp + 4);
                dd[p][0] = __builtin_elementwise_fma(floatx2{u0[0], u0[1]}, floatx2{w0[0], w0[1]}, dd[p][0]);
                dd[p][1] = __builtin_elementwise_fma(floatx2{u0[2], u0[3]}, floatx2{w0[2], w0[3]}, dd[p][1]);
                dd[p][2] = __builtin_elementwise_fma(floatx2{u1[0], u1[1]}, floatx2{w1[0], w1[1]}, dd[p][2]);
                dd[p][3] = __builtin_elementwise_fma(floatx2{u1[2], u1[3]}, floatx2{w1[2], w1[3]}, dd[p][3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        floatx2 acc[2][8];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[p][c] = floatx2{0.f, 0.f};
        static_for<0, 8>([&](auto cc) {  // pointwise: input channel ci, weights [ci][16]
            constexpr int ci = decltype(cc)::value;
            int z;
            asm volatile("s_mov_b32 %0, 0" : "=s"(z));
            floatx4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const floatx4 *>(wl + z + OFF_WP1 + 16 * ci + 4 * q);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float d = fmaxf(dd[p][ci >> 1][ci & 1], 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[p][2 * q] = __builtin_elementwise_fma(floatx2{d, d}, floatx2{w[q][0], w[q][1]}, acc[p][2 * q]);
                    acc[p][2 * q + 1] = __builtin_elementwise_fma(floatx2{d, d}, floatx2{w[q][2], w[q][3]}, acc[p][2 * q + 1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        {
            const float *bp = wl + OFF_BP1;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    floatx4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 4 * q + e;
                        o[e] = fmaxf(acc[p][c >> 1][c & 1] + bp[c], 0.f);
                    }
                    if (p == 0 || has1) *reinterpret_cast<floatx4 *>(&b1s[p ? i1 : lane][4 * q]) = o;
                }
        }
    } else {
#pragma unroll 1
        for (int idx = lane; idx < R2 * R2; idx += NT) {
            int z;
            asm volatile("s_mov_b32 %0, 0" : "=s"(z));
            const auto wdt = uni(a.wdt1) + z, wp = uni(a.wp1) + z, bp = uni(a.bp1) + z;
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
            floatx2 acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = floatx2{0.f, 0.f};
            {
                floatx2 wc[2][8];
                ld_pairs<8>(wp, 0, wc[0]);
                static_for<0, 8>([&](auto cc) {
                    constexpr int ci = decltype(cc)::value;
                    if constexpr (ci + 1 < 8) ld_pairs<8>(wp, ci + 1, wc[(ci + 1) & 1]);
                    const float d = fmaxf(dd[ci >> 1][ci & 1], 0.f);
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = __builtin_elementwise_fma(floatx2{d, d}, wc[ci & 1][c], acc[c]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                floatx4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * q + e;
                    o[e] = inside ? fmaxf(acc[c >> 1][c & 1] + bp[c], 0.f) : 0.f;
                }
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
    __shared__ __attribute__((aligned(16))) float wl[OFF_WDT2];  // the first two layers' weights (448 floats of the packed buffer: a.w1 is its base)
    for (int i = threadIdx.x; i < OFF_WDT2; i += 192) wl[i] = a.w1[i];
    __syncthreads();
    const int tiles_x = a.W2 >> 3, tiles_y = a.H2 >> 3;
    int t = blockIdx.x, b, ty, tx;
    if (t < n_interior) {
        const int ix = tiles_x - 2, per = ix * (tiles_y - 2);
        b = t / per;
        t -= b * per;
        ty = t / ix;
        tx = t - ty * ix + 1;
        ty += 1;
        stem_body<true>(a, c1, b1s, wl, b, ty * 8, tx * 8);
    } else {  // the ring: top row, bottom row, then the left / right columns of the rows in between
        t -= n_interior;
        const int per = 2 * tiles_x + 2 * (tiles_y - 2);
        b = t / per;
        t -= b * per;
        if (t < tiles_x) { ty = 0; tx = t; }
        else if (t < 2 * tiles_x) { ty = tiles_y - 1; tx = t - tiles_x; }
        else { t -= 2 * tiles_x; ty = 1 + (t >> 1); tx = (t & 1) ? tiles_x - 1 : 0; }
        stem_body<false>(a, c1, b1s, wl, b, ty * 8, tx * 8);
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
