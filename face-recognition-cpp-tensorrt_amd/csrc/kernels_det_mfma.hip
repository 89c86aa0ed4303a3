// RetinaFace-mobilenet0.25 conv_dw blocks and 1x1 convs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: true fp32
// multiply-add, so the result stays inside the fp32 tolerances of tests/test_gpu_detector.py).
//
// Arithmetic spec: /root/reference/conversion/retina/models/net.py:19-31 (conv_dw = depthwise 3x3 + BN + ReLU, pointwise 1x1
// + BN + ReLU), :68-98 (FPN laterals: 1x1 + BN + ReLU, nearest upsample + add).  BN is folded on the host.
//
// Why MFMA here although the detector is memory-bound: the scalar-FMA version (kernels_det.hip, kept as the generic
// fallback) spends one VALU instruction per 64 FMAs and reaches 15-37 TFLOP/s on the pointwise part; every block below
// 80x80 was latency-, not bandwidth-bound (measured 45-170 us against HBM floors of 5-20 us).  One MFMA retires 4096 FMAs
// per wave instruction, which leaves the issue slots to the depthwise stencil and the loads.
//
//   dwpw_mfma_kernel   one workgroup = NPW pixel groups (32 output pixels each) x NCW output-channel groups.
//                      Per chunk of KC input channels: all threads compute depthwise+bias+ReLU for the chunk's
//                      KC x 32*NPW values (4 consecutive pixels per thread: 3 rows x (aligned float4 + 2 edge scalars)) and
//                      write them to LDS [KC][32*NPW] - already the B-operand layout (k = row, pixel = lane) - then every
//                      wave runs KC/2 x CBW MFMAs with A = pointwise weights [Cout][k] streamed L2 -> registers one chunk
//                      ahead.  The LDS chunk is double-buffered: one barrier per chunk.  The depthwise intermediate never
//                      touches HBM (the split path wrote and re-read it) and is computed exactly once (no per-channel-tile
//                      recompute of the fused scalar path).
//   pw_mfma_kernel     plain 1x1 conv (FPN laterals): B operand straight from global (32 consecutive pixels x 2 channels per
//                      load instruction, 128-byte segments), no LDS, waves independent; fused bias/ReLU/upsample-add epilogue.
//
// Pixel mapping: 8x16 2-D tiles where the map is wide (halo re-read 1.4x instead of 3x), linear order over (b, y, x) on the
// small maps (no tile-shape waste at 40x40 / 20x20).  Logical tiles are assigned so that every XCD's L2 sees a contiguous
// range (neighbouring tiles share halo rows).
#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "frt_kernels.h"

namespace {

struct PixMap {
    int b, oy, ox;
    bool ok;
};

// t: pixel index inside the workgroup tile (TP pixels), lid: logical tile id
template <bool MODE2D, int TP>
__device__ __forceinline__ PixMap map_pixel(int lid, int t, int B, int Ho, int Wo, int tiles_x, int tiles_y) {
    PixMap m;
    if (MODE2D) {
        const int per = tiles_x * tiles_y;
        m.b = lid / per;
        const int rem = lid - m.b * per;
        const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
        m.oy = tyi * 8 + (t >> 4);
        m.ox = txi * 16 + (t & 15);
        m.ok = m.b < B && m.oy < Ho && m.ox < Wo;
    } else {
        const long g = (long)lid * TP + t;
        const int HoWo = Ho * Wo;
        m.b = (int)(g / HoWo);
        const int p = (int)(g - (long)m.b * HoWo);
        m.oy = p / Wo;
        m.ox = p - m.oy * Wo;
        m.ok = m.b < B;
    }
    return m;
}

__device__ __forceinline__ int xcd_logical_tile(int nblocks) {
    // block b runs on XCD b % 8: give every XCD a contiguous range of logical tiles (bijective)
    const int bq = nblocks >> 3, brem = nblocks & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    return (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
}

// SPLIT: the pointwise product runs on the fp16 matrix cores at fp32 accuracy (x = hi + lo, three v_mfma_f32_32x32x16_f16 per 16
// channels = 96 matrix-pipe clocks against 512 for v_mfma_f32_32x32x2f32; same idea and error analysis as kernels_det_conv3h.hip):
// the depthwise stage stores its output split into LDS rows [pixel][KC hi | KC lo | pad], weights come pre-split from the host.
//
// PRE > 0 (SPLIT only; a frame or two per call, where a workgroup is alone on its CU and its lifetime is the chain "loads of chunk c + 1 ->
// depthwise -> LDS -> barrier" once per chunk): the taps AND the depthwise weights of PRE chunks are in flight ahead of the chunk whose
// depthwise part is being computed - unconditional loads from clamped addresses, the zero padding applied by selects when the chunk is
// consumed.  Same values, same operations in the same order: bit-identical to PRE = 0.
template <int NPW, int NCW, int KC, int CBW, int STRIDE, bool MODE2D, bool SPLIT = false, int PRE = 0>
__global__ __launch_bounds__(64 * NPW * NCW) void dwpw_mfma_kernel(DwPwArgs a, int tiles_x, int tiles_y) {
    constexpr int T = 64 * NPW * NCW;       // threads
    constexpr int TP = 32 * NPW;            // pixels per workgroup tile
    constexpr int SEGS = TP / 4;            // 4-pixel segments per channel
    constexpr int CH_PASS = T / SEGS;       // channels covered by one pass of all threads
    constexpr int ITEMS = KC / CH_PASS;     // depthwise segments per thread per chunk
    static_assert(ITEMS >= 1 && KC % CH_PASS == 0, "chunk must cover whole passes");
    constexpr int KS = KC / 2;              // MFMA k-steps per chunk

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // depthwise weights [Cin][12] (9 taps, bias, 2 pad), host-packed: read straight from global (three 16-byte L1/L2 hits per item).
    // Staging them in LDS first cost every workgroup a load -> store -> barrier phase (~2 us of a ~13 us workgroup lifetime).
    const float *wsm = a.wd12;
    float *buf = smem;                       // [2][KC][TP]
    constexpr int ROWH = 2 * KC + 8;         // SPLIT: halves per pixel row (80 / 144 bytes: conflict-free ds_read_b128)
    half_t *hbuf = reinterpret_cast<half_t *>(smem);  // SPLIT: [2][TP][ROWH]

    const int lid = xcd_logical_tile(gridDim.x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wp = wave % NPW, wc = wave / NPW;
    const int r = lane & 31, hi = lane >> 5;
    const int HW = a.H * a.W;

    // ---- depthwise role: one 4-pixel segment, CH_PASS-strided channels
    const int seg = threadIdx.x % SEGS, chl = threadIdx.x / SEGS;
    const PixMap ms = map_pixel<MODE2D, TP>(lid, seg * 4, a.B, a.Ho, a.Wo, tiles_x, tiles_y);
#ifdef FRT_ABLATE
    // timing build, linear tiles only (tiles_x is unused there; FRT_DWPW_ABLATE): bit 0 no output stores, bit 1 every workgroup reads the
    // first rows of image 0 (cache-resident input).  Round 4, 128 -> 128 block at 40x40, 32 frames: 45 us -> 44 (no stores) / 44 (cached
    // input) / 38 (both) in the ablation build - the launch is not memory-bound; profiles/r04/r04o_det_ablations.txt
    const int abl = MODE2D ? 0 : tiles_x;
    const float *inb = a.in + (long)((ms.ok && !(abl & 2)) ? ms.b : 0) * a.Cin * HW;
#else
    const float *inb = a.in + (long)(ms.ok ? ms.b : 0) * a.Cin * HW;
#endif
    int roff[3];
    bool rok[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int iy = ms.oy * STRIDE - 1 + k;
        rok[k] = ms.ok && iy >= 0 && iy < a.H;
        roff[k] = rok[k] ? iy * a.W + ms.ox * STRIDE : 0;
#ifdef FRT_ABLATE
        if (abl & 2) roff[k] = rok[k] ? (iy % 3) * a.W + ms.ox * STRIDE : 0;
#endif
    }
    const bool left_ok = ms.ox > 0;                            // column ox*STRIDE - 1 exists
    const bool right_ok = STRIDE == 1 && ms.ox + 4 < a.W;      // column ox + 4 exists (stride 1 only)

    auto depthwise_chunk = [&](int c0, float *dst) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int cl = it * CH_PASS + chl;
            const int ch = c0 + cl;
            const float *x = inb + (long)ch * HW;
            const floatx4 w0 = *reinterpret_cast<const floatx4 *>(wsm + ch * 12);
            const floatx4 w1 = *reinterpret_cast<const floatx4 *>(wsm + ch * 12 + 4);
            const floatx4 w2 = *reinterpret_cast<const floatx4 *>(wsm + ch * 12 + 8);
            const float wt[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
            floatx4 o = {w2[1], w2[1], w2[1], w2[1]};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (STRIDE == 1) {
                    float v[6];
                    const floatx4 m = rok[k] ? *reinterpret_cast<const floatx4 *>(x + roff[k]) : floatx4{0.f, 0.f, 0.f, 0.f};
                    v[0] = (rok[k] && left_ok) ? x[roff[k] - 1] : 0.f;
                    v[5] = (rok[k] && right_ok) ? x[roff[k] + 4] : 0.f;
                    v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = fmaf(v[j + 2], wt[3 * k + 2], fmaf(v[j + 1], wt[3 * k + 1], fmaf(v[j], wt[3 * k], o[j])));
                } else {
                    float v[9];
                    const floatx4 m0 = rok[k] ? *reinterpret_cast<const floatx4 *>(x + roff[k]) : floatx4{0.f, 0.f, 0.f, 0.f};
                    const floatx4 m1 = rok[k] ? *reinterpret_cast<const floatx4 *>(x + roff[k] + 4) : floatx4{0.f, 0.f, 0.f, 0.f};
                    v[0] = (rok[k] && left_ok) ? x[roff[k] - 1] : 0.f;
                    v[1] = m0[0]; v[2] = m0[1]; v[3] = m0[2]; v[4] = m0[3];
                    v[5] = m1[0]; v[6] = m1[1]; v[7] = m1[2]; v[8] = m1[3];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = fmaf(v[2 * j + 2], wt[3 * k + 2], fmaf(v[2 * j + 1], wt[3 * k + 1], fmaf(v[2 * j], wt[3 * k], o[j])));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
            if (SPLIT) {
                half_t *hd = reinterpret_cast<half_t *>(dst);  // dst = this chunk's buffer (byte offset chosen by the caller)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const half_t xh = (half_t)o[j];
                    hd[(seg * 4 + j) * ROWH + cl] = xh;
                    hd[(seg * 4 + j) * ROWH + KC + cl] = (half_t)(o[j] - (float)xh);
                }
            } else {
                *reinterpret_cast<floatx4 *>(dst + cl * TP + seg * 4) = o;
            }
        }
    };

    // PRE > 0: the same chunk in two halves - loads now, arithmetic PRE chunks later
    struct Taps {
        floatx4 m0[3], m1[STRIDE == 2 ? 3 : 1], w0, w1, w2;
        float lft[3], rgt[STRIDE == 1 ? 3 : 1];
    };
    const int loff = left_ok ? 1 : 0, roffr = right_ok ? 4 : 3;
    auto dw_load = [&](int c0, Taps (&tp)[ITEMS]) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int ch = c0 + it * CH_PASS + chl;
            const float *x = inb + (long)ch * HW;
            tp[it].w0 = *reinterpret_cast<const floatx4 *>(wsm + ch * 12);
            tp[it].w1 = *reinterpret_cast<const floatx4 *>(wsm + ch * 12 + 4);
            tp[it].w2 = *reinterpret_cast<const floatx4 *>(wsm + ch * 12 + 8);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                tp[it].m0[k] = *reinterpret_cast<const floatx4 *>(x + roff[k]);
                tp[it].lft[k] = x[rok[k] ? roff[k] - loff : 0];
                if (STRIDE == 1) tp[it].rgt[k] = x[roff[k] + roffr];
                else tp[it].m1[k] = *reinterpret_cast<const floatx4 *>(x + roff[k] + 4);
            }
        }
    };
    auto dw_compute = [&](int c0, const Taps (&tp)[ITEMS], float *dst) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int cl = it * CH_PASS + chl;
            const floatx4 w0 = tp[it].w0, w1 = tp[it].w1, w2 = tp[it].w2;
            const float wt[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
            const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
            floatx4 o = {w2[1], w2[1], w2[1], w2[1]};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (STRIDE == 1) {
                    float v[6];
                    const floatx4 m = rok[k] ? tp[it].m0[k] : zero4;
                    v[0] = (rok[k] && left_ok) ? tp[it].lft[k] : 0.f;
                    v[5] = (rok[k] && right_ok) ? tp[it].rgt[k] : 0.f;
                    v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = fmaf(v[j + 2], wt[3 * k + 2], fmaf(v[j + 1], wt[3 * k + 1], fmaf(v[j], wt[3 * k], o[j])));
                } else {
                    float v[9];
                    const floatx4 m0 = rok[k] ? tp[it].m0[k] : zero4;
                    const floatx4 m1 = rok[k] ? tp[it].m1[k] : zero4;
                    v[0] = (rok[k] && left_ok) ? tp[it].lft[k] : 0.f;
                    v[1] = m0[0]; v[2] = m0[1]; v[3] = m0[2]; v[4] = m0[3];
                    v[5] = m1[0]; v[6] = m1[1]; v[7] = m1[2]; v[8] = m1[3];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = fmaf(v[2 * j + 2], wt[3 * k + 2], fmaf(v[2 * j + 1], wt[3 * k + 1], fmaf(v[2 * j], wt[3 * k], o[j])));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
            half_t *hd = reinterpret_cast<half_t *>(dst);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const half_t xh = (half_t)o[j];
                hd[(seg * 4 + j) * ROWH + cl] = xh;
                hd[(seg * 4 + j) * ROWH + KC + cl] = (half_t)(o[j] - (float)xh);
            }
        }
    };

    // ---- MFMA role: 32 pixels (wp) x CBW blocks of 32 output channels (wc)
    floatx16 acc[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
    const int co_base = ((int)blockIdx.y * NCW + wc) * CBW * 32;  // blockIdx.y: output-channel group (depthwise part recomputed per group)
    const int nchunk = a.Cin / KC;
    if constexpr (SPLIT) {
        constexpr int KK = KC / 16;
        half8 ah[KK][CBW], al[KK][CBW], ahn[KK][CBW], aln[KK][CBW];
        auto load_weights_h = [&](int c0, half8 (&dh)[KK][CBW], half8 (&dl)[KK][CBW]) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) {
                    const int co = co_base + cb * 32 + r;
                    const half_t *row = a.wph + ((long)(co < a.Cout ? co : 0) * (a.Cin / 16) + (c0 / 16 + kk)) * 32 + 8 * hi;
                    dh[kk][cb] = *reinterpret_cast<const half8 *>(row);
                    dl[kk][cb] = *reinterpret_cast<const half8 *>(row + 16);
                    if (co >= a.Cout) {
                        dh[kk][cb] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                        dl[kk][cb] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                }
        };
        auto hb = [&](int c) { return reinterpret_cast<float *>(hbuf + (c & 1) * TP * ROWH); };
        if constexpr (PRE > 0) {
            static_assert(PRE % 2 == 0, "the LDS double buffer's slot must be a compile-time value");
            Taps ring[PRE][ITEMS];
#pragma unroll
            for (int k = 0; k < PRE; ++k)
                if (k < nchunk) dw_load(k * KC, ring[k]);
            load_weights_h(0, ah, al);
            dw_compute(0, ring[0], hb(0));
            if (PRE < nchunk) dw_load(PRE * KC, ring[0]);
            __syncthreads();
            for (int c0 = 0; c0 < nchunk; c0 += PRE) {  // (nchunk % PRE == 0: the launcher's condition)
#pragma unroll
                for (int d = 0; d < PRE; ++d) {
                    const int c = c0 + d;
                    const half_t *cur = hbuf + (d & 1) * TP * ROWH + (wp * 32 + r) * ROWH + 8 * hi;
                    if (c + 1 < nchunk) {
                        load_weights_h((c + 1) * KC, ahn, aln);
                        dw_compute((c + 1) * KC, ring[(d + 1) % PRE], hb(d + 1));
                        if (c + 1 + PRE < nchunk) dw_load((c + 1 + PRE) * KC, ring[(d + 1) % PRE]);
                    }
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        const half8 bh = *reinterpret_cast<const half8 *>(cur + kk * 16);
                        const half8 bl = *reinterpret_cast<const half8 *>(cur + KC + kk * 16);
#pragma unroll
                        for (int cb = 0; cb < CBW; ++cb) {
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk][cb], bh, acc[cb], 0, 0, 0);
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk][cb], bl, acc[cb], 0, 0, 0);
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk][cb], bh, acc[cb], 0, 0, 0);
                        }
                    }
                    if (c + 1 < nchunk) {
#pragma unroll
                        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                            for (int cb = 0; cb < CBW; ++cb) {
                                ah[kk][cb] = ahn[kk][cb];
                                al[kk][cb] = aln[kk][cb];
                            }
                    }
                    __syncthreads();
                }
            }
        } else {
        load_weights_h(0, ah, al);
        depthwise_chunk(0, hb(0));
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            const half_t *cur = hbuf + (c & 1) * TP * ROWH + (wp * 32 + r) * ROWH + 8 * hi;
            if (c + 1 < nchunk) {
                load_weights_h((c + 1) * KC, ahn, aln);
                depthwise_chunk((c + 1) * KC, hb(c + 1));
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const half8 bh = *reinterpret_cast<const half8 *>(cur + kk * 16);
                const half8 bl = *reinterpret_cast<const half8 *>(cur + KC + kk * 16);
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) {
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk][cb], bh, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk][cb], bl, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk][cb], bh, acc[cb], 0, 0, 0);
                }
            }
            if (c + 1 < nchunk) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int cb = 0; cb < CBW; ++cb) {
                        ah[kk][cb] = ahn[kk][cb];
                        al[kk][cb] = aln[kk][cb];
                    }
            }
            __syncthreads();
        }
        }
    } else {
    float areg[KS][CBW], anext[KS][CBW];
    auto load_weights = [&](int c0, float (&dst)[KS][CBW]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                const int co = co_base + cb * 32 + r;
                dst[ks][cb] = co < a.Cout ? a.wp[(long)(c0 + 2 * ks + hi) * a.Cout + co] : 0.f;
            }
    };

    load_weights(0, areg);
    depthwise_chunk(0, buf);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const float *cur = buf + (c & 1) * KC * TP;
        if (c + 1 < nchunk) {
            load_weights((c + 1) * KC, anext);
            depthwise_chunk((c + 1) * KC, buf + ((c + 1) & 1) * KC * TP);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float b = cur[(2 * ks + hi) * TP + wp * 32 + r];
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[ks][cb], b, acc[cb], 0, 0, 0);
        }
        if (c + 1 < nchunk) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) areg[ks][cb] = anext[ks][cb];
        }
        __syncthreads();
    }
    }

    // ---- epilogue: lane (r, hi) owns pixel r and channels cb*32 + (e&3) + 8*(e>>2) + 4*hi
    const PixMap mo = map_pixel<MODE2D, TP>(lid, wp * 32 + r, a.B, a.Ho, a.Wo, tiles_x, tiles_y);
    if (!mo.ok) return;
    const int HoWo = a.Ho * a.Wo;
    float *ob = a.out + (long)mo.b * a.Cout * HoWo + mo.oy * a.Wo + mo.ox;
    // Round 5: the biases in front of the first store.  For all the compiler knows the output aliases them: it emitted load -> s_waitcnt
    // vmcnt(0) -> store per channel, i.e. 16 CBW dependent memory round trips (each also waiting for the previous STORE to complete) at the end of
    // every workgroup - the largest part of this kernel's 68 % of wave cycles in s_waitcnt (profiles/r04/r04x_det_pmc.txt).
    float bq[CBW][16];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co_base + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            bq[cb][e] = a.bp[co < a.Cout ? co : 0];  // (clamped, unconditional: a conditional load is a branch and a basic block of its own)
        }
    if (co_base + CBW * 32 <= a.Cout) {  // (uniform; every shape of the network: straight-line stores, no per-channel branch)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co_base + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                float v = acc[cb][e] + bq[cb][e];
                if (a.relu) v = fmaxf(v, 0.f);
#ifdef FRT_ABLATE
                if ((abl & 1) && v != 12345.678f) continue;
#endif
                ob[(long)co * HoWo] = v;
            }
        return;
    }
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co_base + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (co < a.Cout) {
                float v = acc[cb][e] + bq[cb][e];
                if (a.relu) v = fmaxf(v, 0.f);
#ifdef FRT_ABLATE
                if ((abl & 1) && v != 12345.678f) continue;
#endif
                ob[(long)co * HoWo] = v;
            }
        }
}

// ---------------------------------------------------------------- plain 1x1 conv, one wave = 32 pixels x CBW*32 output channels
template <int CBW, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? (CBW == 1 ? 4 : 3) : 2) void pw_mfma_kernel(DwPwArgs a, int n_pix_groups) {  // (2nd: waves per SIMD asked of the register allocator)
    // Persistent waves (round 4): a wave lives ~ 4 us here, 70 % of it in s_waitcnt, and the counters put 1.5 waves per CU in flight on average
    // (profiles/r04/r04x_det_pmc.txt).  The grid is now what fits the chip at once and every wave walks its share of the (pixel group, channel
    // group) items; with the scalar-base addressing (240 -> 180 / 168 -> 104 registers) that is 61 -> 59 us for the 64-channel lateral at
    // 80x80 and 23 -> 18.5 us for the other two at 32 frames, 13 -> 11.6 us at 4 - the wide one is still not understood (1.9 TB/s, clean
    // 128-byte accesses).
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int n_cgroups = (a.Cout + CBW * 32 - 1) / (CBW * 32);
    const int n_items = n_pix_groups * n_cgroups;
    for (int gw = blockIdx.x * 4 + (threadIdx.x >> 6); gw < n_items; gw += gridDim.x * 4) {
    const int pg = gw / n_cgroups, cg = gw - pg * n_cgroups;
    const int HW = a.H * a.W;  // == Ho*Wo (stride 1)
    const long g = (long)pg * 32 + r;
    const bool ok = g < (long)a.B * HW;
    const int b = ok ? (int)(g / HW) : 0;
    const int p = ok ? (int)(g - (long)b * HW) : 0;
    const float *x = a.in + (long)b * a.Cin * HW + p + (long)hi * HW;  // channel hi of k-step 0
    const int co_base = cg * CBW * 32;
    const float *w = a.wp + (long)hi * a.Cout + co_base + r;

    floatx16 acc[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;

    if constexpr (SPLIT) {
        // fp16 hi/lo split (see dwpw_mfma_kernel): per 16 input channels a lane fetches its 8 channel values of the pixel (the same
        // 8 scalar loads the fp32 path spends on 8 k-steps), splits them, and three fp16 MFMAs replace eight fp32 ones
        const int ngroups = a.Cin / 16;
        // activations FOUR groups ahead (32 registers), weights (L2 hits) one: with one group in flight a wave paid the HBM latency once per 16
        // channels, and few waves fit a CU (round 4: lateral 64->64 at 80x80 61 -> see profiles/r04/r04v_laterals.txt)
        constexpr int DA = 4;
        float bx[DA][8];
        half8 ah[CBW], al[CBW], nah[CBW], nal[CBW];
        // addresses as (uniform 64-bit base) + (32-bit lane offset): the scalar-base form of global_load.  As per-lane 64-bit pointers the
        // compiler kept one register pair per load in flight and the kernel sat at 240 VGPRs = two waves per SIMD
        // Round 5: ONE uniform base and a RUNNING 32-bit lane offset, advanced by per-lane (opaque) strides.  Written as base + (uniform channel
        // offset) + lane offset the compiler formed one scalar 64-bit base per load - 128 pairs over the item, 207 scalar spills, and the
        // spill traffic (v_readlane / v_writelane) WAS the kernel: 1 245 VALU + 620 SALU instructions per 32-pixel item for ~ 300 useful ones
        // (profiles/r04/r04x_det_pmc.txt), 15 us per item.  The groups are requested in ascending order, so one running offset serves them all.
        const char *xin_c = reinterpret_cast<const char *>(a.in);
        unsigned st1, st9;  // one channel plane / nine (from channel 16 g + 8 hi + 7 to 16 (g + 1) + 8 hi)
        asm volatile("v_mov_b32 %0, %1" : "=v"(st1) : "s"(HW * 4));
        asm volatile("v_mov_b32 %0, %1" : "=v"(st9) : "s"(HW * 36));
        unsigned xo = (unsigned)((((long)b * a.Cin + 8 * hi) * HW + p) * 4);
        auto load_x = [&](int g, float (&bxv)[8]) {
            if (g >= ngroups) return;  // (uniform)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bxv[e] = ok ? *reinterpret_cast<const float *>(xin_c + (size_t)xo) : 0.f;
                xo += e == 7 ? st9 : st1;
            }
        };
        auto load_w = [&](int g, half8 (&dh)[CBW], half8 (&dl)[CBW]) {
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                const int co = co_base + cb * 32 + r;
                const bool wok = g < ngroups && co < a.Cout;
                const half_t *row = a.wph + ((long)(wok ? co : 0) * ngroups + (wok ? g : 0)) * 32 + 8 * hi;
                dh[cb] = *reinterpret_cast<const half8 *>(row);
                dl[cb] = *reinterpret_cast<const half8 *>(row + 16);
                if (!wok) {
                    dh[cb] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    dl[cb] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
        };
#pragma unroll
        for (int i = 0; i < DA; ++i) load_x(i, bx[i]);
        load_w(0, ah, al);
        for (int g0 = 0; g0 < ngroups; g0 += DA) {
#pragma unroll
            for (int i = 0; i < DA; ++i) {
                const int g = g0 + i;
                if (g >= ngroups) break;
                if (g + 1 < ngroups) load_w(g + 1, nah, nal);
                half8 bh, bl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const half_t xh = (half_t)bx[i][e];
                    bh[e] = xh;
                    bl[e] = (half_t)(bx[i][e] - (float)xh);
                }
                if (g + DA < ngroups) load_x(g + DA, bx[i]);
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) {
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bh, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bl, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], bh, acc[cb], 0, 0, 0);
                }
                if (g + 1 < ngroups) {
#pragma unroll
                    for (int cb = 0; cb < CBW; ++cb) {
                        ah[cb] = nah[cb];
                        al[cb] = nal[cb];
                    }
                }
            }
        }
    } else {
    constexpr int U = 8;  // k-steps in flight
    const int ksteps = a.Cin / 2;
    float bx[U], aw[U][CBW];
    auto load = [&](int ks0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = ks0 + u;
            bx[u] = (ok && ks < ksteps) ? x[(long)2 * ks * HW] : 0.f;
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
                aw[u][cb] = (ks < ksteps && co_base + cb * 32 + r < a.Cout) ? w[(long)2 * ks * a.Cout + cb * 32] : 0.f;
        }
    };
    load(0);
    for (int ks0 = 0; ks0 < ksteps; ks0 += U) {
        float cbx[U], caw[U][CBW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cbx[u] = bx[u];
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) caw[u][cb] = aw[u][cb];
        }
        if (ks0 + U < ksteps) load(ks0 + U);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(caw[u][cb], cbx[u], acc[cb], 0, 0, 0);
    }
    }

    if (!ok) continue;
    const int oy = p / a.W, ox = p - oy * a.W;
    const float *addb = nullptr;
    long add_cs = 0;
    if (a.add) {
        // F.interpolate(mode="nearest") to (Ho,Wo): src = min(floor(dst * (float)in/out), in-1)   (net.py:89,93)
        const float sh = (float)a.add_h / (float)a.Ho, sw = (float)a.add_w / (float)a.Wo;
        int ah = (int)floorf(oy * sh), awd = (int)floorf(ox * sw);
        ah = ah < a.add_h - 1 ? ah : a.add_h - 1;
        awd = awd < a.add_w - 1 ? awd : a.add_w - 1;
        add_cs = (long)a.add_h * a.add_w;
        addb = a.add + (long)b * a.Cout * add_cs + ah * a.add_w + awd;
    }
    // (addresses again as uniform base + 32-bit lane offset; every load of the epilogue before its first store: the output may alias the add
    //  source for all the compiler knows, left alone it serialises load -> store -> load)
    char *out_c = reinterpret_cast<char *>(a.out);
    const char *add_c = reinterpret_cast<const char *>(a.add), *bp_c = reinterpret_cast<const char *>(a.bp);
    const int co_l = co_base + 4 * hi;  // this lane's first channel; + cb * 32 + (e & 3) + 8 * (e >> 2)
    // running lane offsets (see load_x): channel steps of the epilogue's enumeration are +1, +1, +1, +5 planes
    unsigned oo = (unsigned)((((long)b * a.Cout + co_l) * HW + p) * 4);
    unsigned ao = addb ? (unsigned)(((addb - a.add) + (long)co_l * add_cs) * 4) : 0u;
    unsigned os1, os5, as1, as5;
    asm volatile("v_mov_b32 %0, %1" : "=v"(os1) : "s"(HW * 4));
    asm volatile("v_mov_b32 %0, %1" : "=v"(os5) : "s"(HW * 20));
    asm volatile("v_mov_b32 %0, %1" : "=v"(as1) : "s"((int)add_cs * 4));
    asm volatile("v_mov_b32 %0, %1" : "=v"(as5) : "s"((int)add_cs * 20));
    // Round 5: STRAIGHT-LINE epilogues.  With a per-channel "co < Cout" around every load and store each of them became a basic block of its own and
    // the compiler's waits degenerated to "s_waitcnt vmcnt(0)" in front of every store - 32 stores per item, each waiting for the previous one to
    // be acknowledged by memory: most of the 15 us an item took (profiles/r04/r04x_det_pmc.txt: 70 % of the wave cycles in s_waitcnt).  The channel
    // range of an item is whole for every shape of the network (uniform test), and "has an upsample-add source" is a compile-time variant.
    auto epilogue = [&](auto has_add_c, auto full_c) {
        constexpr bool HAS_ADD = decltype(has_add_c)::value, FULL = decltype(full_c)::value;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {  // (one 32-channel block at a time: 32 live values instead of 64)
            float bias_v[16], add_v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cu = cb * 32 + (e & 3) + 8 * (e >> 2);  // uniform part of the channel
                const bool cok = FULL || co_l + cu < a.Cout;
                bias_v[e] = *reinterpret_cast<const float *>(bp_c + (size_t)(cok ? cu : 0) * 4 + (size_t)(unsigned)((cok ? co_l : 0) * 4));  // (clamped, unconditional)
                if (HAS_ADD) add_v[e] = *reinterpret_cast<const float *>(add_c + (size_t)(cok ? ao : 0u));
                ao += (e & 3) == 3 ? as5 : as1;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cu = cb * 32 + (e & 3) + 8 * (e >> 2);
                float v = acc[cb][e] + bias_v[e];
                if (a.relu) v = fmaxf(v, 0.f);
                if (HAS_ADD) v += add_v[e];
                if (FULL || co_l + cu < a.Cout) *reinterpret_cast<float *>(out_c + (size_t)oo) = v;
                oo += (e & 3) == 3 ? os5 : os1;
            }
        }
    };
    const bool full = co_base + CBW * 32 <= a.Cout;
    if (addb) {
        if (full) epilogue(std::true_type{}, std::true_type{});
        else epilogue(std::true_type{}, std::false_type{});
    } else {
        if (full) epilogue(std::false_type{}, std::true_type{});
        else epilogue(std::false_type{}, std::false_type{});
    }
    }  // persistent item loop
}

template <int NPW, int NCW, int KC, int CBW, bool MODE2D>
void launch_fused(const DwPwArgs &a, hipStream_t s) {
    constexpr int TP = 32 * NPW;
    int tiles_x = 0, tiles_y = 0;
    long nblocks;
    if (MODE2D) {
        tiles_x = (a.Wo + 15) / 16;
        tiles_y = (a.Ho + 7) / 8;
        nblocks = (long)a.B * tiles_x * tiles_y;
    } else {
        nblocks = ((long)a.B * a.Ho * a.Wo + TP - 1) / TP;
    }
#ifdef FRT_ABLATE
    if (!MODE2D) tiles_x = frt_tuning_env("FRT_DWPW_ABLATE") ? atoi(frt_tuning_env("FRT_DWPW_ABLATE")) : 0;
#endif
    const unsigned cgroups = (unsigned)((a.Cout + 32 * NCW * CBW - 1) / (32 * NCW * CBW));
    const dim3 grid((unsigned)nblocks, cgroups);
    static const bool split = !(frt_tuning_env("FRT_DET_PW_SPLIT") && frt_tuning_env("FRT_DET_PW_SPLIT")[0] == '0');
    if constexpr (KC == 32) {  // (the 16-channel block with 128-pixel tiles measured slower split: 113 vs 108 us, scattered 2-byte LDS stores)
        if (split && a.wph) {  // pointwise product on the fp16 matrix cores (hi/lo split, fp32-class accuracy)
            const size_t ldsh = (size_t)2 * TP * (2 * KC + 8) * sizeof(half_t);
            // a frame or two (every workgroup alone on its CU): PRE chunks' loads in flight.  Measured per launch at one frame: see DESIGN 3.9
            static const int pre = frt_tuning_env("FRT_DWPW_PRE") ? atoi(frt_tuning_env("FRT_DWPW_PRE")) : 1;
            if constexpr (!MODE2D && NPW <= 2) {
                const int nchunk = a.Cin / KC;
                if (pre && nblocks * cgroups <= (pre == 2 ? 100000 : 256) && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0) {
                    if (nchunk % 4 == 0) {
                        if (a.stride == 1)
                            hipLaunchKernelGGL((dwpw_mfma_kernel<NPW, NCW, KC, CBW, 1, MODE2D, true, 4>), grid, dim3(64 * NPW * NCW), ldsh, s, a, tiles_x, tiles_y);
                        else
                            hipLaunchKernelGGL((dwpw_mfma_kernel<NPW, NCW, KC, CBW, 2, MODE2D, true, 4>), grid, dim3(64 * NPW * NCW), ldsh, s, a, tiles_x, tiles_y);
                        return;
                    }
                    if (nchunk % 2 == 0) {
                        if (a.stride == 1)
                            hipLaunchKernelGGL((dwpw_mfma_kernel<NPW, NCW, KC, CBW, 1, MODE2D, true, 2>), grid, dim3(64 * NPW * NCW), ldsh, s, a, tiles_x, tiles_y);
                        else
                            hipLaunchKernelGGL((dwpw_mfma_kernel<NPW, NCW, KC, CBW, 2, MODE2D, true, 2>), grid, dim3(64 * NPW * NCW), ldsh, s, a, tiles_x, tiles_y);
                        return;
                    }
                }
            }
            if (a.stride == 1)
                hipLaunchKernelGGL((dwpw_mfma_kernel<NPW, NCW, KC, CBW, 1, MODE2D, true>), grid, dim3(64 * NPW * NCW), ldsh, s, a, tiles_x, tiles_y);
            else
                hipLaunchKernelGGL((dwpw_mfma_kernel<NPW, NCW, KC, CBW, 2, MODE2D, true>), grid, dim3(64 * NPW * NCW), ldsh, s, a, tiles_x, tiles_y);
            return;
        }
    }
    const size_t lds = (2 * (size_t)KC * TP) * sizeof(float);
    if (a.stride == 1)
        hipLaunchKernelGGL((dwpw_mfma_kernel<NPW, NCW, KC, CBW, 1, MODE2D>), grid, dim3(64 * NPW * NCW), lds, s, a, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL((dwpw_mfma_kernel<NPW, NCW, KC, CBW, 2, MODE2D>), grid, dim3(64 * NPW * NCW), lds, s, a, tiles_x, tiles_y);
}

}  // namespace

// Returns false when the shape is outside what the MFMA kernels cover (the caller falls back to the scalar kernels).
bool launch_dwpw_mfma(const DwPwArgs &a, hipStream_t s) {
    if ((a.Cin & 1) || a.Cout < 16) return false;
    if (!a.wd) {
        if (a.stride != 1 || a.H != a.Ho || a.W != a.Wo) return false;
        const long total = (long)a.B * a.H * a.W;
        const int n_pix_groups = (int)((total + 31) / 32);
        // one wave covers all output channels while that still fills the chip, otherwise 32 channels per wave
        const bool wide = a.Cout >= 64 && (long)n_pix_groups >= 2048;
        const int cbw = wide ? 2 : 1;
        const int n_cgroups = (a.Cout + cbw * 32 - 1) / (cbw * 32);
        const long waves = (long)n_pix_groups * n_cgroups;
        // (persistent: two 256-thread workgroups per CU for the wide variant (180 registers), four for the narrow one (104))
        const unsigned cap = wide ? 3 * 256 : 4 * 256;  // (round 5: 154 / 114 registers)
        const unsigned grid = (unsigned)std::min<long>((waves + 3) / 4, cap);
        static const bool split = !(frt_tuning_env("FRT_DET_PW_SPLIT") && frt_tuning_env("FRT_DET_PW_SPLIT")[0] == '0');
        if (split && a.wph && a.Cin % 16 == 0) {
            if (wide) hipLaunchKernelGGL((pw_mfma_kernel<2, true>), dim3(grid), dim3(256), 0, s, a, n_pix_groups);
            else hipLaunchKernelGGL((pw_mfma_kernel<1, true>), dim3(grid), dim3(256), 0, s, a, n_pix_groups);
            return true;
        }
        if (wide) hipLaunchKernelGGL((pw_mfma_kernel<2>), dim3(grid), dim3(256), 0, s, a, n_pix_groups);
        else hipLaunchKernelGGL((pw_mfma_kernel<1>), dim3(grid), dim3(256), 0, s, a, n_pix_groups);
        return true;
    }
    if (a.add || (a.stride != 1 && a.stride != 2) || !a.wd12) return false;
    if ((a.Wo & 3) || a.W != a.stride * a.Wo) return false;                           // 4-pixel segments, aligned float4 rows
    if (a.stride == 1 ? a.H != a.Ho : (a.H + 1) / 2 != a.Ho) return false;
    if ((reinterpret_cast<uintptr_t>(a.in) & 15) || ((a.H * a.W) & 3)) return false;
    // Measured per block at batch 32 (us, scalar fused kernel vs this one): 8->16 @320^2 112 vs 152, 32->32 @160^2 98 vs 118 -
    // the stride-1 blocks with <= 32 channels are pure stencil + HBM traffic and the per-tile LDS round trip only costs;
    // everything else is faster here (16->32/2: 145 vs 100, 64->64 @80^2: 169 vs 70, 128->128 @40^2: 77 vs 59).
    // (A persistent, 3-stage software-pipelined variant of this kernel was also tried: slower on every block - fewer, fatter
    // waves hide the load latency worse than three small resident workgroups per CU do.)
    if (launch_dwpw_wave(a, s)) return true;
    if (a.Cout <= 32 && a.Cin <= 32 && a.stride == 1 && !frt_tuning_env("FRT_DWPW_FORCE_MFMA")) return false;
    const long total = (long)a.B * a.Ho * a.Wo;
    const bool big = a.Wo >= 64 && (a.Wo % 16) == 0 && (a.Ho % 8) == 0;
    if (a.Cout <= 32) {
        if (a.Cin == 8) big ? launch_fused<4, 1, 8, 1, true>(a, s) : launch_fused<4, 1, 8, 1, false>(a, s);
        else if (a.Cin == 16) big ? launch_fused<4, 1, 16, 1, true>(a, s) : launch_fused<4, 1, 16, 1, false>(a, s);
        else if (a.Cin % 32 == 0) big ? launch_fused<4, 1, 32, 1, true>(a, s) : launch_fused<4, 1, 32, 1, false>(a, s);
        else return false;
        return true;
    }
    if (a.Cin % 32) return false;
    if (a.Cout == 64 && total >= 128L * 512) {
        static const int c64 = frt_tuning_env("FRT_DWPW_C64") ? atoi(frt_tuning_env("FRT_DWPW_C64")) : 1;  // measured: 64-px linear tiles, 2x2 waves: 46/60 us vs 54/71 for 8x16 tiles
        if (c64 == 1) launch_fused<2, 2, 32, 1, false>(a, s);
        else if (c64 == 2) launch_fused<1, 2, 32, 1, false>(a, s);
        else if (c64 == 3) big ? launch_fused<4, 1, 16, 2, true>(a, s) : launch_fused<4, 1, 16, 2, false>(a, s);
        else big ? launch_fused<4, 1, 32, 2, true>(a, s) : launch_fused<4, 1, 32, 2, false>(a, s);
    } else if (a.Cout == 128 || (a.Cout == 64)) {
        if (a.Cout == 128) {
            static const int small = frt_tuning_env("FRT_DWPW_SMALL") ? atoi(frt_tuning_env("FRT_DWPW_SMALL")) : 1;
            if (small == 1) launch_fused<1, 4, 32, 1, false>(a, s);  // (KC = 64, half as many rounds, measured 52 vs 46 us: registers)
            else if (small == 2) launch_fused<1, 2, 32, 2, false>(a, s);
            else if (small == 3) launch_fused<1, 2, 32, 1, false>(a, s);
            else launch_fused<2, 2, 32, 2, false>(a, s);
        }
        else launch_fused<2, 2, 32, 1, false>(a, s);
    } else if (a.Cout == 256) {
        static const int small = frt_tuning_env("FRT_DWPW_SMALL") ? atoi(frt_tuning_env("FRT_DWPW_SMALL")) : 1;
        if (small == 2) launch_fused<1, 2, 32, 2, false>(a, s);
        else if (small == 3 || total <= 2048) launch_fused<1, 2, 32, 1, false>(a, s);  // a frame or two: twice the workgroups (17 + 24 -> 12 + 18 us)
        else launch_fused<1, 4, 32, 2, false>(a, s);
    } else {
        return false;
    }
    return true;
}
