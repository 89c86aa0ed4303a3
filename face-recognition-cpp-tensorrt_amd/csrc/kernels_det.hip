// RetinaFace-mobilenet0.25 forward (fp32, NCHW) as hand-written HIP kernels for gfx950.
//
// Arithmetic spec: /root/reference/conversion/retina/models/net.py:9-38 (conv_bn / conv_dw blocks), :40-66 (SSH), :68-98
// (FPN), :102-124 (MobileNetV1 stages) and retinaface_trim.py:14-35,107-127 (heads, concat, softmax).  In the reference
// this network runs inside a TensorRT engine (src/retinaface.cpp:141).
//
// The detector is ~35 flop/byte - below the machine balance - so no matrix cores (north_star agrees).  What matters:
//   * planar NCHW, one thread per output pixel (or 2-4 pixels): for every input channel the 64 lanes of a wave read 64
//     consecutive pixels (256 contiguous bytes) and write 64 consecutive pixels per output channel;
//   * weights never touch the vector memory path in the inner loop: they are staged once per workgroup into LDS as one
//     record per input channel and read back as wave-uniform (broadcast, conflict-free) ds_read_b128, or - in the dense 3x3
//     kernel - fetched on the scalar path (s_load -> SGPR operand of v_fma_f32);
//   * BatchNorm is folded into the preceding conv on the host (all detector BNs follow their conv); ReLU, the FPN
//     nearest-upsample+add, the SSH concat (+ its ReLU) and the head permute/softmax are fused into the producing kernel;
//   * early conv_dw blocks (few channels, memory-bound) are ONE fused kernel: the depthwise value lives in a register and
//     feeds CT pointwise accumulators.  Late blocks (Cout > one channel tile) would recompute the depthwise part once per
//     output-channel tile, so they run as depthwise kernel + register-blocked pointwise kernel (PPT pixels x 32 couts per
//     thread: 4*PPT..8*PPT FMAs per LDS/global read) - the round trip of the small intermediate is cheaper than the recompute;
//   * the three pyramid levels of every SSH / head op are launched together (blockIdx.z = level): the 40x40 and 20x20 levels
//     are pure latency on their own and hide inside the 80x80 level's launch.
#include <cstdlib>

#include <type_traits>

#include "frt_kernels.h"

namespace {

// ---------------------------------------------------------------- fused depthwise3x3(+ReLU) -> pointwise1x1 (+ReLU) (+upsample-add)
// LDS record per input channel: [9 dw taps, dw bias, 2 pad][CT pointwise weights].  HAS_DW=false: plain 1x1 conv.
template <int CT, bool HAS_DW, int PPT>
__global__ __launch_bounds__(256) void dwpw_kernel(DwPwArgs a) {
    constexpr int REC = (HAS_DW ? 12 : 0) + CT;  // floats per input-channel record (16-byte multiple)
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int co0 = blockIdx.y * CT;
    for (int i = threadIdx.x; i < a.Cin * REC; i += 256) {
        const int ci = i / REC, r = i - ci * REC;
        float v = 0.f;
        if (HAS_DW && r < 9) v = a.wd[ci * 9 + r];
        else if (HAS_DW && r == 9) v = a.bd[ci];
        else if (r >= (HAS_DW ? 12 : 0)) v = a.wp[(long)ci * a.Cout + co0 + (r - (HAS_DW ? 12 : 0))];
        wsm[i] = v;
    }
    __syncthreads();

    const int HoWo = a.Ho * a.Wo;
    const long total = (long)a.B * HoWo;
    const long span = (total + PPT - 1) / PPT;
    const long g0 = (long)blockIdx.x * 256 + threadIdx.x;
    const int HW = a.H * a.W;

    float acc[PPT][CT];
    const float *inb[PPT];
    int off[PPT][HAS_DW ? 9 : 1];
    bool ok[PPT][HAS_DW ? 9 : 1];
    bool live[PPT];
    int pb[PPT], pp[PPT], poh[PPT], pow_[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const long gp = g0 + q * span;
        live[q] = g0 < span && gp < total;
        const long gq = live[q] ? gp : 0;
        pb[q] = (int)(gq / HoWo);
        pp[q] = (int)(gq - (long)pb[q] * HoWo);
        poh[q] = pp[q] / a.Wo;
        pow_[q] = pp[q] - poh[q] * a.Wo;
        inb[q] = a.in + (long)pb[q] * a.Cin * HW;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[q][c] = 0.f;
        if (HAS_DW) {
            const int ih0 = poh[q] * a.stride - 1, iw0 = pow_[q] * a.stride - 1;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ih = ih0 + kh, iw = iw0 + kw;
                    ok[q][kh * 3 + kw] = live[q] && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                    off[q][kh * 3 + kw] = ok[q][kh * 3 + kw] ? ih * a.W + iw : 0;
                }
        } else {
            ok[q][0] = live[q];
            off[q][0] = (poh[q] * a.stride) * a.W + pow_[q] * a.stride;
        }
    }

    // plain 1x1 path: unrolled so that the independent global loads of several input channels are in flight together (the loop
    // is otherwise one exposed memory latency per channel: few waves per SIMD at these grid sizes); measured 57 -> 45 us on the
    // 40x40 layers.  The fused depthwise path is NOT unrolled (measured slower: register pressure).
#pragma unroll(HAS_DW ? 1 : (PPT >= 4 ? 4 : 8))
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float *rec = wsm + ci * REC;
        float d[PPT];
        if (HAS_DW) {
            const floatx4 w0 = *reinterpret_cast<const floatx4 *>(rec);
            const floatx4 w1 = *reinterpret_cast<const floatx4 *>(rec + 4);
            const floatx4 w2 = *reinterpret_cast<const floatx4 *>(rec + 8);
            const float wt[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const float *x = inb[q] + (long)ci * HW;
                float v[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) v[t] = ok[q][t] ? x[off[q][t]] : 0.f;
                float s = w2[1];
#pragma unroll
                for (int t = 0; t < 9; ++t) s = fmaf(v[t], wt[t], s);
                d[q] = fmaxf(s, 0.f);
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q) d[q] = inb[q][(long)ci * HW + off[q][0]];
        }
        const float *wp = rec + (HAS_DW ? 12 : 0);
#pragma unroll
        for (int c4 = 0; c4 < CT; c4 += 4) {
            const floatx4 w = *reinterpret_cast<const floatx4 *>(wp + c4);
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                acc[q][c4] = fmaf(d[q], w[0], acc[q][c4]);
                acc[q][c4 + 1] = fmaf(d[q], w[1], acc[q][c4 + 1]);
                acc[q][c4 + 2] = fmaf(d[q], w[2], acc[q][c4 + 2]);
                acc[q][c4 + 3] = fmaf(d[q], w[3], acc[q][c4 + 3]);
            }
        }
    }

#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (!live[q]) continue;
        float *ob = a.out + ((long)pb[q] * a.Cout + co0) * HoWo + pp[q];
        const float *addb = nullptr;
        if (a.add) {
            // F.interpolate(mode="nearest") to (Ho,Wo): src = min(floor(dst * (float)in/out), in-1)   (net.py:89,93)
            const float sh = (float)a.add_h / (float)a.Ho, sw = (float)a.add_w / (float)a.Wo;
            int ah = (int)floorf(poh[q] * sh), aw = (int)floorf(pow_[q] * sw);
            ah = ah < a.add_h - 1 ? ah : a.add_h - 1;
            aw = aw < a.add_w - 1 ? aw : a.add_w - 1;
            addb = a.add + ((long)pb[q] * a.Cout + co0) * a.add_h * a.add_w + ah * a.add_w + aw;
        }
        float bq[CT], aq[CT];  // (all loads in front of the first store: see dwpw_row4_kernel's epilogue)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            bq[c] = a.bp[co0 + c];
            aq[c] = addb ? addb[(long)c * a.add_h * a.add_w] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            float v = acc[q][c] + bq[c];
            if (a.relu) v = fmaxf(v, 0.f);
            if (addb) v += aq[c];
            ob[(long)c * HoWo] = v;
        }
    }
}

// ---------------------------------------------------------------- fused conv_dw block, stride 1, four output pixels of a row per thread
// The kernel above issues 9 (exec-masked) scalar loads per pixel and input channel.  With four consecutive pixels per thread
// a channel costs three rows x (one aligned float4 + the two neighbours) = 9 loads for FOUR pixels, all unconditional from
// clamped addresses (zeroing deferred to a select), the next channel's nine loads are in flight while the current one is
// consumed, and the outputs leave as float4 stores.  Same arithmetic and summation order as dwpw_kernel (bit-identical results).
// Needs W % 4 == 0 (the 320x320 and 160x160 blocks of the 640x640 network).
// NS = slots of the input prefetch ring (NS - 1 channels in flight ahead of the one being consumed): at 176-240 registers only two
// waves fit a SIMD, so the bytes in flight per CU come from the ring depth (one channel ahead = 37 KB per CU: 2.3 TB/s on the 8 -> 16
// block at 320x320).
// MF (round 6, CT = 32 with two channels per ring step): the pointwise chain - 128 of the ~ 190 vector instructions per input channel - runs on the
// matrix pipe: v_mfma_f32_4x4x1_16B_f32 with CBSZ = 4 is acc[i] += w[ABID][i] * x per lane (kernels_det_stem.hip), one vector register holds the
// 16 weight quads of two input channels (read from LDS once per ring round), pixel q's accumulators are 8 quads.  Same chain order (cin
// ascending) = same bits as the fmaf chain.
template <int I, int N, class F>
__device__ __forceinline__ void static_for_dw(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_dw<I + 1, N>(f);
    }
}
template <int CT, int NS = 2, bool MF = false>
__global__ __launch_bounds__(256, MF ? 2 : 1) void dwpw_row4_kernel(DwPwArgs a) {
    static_assert(!MF || (CT == 32 && NS == 2), "matrix-pipe pointwise: 8 cout quads x 2 channels = the 16 blocks of one A register");
    constexpr int REC = 12 + CT;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int co0 = blockIdx.y * CT;  // (grid.y > 1: a frame or two, narrower channel tiles = more waves; the depthwise part is recomputed per tile)
    for (int i = threadIdx.x; i < a.Cin * REC; i += 256) {
        const int ci = i / REC, r = i - ci * REC;
        float v = 0.f;
        if (r < 9) v = a.wd[ci * 9 + r];
        else if (r == 9) v = a.bd[ci];
        else if (r >= 12) v = a.wp[(long)ci * a.Cout + co0 + (r - 12)];
        wsm[i] = v;
    }
    if constexpr (MF) {  // A registers: [channel pair][lane]: lane l = block l / 4 = (channel l / 32, cout quad (l / 4) % 8), element l & 3
        float *wa = wsm + a.Cin * REC;
        for (int i = threadIdx.x; i < a.Cin * 32; i += 256) {
            const int pr = i >> 6, l = i & 63;
            wa[i] = a.wp[(long)(2 * pr + (l >> 5)) * a.Cout + co0 + ((l >> 2) & 7) * 4 + (l & 3)];
        }
    }
    __syncthreads();
    const int W4 = a.W >> 2, HW = a.H * a.W;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long)a.B * a.H * W4) return;
    const int b = (int)(g / (a.H * W4)), rem = (int)(g - (long)b * (a.H * W4));
    const int oh = rem / W4, ow0 = (rem - oh * W4) * 4;
    const float *inb = a.in + (long)b * a.Cin * HW;
    // rows oh-1, oh, oh+1 (clamped; rmask zeroes the ones outside), columns ow0-1 .. ow0+4 (neighbours clamped, masked)
    int roff[3];
    float rmask[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh - 1 + kh;
        const bool ok = ih >= 0 && ih < a.H;
        roff[kh] = (ok ? ih : oh) * a.W + ow0;
        rmask[kh] = ok ? 1.f : 0.f;
    }
    const bool lok = ow0 > 0, rok = ow0 + 4 < a.W;
    const int loff = lok ? -1 : 0, roff_r = rok ? 4 : 3;

    float acc[MF ? 1 : 4][MF ? 1 : CT];
    floatx4 acc4[MF ? 4 : 1][MF ? CT / 4 : 1];
#pragma unroll
    for (int q = 0; q < (MF ? 1 : 4); ++q)
#pragma unroll
        for (int c = 0; c < (MF ? 1 : CT); ++c) acc[q][c] = 0.f;
#pragma unroll
    for (int q = 0; q < (MF ? 4 : 1); ++q)
#pragma unroll
        for (int c = 0; c < (MF ? CT / 4 : 1); ++c) acc4[q][c] = floatx4{0.f, 0.f, 0.f, 0.f};

    floatx4 mid[NS][3];
    float lft[NS][3], rgt[NS][3];
    auto fetch = [&](int ci, int slot) {
        const float *x = inb + (long)ci * HW;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            mid[slot][kh] = *reinterpret_cast<const floatx4 *>(x + roff[kh]);
            lft[slot][kh] = x[roff[kh] + loff];
            rgt[slot][kh] = x[roff[kh] + roff_r];
        }
    };
#pragma unroll
    for (int sl = 0; sl < NS - 1; ++sl)
        if (sl < a.Cin) fetch(sl, sl);
    for (int ci = 0; ci < a.Cin; ci += NS) {
        float wA = 0.f;
        if constexpr (MF) wA = wsm[a.Cin * REC + (ci >> 1) * 64 + (threadIdx.x & 63)];
#pragma unroll
        for (int half = 0; half < NS; ++half) {  // (compile-time ring slot)
            const int c_ = ci + half;
            if (c_ >= a.Cin) break;
            if (c_ + NS - 1 < a.Cin) fetch(c_ + NS - 1, (half + NS - 1) % NS);
            const float *rec = wsm + c_ * REC;
            const floatx4 w0 = *reinterpret_cast<const floatx4 *>(rec);
            const floatx4 w1 = *reinterpret_cast<const floatx4 *>(rec + 4);
            const floatx4 w2 = *reinterpret_cast<const floatx4 *>(rec + 8);
            const float wt[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
            float v[3][6];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                v[kh][0] = lok ? lft[half][kh] * rmask[kh] : 0.f;
                v[kh][1] = mid[half][kh][0] * rmask[kh];
                v[kh][2] = mid[half][kh][1] * rmask[kh];
                v[kh][3] = mid[half][kh][2] * rmask[kh];
                v[kh][4] = mid[half][kh][3] * rmask[kh];
                v[kh][5] = rok ? rgt[half][kh] * rmask[kh] : 0.f;
            }
            float d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sacc = w2[1];
#pragma unroll
                for (int t = 0; t < 9; ++t) sacc = fmaf(v[t / 3][q + t % 3], wt[t], sacc);
                d[q] = fmaxf(sacc, 0.f);
            }
            if constexpr (MF) {
                static_for_dw<0, 8>([&](auto cc) {
                    constexpr int cq = decltype(cc)::value;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (half == 0) acc4[q][cq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA, d[q], acc4[q][cq], 4, cq, 0);
                        else acc4[q][cq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA, d[q], acc4[q][cq], 4, 8 + cq, 0);
                    }
                });
            } else {
                const float *wp = rec + 12;
#pragma unroll
                for (int c4 = 0; c4 < CT; c4 += 4) {
                    const floatx4 w = *reinterpret_cast<const floatx4 *>(wp + c4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[q][c4] = fmaf(d[q], w[0], acc[q][c4]);
                        acc[q][c4 + 1] = fmaf(d[q], w[1], acc[q][c4 + 1]);
                        acc[q][c4 + 2] = fmaf(d[q], w[2], acc[q][c4 + 2]);
                        acc[q][c4 + 3] = fmaf(d[q], w[3], acc[q][c4 + 3]);
                    }
                }
            }
        }
    }
    float *ob = a.out + ((long)b * a.Cout + co0) * HW + oh * a.W + ow0;
    // (every parameter load of the epilogue in front of its first store: for all the compiler knows the output aliases them, and left alone it
    //  emits load -> s_waitcnt vmcnt(0) -> store once per channel - a chain of dependent memory round trips at the end of every thread; round 5)
    float bq[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bq[c] = a.bp[co0 + c];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        floatx4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = (MF ? acc4[MF ? q : 0][MF ? c >> 2 : 0][c & 3] : acc[MF ? 0 : q][MF ? 0 : c]) + bq[c];
            if (a.relu) v = fmaxf(v, 0.f);
            o[q] = v;
        }
        *reinterpret_cast<floatx4 *>(ob + (long)c * HW) = o;
    }
}

// ---------------------------------------------------------------- fused conv_dw block for a frame or two: thread = pixel, branch-free taps
// dwpw_row4_kernel at one frame of 160x160 is 25 workgroups - one wave on a tenth of the SIMDs, each issuing its 6 400 instructions alone
// (29 us); dwpw_kernel has the threads but its nine exec-masked loads per channel compile to nine branches (15 us, 26 per frame from two
// frames on).  Here the taps are BUFFER loads - a tap outside the image gets an offset past the buffer's size and reads as 0, no branch, no
// select - the input channel is the scalar offset, UNR channels' taps (9 * UNR loads) are in flight ahead of the channel group being
// consumed, and the output channels are split into CT-wide tiles over grid.y (depthwise part recomputed per tile: the chip is empty anyway).
// Same chain per output as the other two kernels (dw bias, nine fmaf, ReLU; then one fmaf per input channel in channel order): bit-identical.
template <int CT, int UNR>
__global__ __launch_bounds__(256) void dwpw_pixs_kernel(DwPwArgs a) {
    constexpr int REC = 12 + CT;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int co0 = blockIdx.y * CT;
    for (int i = threadIdx.x; i < a.Cin * REC; i += 256) {
        const int ci = i / REC, r = i - ci * REC;
        float v = 0.f;
        if (r < 9) v = a.wd[ci * 9 + r];
        else if (r == 9) v = a.bd[ci];
        else if (r >= 12) v = a.wp[(long)ci * a.Cout + co0 + (r - 12)];
        wsm[i] = v;
    }
    __syncthreads();
    const int HoWo = a.Ho * a.Wo, HW = a.H * a.W;
    const long total = (long)a.B * HoWo;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const long gq = g < total ? g : total - 1;  // (the last workgroup's spare threads redo the last pixel and do not store)
    const int b = (int)(gq / HoWo), p = (int)(gq - (long)b * HoWo);
    const int oh = p / a.Wo, ow = p - oh * a.Wo;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, (int)((long)a.B * a.Cin * HW * 4), 0x00020000);
    int vo[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ih = oh * a.stride - 1 + t / 3, iw = ow * a.stride - 1 + t % 3;
        vo[t] = (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) ? (int)((((long)b * a.Cin * a.H + ih) * a.W + iw) * 4) : (int)0x80000000;
    }
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    float v[2][UNR][9];
    auto fetch = [&](int ci0, int slot) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int t = 0; t < 9; ++t) v[slot][u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[t], (ci0 + u) * HW * 4, 0));
    };
    auto consume = [&](int ci0, int slot) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const float *rec = wsm + (ci0 + u) * REC;
            const floatx4 w0 = *reinterpret_cast<const floatx4 *>(rec);
            const floatx4 w1 = *reinterpret_cast<const floatx4 *>(rec + 4);
            const floatx4 w2 = *reinterpret_cast<const floatx4 *>(rec + 8);
            const float wt[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
            float sacc = w2[1];
#pragma unroll
            for (int t = 0; t < 9; ++t) sacc = fmaf(v[slot][u][t], wt[t], sacc);
            const float d = fmaxf(sacc, 0.f);
#pragma unroll
            for (int c4 = 0; c4 < CT; c4 += 4) {
                const floatx4 w = *reinterpret_cast<const floatx4 *>(rec + 12 + c4);
                acc[c4] = fmaf(d, w[0], acc[c4]);
                acc[c4 + 1] = fmaf(d, w[1], acc[c4 + 1]);
                acc[c4 + 2] = fmaf(d, w[2], acc[c4 + 2]);
                acc[c4 + 3] = fmaf(d, w[3], acc[c4 + 3]);
            }
        }
    };
    fetch(0, 0);
    for (int ci = 0; ci < a.Cin; ci += 2 * UNR) {  // (Cin % (2 * UNR) == 0: checked by the launcher)
        fetch(ci + UNR, 1);
        consume(ci, 0);
        if (ci + 2 * UNR < a.Cin) fetch(ci + 2 * UNR, 0);
        consume(ci + UNR, 1);
    }
    if (g >= total) return;
    float *ob = a.out + ((long)b * a.Cout + co0) * HoWo + p;
    float bq[CT];  // (loads in front of the first store: see dwpw_row4_kernel's epilogue)
#pragma unroll
    for (int c = 0; c < CT; ++c) bq[c] = a.bp[co0 + c];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float o = acc[c] + bq[c];
        if (a.relu) o = fmaxf(o, 0.f);
        ob[(long)c * HoWo] = o;
    }
}

// ---------------------------------------------------------------- depthwise 3x3 + bias + ReLU (split path), thread = (b, c, pixel)
__global__ __launch_bounds__(256) void dw_kernel(DwPwArgs a) {
    const int HoWo = a.Ho * a.Wo;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long)a.B * a.Cin * HoWo) return;
    const int bc = (int)(g / HoWo), p = (int)(g - (long)bc * HoWo);
    const int c = bc % a.Cin;
    const int oh = p / a.Wo, ow = p - oh * a.Wo;
    const float *x = a.in + (long)bc * a.H * a.W;
    const float *w = a.wd + c * 9;  // NOT wave-uniform (a wave may straddle channels): ordinary cached loads
    float s = a.bd[c];
    const int ih0 = oh * a.stride - 1, iw0 = ow * a.stride - 1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = ih0 + kh, iw = iw0 + kw;
            const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            const float v = ok ? x[ih * a.W + iw] : 0.f;
            s = fmaf(v, w[kh * 3 + kw], s);
        }
    a.tmp[g] = fmaxf(s, 0.f);
}

// ---------------------------------------------------------------- dense 3x3 (pad 1) + bias (+ReLU), writes a channel slice
// Up to 3 independent problems per launch (blockIdx.z): the pyramid levels of one SSH op.
// (A four-pixels-per-thread variant with the weights in LDS, the recipe of dwpw_row4_kernel, measured 74 / 50 us against 54 / 32 us
//  here on the 16-channel SSH convs: a quarter of the threads leaves about one wave per SIMD on these small maps.)
struct Conv3Multi {
    Conv3Args p[3];
};
template <int CT>
__global__ __launch_bounds__(256) void conv3x3_kernel(Conv3Multi mm) {
    const Conv3Args &a = mm.p[blockIdx.z];
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    const int HoWo = a.Ho * a.Wo;
    if (gp >= (long)a.B * HoWo || (int)blockIdx.y * CT >= a.Cout) return;
    const int b = (int)(gp / HoWo), p = (int)(gp - (long)b * HoWo);
    const int oh = p / a.Wo, ow = p - oh * a.Wo;
    const int co0 = blockIdx.y * CT;
    const int HW = a.H * a.W;

    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    const int ih0 = oh * a.stride - 1, iw0 = ow * a.stride - 1;
    int off[9];
    bool ok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = ih0 + kh, iw = iw0 + kw;
            ok[kh * 3 + kw] = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            off[kh * 3 + kw] = ok[kh * 3 + kw] ? ih * a.W + iw : 0;
        }
    const float *inb = a.in + (long)b * a.Cin * HW;
    for (int ci = 0; ci < a.Cin; ++ci) {  // (unrolling this loop measured 1.8x SLOWER: the scalar weight fetches serialise)
        const float *x = inb + (long)ci * HW;
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = ok[t] ? x[off[t]] : 0.f;
        const float *w = a.w + ((long)ci * 9) * a.Cout + co0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = fmaf(v[t], w[t * a.Cout + c], acc[c]);
    }
    // channel tiles at or beyond `split` belong to the second output tensor (two convs that share their input, run as one)
    float *ob = (a.out2 && co0 >= a.split) ? a.out2 + ((long)b * a.out2_ctotal + a.out2_coff + co0 - a.split) * HoWo + p
                                           : a.out + ((long)b * a.out_ctotal + a.out_coff + co0) * HoWo + p;
    float bq[CT];  // (loads in front of the first store: see dwpw_row4_kernel's epilogue)
#pragma unroll
    for (int c = 0; c < CT; ++c) bq[c] = a.b[co0 + c];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float v = acc[c] + bq[c];
        if (a.relu) v = fmaxf(v, 0.f);
        ob[(long)c * HoWo] = v;
    }
}

// ---------------------------------------------------------------- the same conv for a frame or two: weights through LDS, UNR channels' taps in flight
// conv3x3_kernel fetches its 144 weights per input channel through the scalar cache, in register-sized batches that wait on each other: at
// 32 frames other waves hide that, at one frame (8 400 pixels on three levels = 1 wave per SIMD) it IS the kernel: 22 us for 39 MFLOP.  Here the
// workgroup copies its (level, channel tile)'s weights to LDS once ([Cin * 9][CT], 9 KB), reads them back as broadcast ds_read_b128 and keeps
// the taps of UNR input channels in flight.  Same products in the same order (ci outer, tap inner, one fmaf each): bit-identical results.
template <int CT, int UNR>
__global__ __launch_bounds__(256) void conv3x3_ldsw_kernel(Conv3Multi mm) {
    const Conv3Args &a = mm.p[blockIdx.z];
    const int HoWo = a.Ho * a.Wo;
    const long total = (long)a.B * HoWo;
    if ((long)blockIdx.x * 256 >= total || (int)blockIdx.y * CT >= a.Cout) return;  // (uniform over the workgroup)
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int co0 = blockIdx.y * CT;
    for (int i = threadIdx.x; i < a.Cin * 9 * (CT / 4); i += 256) {
        const int row = i / (CT / 4), c4 = i - row * (CT / 4);
        *reinterpret_cast<floatx4 *>(wsm + row * CT + c4 * 4) = *reinterpret_cast<const floatx4 *>(a.w + (long)row * a.Cout + co0 + c4 * 4);
    }
    __syncthreads();
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    if (gp >= total) return;
    const int b = (int)(gp / HoWo), p = (int)(gp - (long)b * HoWo);
    const int oh = p / a.Wo, ow = p - oh * a.Wo;
    const int HW = a.H * a.W;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    const int ih0 = oh * a.stride - 1, iw0 = ow * a.stride - 1;
    int off[9];
    bool ok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = ih0 + kh, iw = iw0 + kw;
            ok[kh * 3 + kw] = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            off[kh * 3 + kw] = ok[kh * 3 + kw] ? ih * a.W + iw : 0;
        }
    const float *inb = a.in + (long)b * a.Cin * HW;
    for (int ci = 0; ci < a.Cin; ci += UNR) {
        float v[UNR][9];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const float *x = inb + (long)(ci + u) * HW;
#pragma unroll
            for (int t = 0; t < 9; ++t) v[u][t] = ok[t] ? x[off[t]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const float *w = wsm + (ci + u) * 9 * CT;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c4 = 0; c4 < CT; c4 += 4) {
                    const floatx4 ww = *reinterpret_cast<const floatx4 *>(w + t * CT + c4);
                    acc[c4] = fmaf(v[u][t], ww[0], acc[c4]);
                    acc[c4 + 1] = fmaf(v[u][t], ww[1], acc[c4 + 1]);
                    acc[c4 + 2] = fmaf(v[u][t], ww[2], acc[c4 + 2]);
                    acc[c4 + 3] = fmaf(v[u][t], ww[3], acc[c4 + 3]);
                }
        }
    }
    float *ob = (a.out2 && co0 >= a.split) ? a.out2 + ((long)b * a.out2_ctotal + a.out2_coff + co0 - a.split) * HoWo + p
                                           : a.out + ((long)b * a.out_ctotal + a.out_coff + co0) * HoWo + p;
    float bq[CT];  // (loads in front of the first store: see dwpw_row4_kernel's epilogue)
#pragma unroll
    for (int c = 0; c < CT; ++c) bq[c] = a.b[co0 + c];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float v = acc[c] + bq[c];
        if (a.relu) v = fmaxf(v, 0.f);
        ob[(long)c * HoWo] = v;
    }
}

// ---------------------------------------------------------------- first conv straight from the u8 frame (identity letterbox only)
// RetinaFace::preprocess (retinaface.cpp:106-136) followed by body.stage1.0 (net.py:104, conv_bn 3->8, stride 2): when the frame
// already has the network's input size the letterbox is the identity, preprocessing is "minus (104,117,123)" and the fp32
// planar input tensor (157 MB per 32 frames, written once and read once) need not exist: every tap is read from the u8 HWC
// frame, converted, mean-subtracted in a register (the same fp32 value the separate kernel would have stored) and fed to
// the same FMA chain.  Zero padding applies to the preprocessed tensor, so taps outside the frame contribute 0, not -mean.
__global__ __launch_bounds__(256) void det_conv1_u8_kernel(const uint8_t *__restrict__ frames, size_t row_stride, size_t frame_stride, Conv3Args a) {
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    const int HoWo = a.Ho * a.Wo;
    if (gp >= (long)a.B * HoWo) return;
    const int b = (int)(gp / HoWo), p = (int)(gp - (long)b * HoWo);
    const int oh = p / a.Wo, ow = p - oh * a.Wo;
    const uint8_t *fb = frames + (size_t)b * frame_stride;
    const float mean[3] = {104.f, 117.f, 123.f};
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    float v[3][9];
    if (ow * 2 + 1 < a.W) {
        // interior columns (all but a right edge that only odd widths have): the 9 bytes of a row's three pixels as two unaligned
        // dword loads + one byte load instead of nine byte loads (the byte loads were the kernel's bottleneck: 91 -> 40 us without them)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            const bool rok = ih >= 0 && ih < a.H;
            const uint8_t *row = fb + (size_t)(rok ? ih : 0) * row_stride + (size_t)ow * 6;
            uint32_t d0, d1;
            if (ow > 0) __builtin_memcpy(&d0, row - 3, 4);
            else { __builtin_memcpy(&d0, row, 4); d0 <<= 24; }  // left edge: only byte 3 (= byte 0 of the row) is used
            __builtin_memcpy(&d1, row + 1, 4);
            const uint32_t d2 = row[5];
            const uint32_t by[9] = {d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u, d0 >> 24, d1 & 255u, (d1 >> 8) & 255u, (d1 >> 16) & 255u, d1 >> 24, d2};
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool ok = rok && (kw > 0 || ow > 0);
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
                const uint8_t *px = fb + (size_t)(ok ? ih : 0) * row_stride + (size_t)(ok ? iw : 0) * 3;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float raw = (float)px[ci];  // unconditional load, masked afterwards
                    v[ci][kh * 3 + kw] = ok ? raw - mean[ci] : 0.f;
                }
            }
    }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {  // same accumulation order as conv3x3_kernel: ci outer, tap inner
        const float *w = a.w + ((long)ci * 9) * a.Cout;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = fmaf(v[ci][t], w[t * a.Cout + c], acc[c]);
    }
    float *ob = a.out + ((long)b * a.out_ctotal + a.out_coff) * HoWo + p;
    float bq[8];  // (loads in front of the first store: see dwpw_row4_kernel's epilogue)
#pragma unroll
    for (int c = 0; c < 8; ++c) bq[c] = a.b[c];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float o = acc[c] + bq[c];
        if (a.relu) o = fmaxf(o, 0.f);
        ob[(long)c * HoWo] = o;
    }
}

// ---------------------------------------------------------------- heads: 1x1 64->8 (bbox) and 64->4 (class) + softmax, NHWC order
struct HeadMulti {
    HeadArgs p[3];
};
template <int UNR>
__global__ __launch_bounds__(256) void heads_kernel(HeadMulti mm) {
    const HeadArgs &a = mm.p[blockIdx.z];
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    const int HW = a.H * a.W;
    if (gp >= (long)a.B * HW) return;
    const int b = (int)(gp / HW), p = (int)(gp - (long)b * HW);
    float lb[8], lc[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) lb[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) lc[c] = 0.f;
    const float *inb = a.in + (long)b * a.C * HW + p;
    // (sixteen loads in flight per thread: as a rolled loop every input channel was its own memory round trip - 32 us per launch at 32 frames)
#pragma unroll UNR
    for (int ci = 0; ci < a.C; ++ci) {
        const float x = inb[(long)ci * HW];
#pragma unroll
        for (int c = 0; c < 8; ++c) lb[c] = fmaf(x, a.wb[ci * 8 + c], lb[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) lc[c] = fmaf(x, a.wc[ci * 4 + c], lc[c]);
    }
    // anchor index = base + (i*W + j)*2 + l ; channels 4l..4l+3 / 2l..2l+1 belong to anchor l (retinaface_trim.py:22-24,33-35)
    const long an = (long)b * a.A + a.base + (long)p * 2;
    if (a.wl) {  // optional LandmarkHead (retinaface.py:37-46): 1x1 64 -> 20, channels 10l..10l+9 belong to anchor l
        float ll[20];
#pragma unroll
        for (int c = 0; c < 20; ++c) ll[c] = 0.f;
        for (int ci = 0; ci < a.C; ++ci) {
            const float x = inb[(long)ci * HW];
#pragma unroll
            for (int c = 0; c < 20; ++c) ll[c] = fmaf(x, a.wl[ci * 20 + c], ll[c]);
        }
        float blq[20];  // (in front of the first store: see dwpw_row4_kernel's epilogue)
#pragma unroll
        for (int c = 0; c < 20; ++c) blq[c] = a.bl[c];
#pragma unroll
        for (int c = 0; c < 20; ++c) a.ldm[an * 10 + c] = ll[c] + blq[c];
    }
    const float bc0 = a.bc[0], bc1 = a.bc[1], bc2 = a.bc[2], bc3 = a.bc[3];  // (in front of the stores: see dwpw_row4_kernel's epilogue)
    floatx4 o0 = {lb[0] + a.bb[0], lb[1] + a.bb[1], lb[2] + a.bb[2], lb[3] + a.bb[3]};
    floatx4 o1 = {lb[4] + a.bb[4], lb[5] + a.bb[5], lb[6] + a.bb[6], lb[7] + a.bb[7]};
    *reinterpret_cast<floatx4 *>(a.loc + an * 4) = o0;
    *reinterpret_cast<floatx4 *>(a.loc + an * 4 + 4) = o1;
    floatx4 cf;
#pragma unroll
    for (int l = 0; l < 2; ++l) {  // F.softmax(dim=-1) over the 2 classes (retinaface_trim.py:126): max-subtracted exp / sum
        const float c0 = lc[2 * l] + (l ? bc2 : bc0), c1 = lc[2 * l + 1] + (l ? bc3 : bc1);
        const float m = fmaxf(c0, c1);
        const float e0 = expf(c0 - m), e1 = expf(c1 - m);
        const float s = e0 + e1;
        cf[2 * l] = e0 / s;
        cf[2 * l + 1] = e1 / s;
    }
    *reinterpret_cast<floatx4 *>(a.conf + an * 2) = cf;
}

template <int CT, bool HAS_DW, int PPT>
void launch_dwpw_t(const DwPwArgs &a, hipStream_t s) {
    const long total = (long)a.B * a.Ho * a.Wo;
    const long span = (total + PPT - 1) / PPT;
    dim3 grid((unsigned)((span + 255) / 256), a.Cout / CT);
    const size_t lds = (size_t)a.Cin * ((HAS_DW ? 12 : 0) + CT) * sizeof(float);
    hipLaunchKernelGGL((dwpw_kernel<CT, HAS_DW, PPT>), grid, dim3(256), lds, s, a);
}

// 1x1 conv (no depthwise part): as many pixels per thread as keeps >= ~100k threads in flight
void launch_pw(const DwPwArgs &a, hipStream_t s) {
    const long total = (long)a.B * a.Ho * a.Wo;
    if (a.Cout % 32 == 0) {
        const long tiles = a.Cout / 32;
        if (total / 4 * tiles >= 100000) launch_dwpw_t<32, false, 4>(a, s);
        else if (total / 2 * tiles >= 100000) launch_dwpw_t<32, false, 2>(a, s);
        else launch_dwpw_t<32, false, 1>(a, s);
    } else if (a.Cout % 16 == 0) {
        launch_dwpw_t<16, false, 1>(a, s);
    } else {
        launch_dwpw_t<8, false, 1>(a, s);
    }
}

}  // namespace

bool det_mfma_enabled() {
    static const bool use_mfma = [] {
        const char *e = frt_tuning_env("FRT_DET_MFMA");
        return !(e && e[0] == '0');
    }();
    return use_mfma;
}

void launch_dwpw(const DwPwArgs &a, hipStream_t s) {
    if (det_mfma_enabled() && launch_dwpw_mfma(a, s)) return;  // fp32 matrix-core kernels (kernels_det_mfma.hip); scalar kernels = generic fallback
    if (!a.wd) return launch_pw(a, s);
    const long total = (long)a.B * a.Ho * a.Wo;
    const long blocks = (total + 255) / 256;
    // stride-1 blocks whose one channel tile covers every output channel: four pixels of a row per thread
    static const bool row4 = !(frt_tuning_env("FRT_DWPW_ROW4") && frt_tuning_env("FRT_DWPW_ROW4")[0] == '0');
    // the 32 -> 32 block at a few frames per call: thread = pixel with branch-free taps.  Measured per launch (us), dwpw_row4_kernel / this:
    // 29.3 / 12.7 at 1 frame and 29.3 / 17.4 at 2 (two 16-channel tiles over grid.y), 29.8 / 21.3 at 4 (one tile); 31 / 40 at 8 (nine dword
    // loads per pixel and channel: the texture path saturates at ~ 9 lanes per CU and clock)
    static const bool pixs = !(frt_tuning_env("FRT_DWPW_PIXS") && frt_tuning_env("FRT_DWPW_PIXS")[0] == '0');
    if (pixs && !a.add && a.Cout == 32 && a.Cin == 32 && blocks <= 512 && (long)a.B * a.Cin * a.H * a.W * 4 < (1L << 31)) {
        if (blocks <= 256) hipLaunchKernelGGL((dwpw_pixs_kernel<16, 4>), dim3((unsigned)blocks, 2), dim3(256), (size_t)a.Cin * 28 * sizeof(float), s, a);
        else hipLaunchKernelGGL((dwpw_pixs_kernel<32, 2>), dim3((unsigned)blocks, 1), dim3(256), (size_t)a.Cin * 44 * sizeof(float), s, a);
        return;
    }
    if (row4 && a.stride == 1 && !a.add && a.H == a.Ho && a.W == a.Wo && a.W % 4 == 0 && (a.Cout == 16 || a.Cout == 32) && a.Cin <= 64) {
        const long threads = (long)a.B * a.H * (a.W / 4);
        const dim3 grid((unsigned)((threads + 255) / 256));
        static const bool mf = !(frt_tuning_env("FRT_ROW4_MFMA") && frt_tuning_env("FRT_ROW4_MFMA")[0] == '0');
        const bool use_mf = mf && a.Cout == 32 && a.Cin % 2 == 0;
        const size_t lds = (size_t)a.Cin * (12 + a.Cout + (use_mf ? 32 : 0)) * sizeof(float);
        // (ring depth measured on the 8 -> 16 block: 98 / 97 / 99 us with 1 / 2 / 3 channels in flight - the kernel waits on memory 60 % of
        //  its wave cycles but not for lack of bytes in flight; default = the shallow ring, 176 registers)
        static const int ns16 = frt_tuning_env("FRT_ROW4_NS") ? atoi(frt_tuning_env("FRT_ROW4_NS")) : 2;
        if (a.Cout == 16 && ns16 == 4) hipLaunchKernelGGL((dwpw_row4_kernel<16, 4>), grid, dim3(256), lds, s, a);
        else if (a.Cout == 16 && ns16 == 3) hipLaunchKernelGGL((dwpw_row4_kernel<16, 3>), grid, dim3(256), lds, s, a);
        else if (a.Cout == 16) hipLaunchKernelGGL((dwpw_row4_kernel<16, 2>), grid, dim3(256), lds, s, a);
        else if (use_mf) hipLaunchKernelGGL((dwpw_row4_kernel<32, 2, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((dwpw_row4_kernel<32, 2>), grid, dim3(256), lds, s, a);
        return;
    }
    // fused while one channel tile covers every output channel (no depthwise recompute) ...
    if (a.Cout <= 32 || !a.tmp) {
        if (a.Cout % 32 == 0) {
            if (blocks >= 2048) launch_dwpw_t<32, true, 2>(a, s);
            else launch_dwpw_t<32, true, 1>(a, s);
        } else if (a.Cout % 16 == 0) {
            if (blocks >= 2048) launch_dwpw_t<16, true, 2>(a, s);
            else launch_dwpw_t<16, true, 1>(a, s);
        } else {
            launch_dwpw_t<8, true, 1>(a, s);
        }
        return;
    }
    // ... otherwise depthwise -> tmp [B][Cin][Ho][Wo], then a register-blocked pointwise conv from tmp
    const long n = (long)a.B * a.Cin * a.Ho * a.Wo;
    hipLaunchKernelGGL(dw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    DwPwArgs b = a;
    b.in = a.tmp;
    b.wd = nullptr;
    b.bd = nullptr;
    b.H = a.Ho;
    b.W = a.Wo;
    b.stride = 1;
    launch_pw(b, s);
}

void launch_conv3x3_multi(const Conv3Args *a, int n, hipStream_t s) {
    if (det_mfma_enabled() && launch_conv3x3_split(a, n, s)) return;  // fp16 hi/lo split on the fp16 matrix cores (kernels_det_conv3h.hip)
    if (det_mfma_enabled() && launch_conv3x3_mfma(a, n, s)) return;  // fp32 matrix-core kernel (kernels_det_mfma.hip)
    Conv3Multi mm;
    long max_total = 0;
    int cout = a[0].Cout;
    for (int i = 0; i < n; ++i) {
        mm.p[i] = a[i];
        max_total = max_total > (long)a[i].B * a[i].Ho * a[i].Wo ? max_total : (long)a[i].B * a[i].Ho * a[i].Wo;
    }
    for (int i = n; i < 3; ++i) mm.p[i] = a[0];
    const unsigned gx = (unsigned)((max_total + 255) / 256);
    const bool two_out = a[0].out2 != nullptr;  // the split must fall on a channel-tile boundary: 16-channel tiles
    // a few frames: the LDS-weight variant (bit-identical).  Measured per launch, conv3x3_kernel<16> / this: 22.4 / 10.7 us at 1 frame (8-channel
    // tiles, all 16 input channels' taps in flight), 23.4 / 16.0 at 4 frames (16-channel tiles), equal at 8, 43 / 60 at 32
    static const int ldsw = frt_tuning_env("FRT_C3_LDSW") ? atoi(frt_tuning_env("FRT_C3_LDSW")) : 1;  // 0: off, 2: at every batch size
    bool ldsw_ok = ldsw && cout % 16 == 0 && (ldsw == 2 || gx <= 128);
    for (int i = 0; i < n; ++i) ldsw_ok = ldsw_ok && a[i].Cin == 16 && a[i].Cout == cout && !(reinterpret_cast<uintptr_t>(a[i].w) & 15);
    if (ldsw_ok) {
        if (gx <= 32) hipLaunchKernelGGL((conv3x3_ldsw_kernel<8, 16>), dim3(gx, cout / 8, n), dim3(256), (size_t)16 * 9 * 8 * sizeof(float), s, mm);
        else hipLaunchKernelGGL((conv3x3_ldsw_kernel<16, 4>), dim3(gx, cout / 16, n), dim3(256), (size_t)16 * 9 * 16 * sizeof(float), s, mm);
        return;
    }
    if (cout % 32 == 0 && max_total >= 256L * 256 && !(two_out && a[0].split % 32))
        hipLaunchKernelGGL((conv3x3_kernel<32>), dim3(gx, cout / 32, n), dim3(256), 0, s, mm);
    else if (cout % 16 == 0)
        hipLaunchKernelGGL((conv3x3_kernel<16>), dim3(gx, cout / 16, n), dim3(256), 0, s, mm);
    else
        hipLaunchKernelGGL((conv3x3_kernel<8>), dim3(gx, cout / 8, n), dim3(256), 0, s, mm);
}

void launch_conv3x3(const Conv3Args &a, hipStream_t s) { launch_conv3x3_multi(&a, 1, s); }

bool launch_det_conv1_u8(const uint8_t *frames, size_t row_stride, size_t frame_stride, const Conv3Args &a, hipStream_t s) {
    if (a.Cin != 3 || a.Cout != 8 || a.stride != 2 || frt_tuning_env("FRT_DET_NO_FUSED_INPUT")) return false;
    const long total = (long)a.B * a.Ho * a.Wo;
    hipLaunchKernelGGL(det_conv1_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, row_stride, frame_stride, a);
    return true;
}

void launch_heads_multi(const HeadArgs *a, int n, hipStream_t s) {
    HeadMulti mm;
    long max_total = 0;
    for (int i = 0; i < n; ++i) {
        mm.p[i] = a[i];
        max_total = max_total > (long)a[i].B * a[i].H * a[i].W ? max_total : (long)a[i].B * a[i].H * a[i].W;
    }
    for (int i = n; i < 3; ++i) mm.p[i] = a[0];
    static const int unr = frt_tuning_env("FRT_HEADS_UNROLL") ? atoi(frt_tuning_env("FRT_HEADS_UNROLL")) : 16;  // (tuning build: input channels' loads in flight)
    const dim3 grid((unsigned)((max_total + 255) / 256), 1, n);
    if (unr == 64) hipLaunchKernelGGL(heads_kernel<64>, grid, dim3(256), 0, s, mm);
    else if (unr == 32) hipLaunchKernelGGL(heads_kernel<32>, grid, dim3(256), 0, s, mm);
    else hipLaunchKernelGGL(heads_kernel<16>, grid, dim3(256), 0, s, mm);
}

void launch_heads(const HeadArgs &a, hipStream_t s) { launch_heads_multi(&a, 1, s); }
