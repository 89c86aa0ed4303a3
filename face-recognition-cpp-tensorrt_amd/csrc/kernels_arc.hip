// ArcFace IR-50 / IR-SE-50 forward for gfx950: fp16 NHWC activations, fp32 accumulation on the matrix cores.
//
// Arithmetic spec: /root/reference/conversion/arcface/model_irse.py:48-90 (bottleneck_IR / _IR_SE), :128-173 (Backbone).  In
// the reference this network is a TensorRT fp16 engine (src/arcface.cpp:134,145; conversion/arcface/torch2trt.py:42-43).
//
// 99.5 % of the 12.6 GFLOP/face are 3x3 convolutions with Cin, Cout in {64,128,256,512}: genuine dense contractions
//   D[cout][pixel] = sum_{tap,ci} W[cout][tap][ci] * X[pixel shifted by tap][ci]
// run as an implicit GEMM on v_mfma_f32_32x32x16_f16 (M = Cout, N = B*Ho*Wo pixels, K = 9*Cin), never materialising im2col.
//   * workgroup = 4 waves; tile = (WCO*64 couts) x (WPX*64 pixels); each wave owns a 64x64 sub-tile = 2x2 MFMA tiles;
//   * K is walked tap-major in steps of 64 input channels (Cin % 64 == 0, so a step never straddles a tap);
//   * both operand tiles are staged through LDS with coalesced 16-byte global loads (8 lanes = one 128-byte row segment),
//     double-buffered with a register prefetch of step t+1 issued before the MFMAs of step t; zero padding is applied at
//     load time (out-of-image taps load zeros), which is what makes the leading BatchNorm un-foldable (see below);
//   * LDS rows are 128 B; 16-byte chunks are XOR-swizzled with (row>>1)&7, conflict-free for ds_read_b128 lane groups;
//   * the accumulator orientation is D[cout][pixel] (weights are the MFMA A operand): a lane then owns 4 consecutive output
//     channels of one pixel per accumulator group -> 8-byte NHWC stores and float4 per-channel parameter loads.
// Fused epilogues (all in fp32 on the accumulator, one rounding to fp16 at the store):
//   EPI_PRELU      out0 = prelu(acc, slope[c])                                   (res_layer conv1 + PReLU)
//   EPI_BN         out0 = acc*s[c] + b[c]                                        (1x1 stride-2 shortcut conv + BN; SE path)
//   EPI_BN_ADD_BN  y = acc*s[c] + b[c] + shortcut ; out0 = y ; out1 = y*s'[c]+b'[c]   (conv2 + BN + residual add, plus the
//                  NEXT unit's leading BatchNorm2d: that BN sits before a zero-padded conv (model_irse.py:57-58), so it cannot
//                  be folded into the conv weights exactly - it is applied here, where the value is produced)
//   EPI_PARTIAL    outf[split][pixel][cout] = acc                                (split-K partial sums for the final Linear)
// MaxPool2d(1, stride) shortcuts are pure indexing: the shortcut operand is sampled at (oh*stride, ow*stride).
#include <cstdio>
#include <cstring>

#include "frt_kernels.h"
#include "frt_se_device.h"

#include <stdlib.h>

#include <type_traits>

namespace {

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int WCO, int WPX>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvMfmaArgs p) {
    constexpr int BCO = WCO * 64, BPX = WPX * 64;
    constexpr int CO_CH = BCO * 8 / 256;  // 16-byte chunks per thread, weight tile
    constexpr int PX_CH = BPX * 8 / 256;  // 16-byte chunks per thread, pixel tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *Ws = reinterpret_cast<half_t *>(smem);                  // [2][BCO][64]
    half_t *Xs = reinterpret_cast<half_t *>(smem) + 2 * BCO * 64;   // [2][BPX][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int wco = wave / WPX, wpx = wave % WPX;

    const int M = p.B * p.Ho * p.Wo;
    const int n_co_tiles = p.Cout / BCO;
    const int co_tile = blockIdx.x % n_co_tiles, px_tile = blockIdx.x / n_co_tiles;
    const int co_base = co_tile * BCO, px_base = px_tile * BPX;

    const int cin_steps = p.Cin >> 6;
    const int ksteps = p.ks * p.ks * cin_steps;
    const int per_split = (ksteps + p.splits - 1) / p.splits;
    const int t_begin = blockIdx.z * per_split;
    const int t_end = min(ksteps, t_begin + per_split);
    const long Ktot = (long)p.ks * p.ks * p.Cin;

    const int ld_row = tid >> 3, ld_ch = tid & 7;

    // per-thread im2col row descriptors
    int xb[PX_CH], xih0[PX_CH], xiw0[PX_CH];
#pragma unroll
    for (int i = 0; i < PX_CH; ++i) {
        const int m = px_base + ld_row + 32 * i;
        if (m < M) {
            const int b = m / (p.Ho * p.Wo);
            const int rem = m - b * (p.Ho * p.Wo);
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            xb[i] = b * p.H * p.W;
            xih0[i] = oh * p.stride - p.pad;
            xiw0[i] = ow * p.stride - p.pad;
        } else {
            xb[i] = -1;
            xih0[i] = 0;
            xiw0[i] = 0;
        }
    }

    half8 wreg[CO_CH], xreg[PX_CH];
    auto load_global = [&](int t) {
        const int tap = t / cin_steps;
        const int c0 = (t - tap * cin_steps) << 6;
        const int kh = tap / p.ks, kw = tap - kh * p.ks;
#pragma unroll
        for (int i = 0; i < CO_CH; ++i) {
            const int co = co_base + ld_row + 32 * i;
            wreg[i] = *reinterpret_cast<const half8 *>(p.w + (long)co * Ktot + (long)tap * p.Cin + c0 + ld_ch * 8);
        }
#pragma unroll
        for (int i = 0; i < PX_CH; ++i) {
            const int ih = xih0[i] + kh, iw = xiw0[i] + kw;
            const bool ok = xb[i] >= 0 && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const half8 *>(p.x + ((long)(xb[i] + ih * p.W + iw)) * p.Cin + c0 + ld_ch * 8);
            xreg[i] = v;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CO_CH; ++i) {
            const int row = ld_row + 32 * i;
            *reinterpret_cast<half8 *>(Ws + buf * BCO * 64 + row * 64 + swz(row, ld_ch) * 8) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < PX_CH; ++i) {
            const int row = ld_row + 32 * i;
            *reinterpret_cast<half8 *>(Xs + buf * BPX * 64 + row * 64 + swz(row, ld_ch) * 8) = xreg[i];
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (t_begin < t_end) {
        load_global(t_begin);
        store_lds(0);
    }
    __syncthreads();
    int cur = 0;
    for (int t = t_begin; t < t_end; ++t) {
        if (t + 1 < t_end) load_global(t + 1);
        const half_t *Wb = Ws + cur * BCO * 64;
        const half_t *Xb = Xs + cur * BPX * 64;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + hi;
            half8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wco * 64 + i * 32 + r;
                af[i] = *reinterpret_cast<const half8 *>(Wb + row * 64 + swz(row, ch) * 8);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wpx * 64 + j * 32 + r;
                bf[j] = *reinterpret_cast<const half8 *>(Xb + row * 64 + swz(row, ch) * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < t_end) store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ------------------------------------------------------------------ epilogue
    // acc[i][j][e]: cout = co_base + wco*64 + i*32 + (e&3) + 8*(e>>2) + 4*hi ; pixel = px_base + wpx*64 + j*32 + r
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = px_base + wpx * 64 + j * 32 + r;
        if (m >= M) continue;
        long sc_off = 0;
        if (p.mode == EPI_BN_ADD_BN) {
            if (p.sc_stride == 1 && p.sc_h == p.Ho && p.sc_w == p.Wo) {
                sc_off = (long)m * p.Cout;
            } else {
                const int b = m / (p.Ho * p.Wo);
                const int rem = m - b * (p.Ho * p.Wo);
                const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
                sc_off = ((long)(b * p.sc_h + oh * p.sc_stride) * p.sc_w + ow * p.sc_stride) * p.Cout;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = co_base + wco * 64 + i * 32 + 8 * g + 4 * hi;
                floatx4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (p.mode == EPI_PARTIAL) {
                    *reinterpret_cast<floatx4 *>(p.outf + ((long)blockIdx.z * M + m) * p.Cout + c) = v;
                    continue;
                }
                const floatx4 a0 = *reinterpret_cast<const floatx4 *>(p.p0 + c);
                if (p.mode == EPI_PRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a0[e];
                } else {
                    const floatx4 a1 = *reinterpret_cast<const floatx4 *>(p.p1 + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * a0[e] + a1[e];
                }
                if (p.mode == EPI_BN_ADD_BN) {
                    const half4 s4 = *reinterpret_cast<const half4 *>(p.sc + sc_off + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)s4[e];
                }
                half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4 *>(p.out0 + (long)m * p.Cout + c) = o;
                if (p.mode == EPI_BN_ADD_BN && p.out1) {
                    const floatx4 a2 = *reinterpret_cast<const floatx4 *>(p.p2 + c);
                    const floatx4 a3 = *reinterpret_cast<const floatx4 *>(p.p3 + c);
                    half4 z = {(half_t)(v[0] * a2[0] + a3[0]), (half_t)(v[1] * a2[1] + a3[1]), (half_t)(v[2] * a2[2] + a3[2]),
                               (half_t)(v[3] * a2[3] + a3[3])};
                    *reinterpret_cast<half4 *>(p.out1 + (long)m * p.Cout + c) = z;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- v2: direct global->LDS staging (LDS-DMA), N-stage ring
// Same tiling/arithmetic as conv_mfma_kernel above, restructured around what the v1 profile showed (30 % MFMA duty):
//   * both operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write_b128 issue cost).
//     The DMA writes wave-base + lane*16, so the LDS image is lane-linear; the XOR swizzle is applied to the per-lane SOURCE
//     address (lane L of a row fetches chunk (L&7)^f(row)) and again on the ds_read side - the same involution on both sides.
//     Out-of-image taps / rows beyond M fetch from a small zero buffer (the DMA cannot zero-fill).
//   * NSTAGE-deep ring with counted s_waitcnt vmcnt(N) and a raw s_barrier: tile t+NSTAGE-1 is issued right after the barrier
//     that retires tile t-1's readers; loads stay in flight across barriers (never drained in the steady state).
//   * epilogue staged through LDS: the D[cout][pixel] accumulators are transposed to pixel rows, then every lane handles 8
//     consecutive channels of a pixel -> 16-byte coalesced shortcut loads and output stores, parameters hoisted per lane.
//   * XCD-aware tile order: consecutive logical tiles (which share the input pixels / the weights) land on the same XCD's L2.
template <int WCO, int WPX, int NSTAGE, int ABL = 0>  // ABL: timing ablations only (1 = no DMA in the loop, 2 = no MFMA)
__global__ __launch_bounds__(256) void conv_glds_kernel(ConvMfmaArgs p) {
    constexpr int BCO = WCO * 64, BPX = WPX * 64;
    constexpr int CO_CH = BCO * 8 / 256, PX_CH = BPX * 8 / 256;
    constexpr int LPT = CO_CH + PX_CH;               // DMA instructions per thread per tile
    constexpr int STAGE_HALFS = (BCO + BPX) * 64;    // halfs per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *ring = reinterpret_cast<half_t *>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int wco = wave / WPX, wpx = wave % WPX;

    const int M = p.B * p.Ho * p.Wo;
    const int n_co_tiles = p.Cout / BCO;
    // bijective XCD remap (block b runs on XCD b % 8): give every XCD a contiguous range of logical tiles
    const int nblk = gridDim.x, bq = nblk >> 3, brem = nblk & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
    const int co_tile = lid % n_co_tiles, px_tile = lid / n_co_tiles;
    const int co_base = co_tile * BCO, px_base = px_tile * BPX;

    const int cin_steps = p.Cin >> 6;
    const int ksteps = p.ks * p.ks * cin_steps;
    const int per_split = (ksteps + p.splits - 1) / p.splits;
    const int t_begin = blockIdx.z * per_split;
    const int t_end = min(ksteps, t_begin + per_split);
    const long Ktot = (long)p.ks * p.ks * p.Cin;

    const int lrow = lane >> 3, lpos = lane & 7;  // row within the wave's 8-row slab, 16-byte slot within the row

    // Per-row descriptors, computed once: 32-bit element offset of tap (0,0) / channel 0 (already source-swizzled) and a
    // bit mask of the taps that fall inside the image.  Per K-step the address is then ONE add of a wave-uniform delta.
    int xoff[PX_CH];
    unsigned xmask[PX_CH];
#pragma unroll
    for (int i = 0; i < PX_CH; ++i) {
        const int row = wave * 8 + lrow + 32 * i;
        const int m = px_base + row;
        const int xsw = (lpos ^ ((row >> 1) & 7)) * 8;
        xoff[i] = 0;
        xmask[i] = 0;
        if (m < M) {
            const int b = m / (p.Ho * p.Wo);
            const int rem = m - b * (p.Ho * p.Wo);
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            xoff[i] = ((b * p.H + ih0) * p.W + iw0) * p.Cin + xsw;
            for (int kh = 0; kh < p.ks; ++kh)
                for (int kw = 0; kw < p.ks; ++kw)
                    if (ih0 + kh >= 0 && ih0 + kh < p.H && iw0 + kw >= 0 && iw0 + kw < p.W) xmask[i] |= 1u << (kh * p.ks + kw);
        }
    }
    const half_t *wsrc[CO_CH];
#pragma unroll
    for (int i = 0; i < CO_CH; ++i) {
        const int row = wave * 8 + lrow + 32 * i;
        wsrc[i] = p.w + (long)(co_base + row) * Ktot + (lpos ^ ((row >> 1) & 7)) * 8;
    }
    // LDS byte offsets of this lane's MFMA fragments inside a stage (swizzle folded in), for kk = 0..3
    int aoff[2][4], boff[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ra = wco * 64 + i * 32 + r, rb = wpx * 64 + i * 32 + r;
            aoff[i][kk] = (ra * 64 + swz(ra, kk * 2 + hi) * 8) * 2;
            boff[i][kk] = (BCO * 64 + rb * 64 + swz(rb, kk * 2 + hi) * 8) * 2;
        }

    auto issue = [&](int t, int stage) {
        const int tap = t / cin_steps;
        const int c0 = (t - tap * cin_steps) << 6;
        const int kh = tap / p.ks, kw = tap - kh * p.ks;
        const int doff = (kh * p.W + kw) * p.Cin + c0;  // wave-uniform
        const unsigned tbit = 1u << tap;
        const int woff = tap * p.Cin + c0;
        half_t *wl = ring + stage * STAGE_HALFS + wave * 8 * 64;
        half_t *xl = wl + BCO * 64;
#pragma unroll
        for (int i = 0; i < CO_CH; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[i] + woff),
                                             (__attribute__((address_space(3))) void *)(wl + i * 32 * 64), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PX_CH; ++i) {
            const half_t *src = (xmask[i] & tbit) ? p.x + (unsigned)(xoff[i] + doff) : p.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(xl + i * 32 * 64), 16, 0, 0);
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (t_begin + s < t_end) issue(t_begin + s, s);

    // one K-step on a compile-time stage (LDS offsets become instruction immediates)
    auto kstep = [&](int t, auto stage_c) {
        constexpr int STAGE = decltype(stage_c)::value;
        // tile t must have landed; in the steady state NSTAGE-2 younger tiles stay in flight across the barrier
        if (NSTAGE > 2 && t + NSTAGE - 2 < t_end)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPT) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ABL != 1 && t + NSTAGE - 1 < t_end) issue(t + NSTAGE - 1, (STAGE + NSTAGE - 1) % NSTAGE);  // buffer of tile t-1: all its readers are past the barrier
        const char *sb = smem + STAGE * STAGE_HALFS * 2;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            half8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const half8 *>(sb + aoff[i][kk]);
                bf[i] = *reinterpret_cast<const half8 *>(sb + boff[i][kk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (ABL == 2) {
                        asm volatile("" ::"v"(af[i]), "v"(bf[j]));
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
    for (int t = t_begin; t < t_end; t += NSTAGE) {
        kstep(t, std::integral_constant<int, 0>{});
        if (t + 1 < t_end) kstep(t + 1, std::integral_constant<int, 1>{});
        if (NSTAGE > 2 && t + 2 < t_end) kstep(t + 2, std::integral_constant<int, (NSTAGE > 2 ? 2 : 0)>{});
    }
    __syncthreads();  // every wave is done with the ring: reuse it as the epilogue transpose buffer

    // ------------------------------------------------------------------ epilogue through LDS
    constexpr int EROW = 68;  // floats per pixel row (64 + 4 pad: conflict-free 8-lane ds_write_b128 groups)
    float *ep = reinterpret_cast<float *>(smem) + wave * (32 * EROW);
    const int chunk = lane & 7;
    const int c = co_base + wco * 64 + chunk * 8;
    floatx4 q0[2], q1[2], q2[2], q3[2];
    if (p.mode != EPI_PARTIAL) {
        q0[0] = *reinterpret_cast<const floatx4 *>(p.p0 + c);
        q0[1] = *reinterpret_cast<const floatx4 *>(p.p0 + c + 4);
        if (p.mode != EPI_PRELU) {
            q1[0] = *reinterpret_cast<const floatx4 *>(p.p1 + c);
            q1[1] = *reinterpret_cast<const floatx4 *>(p.p1 + c + 4);
        }
        if (p.mode == EPI_BN_ADD_BN && p.out1) {
            q2[0] = *reinterpret_cast<const floatx4 *>(p.p2 + c);
            q2[1] = *reinterpret_cast<const floatx4 *>(p.p2 + c + 4);
            q3[0] = *reinterpret_cast<const floatx4 *>(p.p3 + c);
            q3[1] = *reinterpret_cast<const floatx4 *>(p.p3 + c + 4);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<floatx4 *>(ep + r * EROW + i * 32 + 8 * g + 4 * hi) = v;
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pr = (lane >> 3) + 8 * it;
            const int m = px_base + wpx * 64 + j * 32 + pr;
            const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + pr * EROW + chunk * 8);
            const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + pr * EROW + chunk * 8 + 4);
            if (m >= M) continue;
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.mode == EPI_PARTIAL) {
                float *o = p.outf + ((long)blockIdx.z * M + m) * p.Cout + c;
                *reinterpret_cast<floatx4 *>(o) = v0;
                *reinterpret_cast<floatx4 *>(o + 4) = v1;
                continue;
            }
            if (p.mode == EPI_PRELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * q0[e >> 2][e & 3];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * q0[e >> 2][e & 3] + q1[e >> 2][e & 3];
            }
            if (p.mode == EPI_BN_ADD_BN) {
                long sc_off;
                if (p.sc_stride == 1 && p.sc_h == p.Ho && p.sc_w == p.Wo) {
                    sc_off = (long)m * p.Cout;
                } else {
                    const int b = m / (p.Ho * p.Wo);
                    const int rem = m - b * (p.Ho * p.Wo);
                    const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
                    sc_off = ((long)(b * p.sc_h + oh * p.sc_stride) * p.sc_w + ow * p.sc_stride) * p.Cout;
                }
                const half8 s8 = *reinterpret_cast<const half8 *>(p.sc + sc_off + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)s8[e];
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
            *reinterpret_cast<half8 *>(p.out0 + (long)m * p.Cout + c) = o;
            if (p.mode == EPI_BN_ADD_BN && p.out1 && ABL != 20) {  // 20 (measurement): what would dropping the BN'd copy save?
                half8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[e] * q2[e >> 2][e & 3] + q3[e >> 2][e & 3]);
                *reinterpret_cast<half8 *>(p.out1 + (long)m * p.Cout + c) = z;
            }
        }
    }
}

__device__ __forceinline__ void se_fc_gate(const float *sp, float *shid, const float *__restrict__ w1, const float *__restrict__ w2, int C, int f,
                                           float *__restrict__ gate) {
    se_fc1(sp, shid, w1, C);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) gate[(long)f * C + c] = se_fc2(shid, w2, C, c);
}

// ---------------------------------------------------------------- v3: LDS-resident halo patch ("strip") kernel, 3x3 / stride 1 / pad 1
// The im2col kernels above move every input pixel through the L2->LDS path 9 times (once per tap); the v2 ablation showed that
// path, not the matrix pipe, bounds them (DMA-only 45 us vs MFMA-only 37 us on the 14x14 layers).  Here a workgroup owns a
// STRIP of output pixels - R image rows x W columns of one image (n_img whole images for 7x7) = up to 224 pixel slots = 7 MFMA
// pixel tiles - and keeps the strip's halo'd input patch ((R+2) x (W+2) pixels x one 64-channel chunk, zero border) resident in
// LDS for all 9 taps: a tap is just a row offset into the patch.  Only the [128 cout x 64] weight tile streams per (chunk, tap)
// step through a 3-stage LDS-DMA ring; the next chunk's patch is prefetched in pieces during taps 0-5 of the current chunk.
//   * DMA per MFMA drops ~3x (16 KB weights + 1/9 patch per 112 MFMAs  vs  32 KB per 64 MFMAs);
//   * 4 waves x (1 cout fragment x 7 pixel fragments): 28 MFMAs per wave per barrier (was 16), A fragment reused 7x;
//   * the patch uses 144-byte pixel rows (9 x 16 B, last slot is padding the DMA fills from the zero buffer): conflict-free
//     ds_read_b128 without an XOR swizzle, so a tap costs ONE address add per pixel tile; kk offsets are immediates;
//   * every step issues the same number of DMA instructions (dummy ones from the zero buffer at the tail), and the 9 taps are
//     unrolled with compile-time (tap, ring stage = tap % 3): every s_waitcnt vmcnt(N) is an exact compile-time count;
//   * grids fit the machine: 14x14x256 -> 128 images x 2 cout tiles = 256 workgroups on 256 CUs.
// PAIR (Cout == 64, Cin == 64): one workgroup handles TWO strips; waves (0,1) own the first, waves (2,3) the second, each wave
// one 32-cout fragment of its strip.  Each half stages its own patch (128 threads per patch image); the whole K loop (9 taps)
// runs on that single resident patch.
// CPT ("compact", round 6): the strip is NT*32 CONSECUTIVE pixels of the flattened (image, row, column) index - no dead pixel slots
// (a 14x14 image is 6.125 tiles: 8 images = 49 tiles = 7 strips; the padded row enumeration spends 28 of every 224 slots on the two halo
// columns).  A strip then spans image boundaries, so the patch is a window of the STACKED images - one zero separator row between two
// images (bottom halo of one, top halo of the next), NO halo columns (they would break the one-to-one slot -> patch-row walk): the taps
// with kw = 0 / kw = 2 point the lanes whose pixel sits in the first / last image column at a zero pixel (patch pixel 0) instead - one
// v_cndmask on the ADDRESS per (tap, tile), masks wave-uniform in scalar registers.  R carries the patch's row count, n_img / linear unused.
template <int PPS, int PT, int NW, bool SINGLE, int ABL, bool PAIR, int NT, int BFD, int WR, bool SEP, bool CPT>  // SEP: SE pooling + gate in the epilogue (below); WR: weight register ring depth in steps (3 or 9); NT pixel tiles per strip; PPS patch DMA pieces per thread per step during taps 0..PT-1; BFD: depth (kk-slots) of the B fragment ring
__device__ __forceinline__ void conv_patch_body(const ConvMfmaArgs &p, int R, int n_img, int linear) {
    // linear != 0: pixel slots are enumerated over the PADDED row width (slot == patch row of tap (0,0), slots in the two halo
    // columns are dead).  The 32 lanes of a fragment read then touch 32 consecutive patch rows -> no LDS bank conflicts; the
    // image-row wrap of the compact enumeration (a 2-row skip) costs ~40 % extra LDS cycles (measured SQ_LDS_BANK_CONFLICT).
    static_assert(!PAIR || SINGLE, "pair mode needs the single-chunk path");
    static_assert(!CPT || (!PAIR && !SINGLE && !SEP && BFD == 1), "compact strips: the main two-buffer variant only (the SE tail pools per image)");
    constexpr int NSLOT = PAIR ? 34 : PT * PPS;
    constexpr int PATCH_B = NSLOT * (PAIR ? 128 : 256) * 16;  // bytes per patch buffer (whole DMA slots)
    constexpr int PROW = 144;                    // bytes per patch pixel row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char *patch = smem + (PAIR ? (wave >> 1) * PATCH_B : 0);  // LDS holds ONLY patches; weights go L2 -> registers
    const int r = lane & 31, hi = lane >> 5;
    const int H = p.H, W = p.W, Wp = CPT ? W : W + 2;
    const int NP = CPT ? R * W + 1 : n_img * (R + 2) * Wp;
    const int strips_per_img = (H + R - 1) / R;  // (the last strip of an image may be ragged - linear enumeration only, see patch_geometry)
    const int n_valid = n_img * R * W;

    const int n_co_tiles = PAIR ? 1 : p.Cout >> 7;
    const int nblk = gridDim.x, bq = nblk >> 3, brem = nblk & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
    const int co_tile = lid % n_co_tiles;
    const int strip = PAIR ? (lid / n_co_tiles) * 2 + (wave >> 1) : lid / n_co_tiles;
    const int co_base = PAIR ? 0 : co_tile * 128;
    const int cow = PAIR ? (wave & 1) * 32 : wave * 32;  // this wave's cout rows inside the tile
    const int c_mlo = strip * (NT * 32);  // compact: first pixel of the strip
    const bool strip_ok = CPT ? c_mlo < p.B * H * W : strip < ((p.B + n_img - 1) / n_img) * strips_per_img;
    const int img0 = CPT ? 0 : (strip / strips_per_img) * n_img;
    const int row0 = CPT ? 0 : (strip % strips_per_img) * R;
    // compact: image / in-image offset of the first pixel, and the STACKED row (image b occupies rows b*(H+1) .. b*(H+1)+H-1, row b*(H+1)+H is
    // the zero separator) of patch row 0 = one above the first pixel's (wave-uniform divisions, once)
    const int c_img_lo = CPT ? c_mlo / (H * W) : 0;
    const int c_rem_lo = CPT ? c_mlo - c_img_lo * (H * W) : 0;
    const int c_top = CPT ? c_img_lo * (H + 1) + c_rem_lo / W - 1 : 0;

    const int n_chunks = p.Cin >> 6;

    // ---- patch DMA descriptors: slot q of this lane covers 16-byte chunk g = (q*4 + wave)*64 + lane of the patch image
    //      (round 5: the two divisions by run-time values per descriptor are reciprocal multiplies with one correction step - integer division
    //      is ~ 40 instructions on this ISA, and conv_s2_kernel's phase stamps had shown 12 us of such arithmetic in front of its first DMA)
    int poff[NSLOT];
    {
        const int patch_px = (R + 2) * Wp;
        const float inv_patch = 1.0f / (float)patch_px, inv_wp = 1.0f / (float)Wp;
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            const int g = PAIR ? (q * 2 + (wave & 1)) * 64 + lane : (q * 4 + wave) * 64 + lane;
            const int prow = g / 9, pos = g - prow * 9;
            if constexpr (CPT) {
                const int pp = prow - 1;  // patch pixel 0 is the zero pixel
                const float inv_w = 1.0f / (float)W, inv_h1 = 1.0f / (float)(H + 1);
                int pr = (int)(((float)pp + 0.5f) * inv_w);
                int pc = pp - pr * W;
                if (pc < 0) { --pr; pc += W; } else if (pc >= W) { ++pr; pc -= W; }
                const int sr = c_top + pr;
                int b = (int)(((float)sr + 0.5f) * inv_h1);
                int iy = sr - b * (H + 1);
                if (iy < 0) { --b; iy += H + 1; } else if (iy > H) { ++b; iy -= H + 1; }
                const bool live = pos < 8 && pp >= 0 && pr < R && strip_ok && sr >= 0 && b < p.B && iy < H;
                poff[q] = live ? ((b * H + iy) * W + pc) * p.Cin + pos * 8 : -1;
                continue;
            }
            int il = (int)(((float)prow + 0.5f) * inv_patch);
            int rem = prow - il * patch_px;
            if (rem < 0) { --il; rem += patch_px; } else if (rem >= patch_px) { ++il; rem -= patch_px; }
            int pr = (int)(((float)rem + 0.5f) * inv_wp);
            int pc = rem - pr * Wp;
            if (pc < 0) { --pr; pc += Wp; } else if (pc >= Wp) { ++pr; pc -= Wp; }
            const int iy = row0 + pr - 1, ix = pc - 1, b = img0 + il;
            const bool live = pos < 8 && prow < NP && strip_ok && b < p.B && iy >= 0 && iy < H && ix >= 0 && ix < W;
            poff[q] = live ? ((b * H + iy) * W + ix) * p.Cin + pos * 8 : -1;
        }
    }
    // ---- weights: each wave consumes only its own 32 cout rows, so the A fragments never touch LDS: lane (r, hi) loads its
    //      four 16-byte fragments (kk = 0..3) of W[co_base + wave*32 + r][tap][chunk*64 + (kk*2+hi)*8 ..] straight from L2 into
    //      registers, two steps ahead (register ring of 3 steps, index = tap % 3 at compile time).  This removes the weight
    //      tile from the LDS write AND read paths - LDS bandwidth (fragment reads + LDS-DMA writes) was the binding resource.
    //      The fragments come from a second, FRAGMENT-ORDERED copy of the weights (p.wf, packed by the host at load time):
    //      [32-cout block][64-channel chunk][tap][kk][lane][8 halfs], so each of these loads is one contiguous kilobyte per wave and a
    //      wave walks its 32 couts' weights as one sequential stream (row-major rows made every load touch 32 different 128-byte lines,
    //      32 bytes of each: four times the address/tag work for the same bytes).
    const half_t *wfrag = p.wf + ((long)((co_base + cow) >> 5) * n_chunks) * (9 * 4 * 512) + lane * 8;
    // ---- B-fragment base addresses: pixel slot -> patch row of tap (0,0)
    int pbase[NT];
    unsigned long long mL[CPT ? NT : 1], mR[CPT ? NT : 1];  // compact: lanes whose pixel sits in the first / last image column (wave-uniform masks)
    const int zoff = hi * 16;                               // ... and where those lanes read instead for kw = 0 / kw = 2: the zero pixel
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int sl = j * 32 + r;
        int pidx = 0;
        if constexpr (CPT) {
            const int u = c_rem_lo + sl;  // < H*W + NT*32
            const int per = H * W;
            int il = (int)(((float)u + 0.5f) * (1.0f / (float)per));
            int rem = u - il * per;
            if (rem < 0) { --il; rem += per; } else if (rem >= per) { ++il; rem -= per; }
            int rr = (int)(((float)rem + 0.5f) * (1.0f / (float)W));
            int cc = rem - rr * W;
            if (cc < 0) { --rr; cc += W; } else if (cc >= W) { ++rr; cc -= W; }
            const bool in = c_mlo + sl < p.B * per;
            // own input pixel = patch pixel 1 + (stacked row - c_top) * W + cc; tap (kh, kw) reads that + (kh - 1) * W + (kw - 1)
            pidx = in ? ((c_img_lo + il) * (H + 1) + rr - c_top) * W + cc - W : 0;
            mL[j] = __builtin_amdgcn_ballot_w64(cc == 0);
            mR[j] = __builtin_amdgcn_ballot_w64(cc == W - 1);
        } else if (linear) {
            pidx = sl < R * Wp ? sl : 0;
        } else if (sl < n_valid) {
            const int per = R * W;
            int il = (int)(((float)sl + 0.5f) * (1.0f / (float)per));
            int rem = sl - il * per;
            if (rem < 0) { --il; rem += per; } else if (rem >= per) { ++il; rem -= per; }
            int rr = (int)(((float)rem + 0.5f) * (1.0f / (float)W));
            int cc = rem - rr * W;
            if (cc < 0) { --rr; cc += W; } else if (cc >= W) { ++rr; cc -= W; }
            pidx = (il * (R + 2) + rr) * Wp + cc;
        }
        pbase[j] = pidx * PROW + hi * 16;
    }

    half8 areg[WR][4];
    constexpr int LA = WR - 1;  // weight fragments are fetched LA steps ahead
    auto load_w = [&](int c, int tap, auto slot_c) {  // wave-uniform c, tap; clamped at the tail (values unused there)
        constexpr int S = decltype(slot_c)::value;
        const int woff = (ABL == 13) ? 0 : (c < n_chunks ? (c * 9 + tap) * (4 * 512) : 0);  // 13: every step re-reads the same fragments (cache hits)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) areg[S][kk] = *reinterpret_cast<const half8 *>(wfrag + woff + kk * 512);
    };
    auto issue_patch = [&](int c, auto q0_c, auto nq_c) {
        constexpr int Q0 = decltype(q0_c)::value, NQ = decltype(nq_c)::value;
        const bool real = c < n_chunks;
        char *pl = patch + (SINGLE ? 0 : (c & 1) * PATCH_B) + (PAIR ? (wave & 1) : wave) * 1024;
#pragma unroll
        for (int q = Q0; q < Q0 + NQ; ++q) {
            const half_t *src = (ABL != 14 && real && poff[q] >= 0) ? p.x + (unsigned)(poff[q] + (c << 6)) : p.zeros;  // 14: all pieces from the zero buffer
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(pl + q * (PAIR ? 2048 : 4096)), 16, 0, 0);
        }
    };

    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // prologue: patch(0) completely, then the weight fragments of steps 0 and 1
    issue_patch(0, std::integral_constant<int, 0>{}, std::integral_constant<int, NSLOT>{});
    load_w(0, 0, std::integral_constant<int, 0>{});
    load_w(0, 1, std::integral_constant<int, 1>{});
    if constexpr (WR == 9) {
        load_w(0, 2, std::integral_constant<int, 2>{});
        load_w(0, 3, std::integral_constant<int, 3>{});
        load_w(0, 4, std::integral_constant<int, 4>{});
        load_w(0, 5, std::integral_constant<int, 5>{});
        load_w(0, 6, std::integral_constant<int, 6>{});
        load_w(0, 7, std::integral_constant<int, 7>{});
    }

    // (An L2 warm-up - every workgroup of an XCD pulling a different slice of the cout tile's weights through L2 at kernel start,
    //  LDS-DMA into the still unused second patch buffer - measured 45.1 us against 44.4 us without: not kept.)
    // B fragment registers: two kk-deep ring that runs CONTINUOUSLY across steps.  One wave per SIMD means only this wave's own
    // instruction stream can hide LDS latency, so every MFMA is followed by exactly one ds_read that refills the register it
    // just consumed with the fragment two kk-slots ahead - in the second half of a step that is the NEXT tap's fragment (the
    // patch is resident).  sched_barrier(0) pins the order (left alone hipcc emits "2 reads, lgkmcnt(0), 1 MFMA").
    half8 bf[BFD][NT];
    // B fragment of pixel tile j for tap T, kk-slot kko, from the patch buffer at byte offset bufoff (kko is a literal after unrolling: an
    // instruction immediate; compact strips redirect the first / last column's lanes for kw = 0 / 2)
    auto frag = [&](int j, auto tap_c, int bufoff, int kko) -> half8 {
        constexpr int T = decltype(tap_c)::value;
        const int dpo = ((T / 3) * Wp + (T % 3)) * PROW + bufoff;  // wave-uniform
        int a = pbase[j] + dpo;
        if constexpr (CPT && T % 3 == 0) a = __builtin_amdgcn_inverse_ballot_w64(mL[j]) ? zoff + bufoff : a;
        if constexpr (CPT && T % 3 == 2) a = __builtin_amdgcn_inverse_ballot_w64(mR[j]) ? zoff + bufoff : a;
        return *reinterpret_cast<const half8 *>(patch + a + kko * 32);
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LA) : "memory");  // this wave's patch(0) pieces have landed (younger: the 4*LA fragment loads)
    __builtin_amdgcn_s_barrier();                     // ... and everybody else's
#pragma unroll
    for (int k2 = 0; k2 < BFD; ++k2)
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[k2][j] = frag(j, std::integral_constant<int, 0>{}, 0, k2);

    auto step = [&](int c, auto tap_c) {
        constexpr int TAP = decltype(tap_c)::value;
        constexpr int NTAP = (TAP + 1) % 9;
        constexpr int AS = TAP % WR;  // register-ring slot of this step's weight fragments (9 taps = 3 x 3: compile time)
        const int pbuf = SINGLE ? 0 : (c & 1) * PATCH_B;
        const int pbufn = TAP == 8 ? (SINGLE ? 0 : ((c + 1) & 1) * PATCH_B) : pbuf;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = BFD == 2 ? (kk & 1) : 0;
            if (kk == 4 - BFD && !SINGLE && TAP == 8) {
                // chunk boundary: from here on the refills read the NEXT chunk's patch buffer.  This wave's pieces (last issued
                // at tap PT-1; younger: the 4 fragment loads of each later step) have landed, every wave's reads of the buffer
                // about to be recycled have returned, then everybody meets.
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (8 - (PT - 1))) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (ABL == 2) asm volatile("" ::"v"(areg[AS][kk]), "v"(bf[cur][j]));
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[AS][kk], bf[cur][j], acc[j], 0, 0, 0);
                if (ABL == 8 || ABL == 9) {
                } else if (kk + BFD < 4) bf[cur][j] = frag(j, std::integral_constant<int, TAP>{}, pbuf, kk + BFD);
                else bf[cur][j] = frag(j, std::integral_constant<int, NTAP>{}, pbufn, kk + BFD - 4);
                constexpr int JL = NT > 1 ? 1 : 0;  // pixel-tile slot that carries the loads of future steps
                if (ABL != 1 && ABL != 7 && ABL != 9 && kk == 0 && j == JL) {  // weight fragments of step t+2 into the slot step t-1 used
                    constexpr int T2 = TAP + LA;
                    load_w(T2 < 9 ? c : c + 1, T2 % 9, std::integral_constant<int, (T2 % 9) % WR>{});
                }
                if (ABL != 1 && ABL != 6 && ABL != 9 && kk == 1 && j == JL && !SINGLE && TAP < PT)
                    issue_patch(c + 1, std::integral_constant<int, (TAP < PT ? TAP : 0) * PPS>{}, std::integral_constant<int, PPS>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int c = 0; c < (ABL == 4 ? 0 : n_chunks); ++c) {
        step(c, std::integral_constant<int, 0>{});
        step(c, std::integral_constant<int, 1>{});
        step(c, std::integral_constant<int, 2>{});
        step(c, std::integral_constant<int, 3>{});
        step(c, std::integral_constant<int, 4>{});
        step(c, std::integral_constant<int, 5>{});
        step(c, std::integral_constant<int, 6>{});
        step(c, std::integral_constant<int, 7>{});
        step(c, std::integral_constant<int, 8>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's dummy DMAs still target LDS
    __syncthreads();
    if (ABL == 5) {  // timing ablation: keep the accumulators alive, skip the epilogue
        float sacc = 0.f;
        for (int j = 0; j < NT; ++j) sacc += acc[j][0];
        if (sacc == 123.456f) p.out0[0] = (half_t)sacc;
        return;
    }

    // ------------------------------------------------------------------ epilogue (per wave: 32 couts x 7 pixel tiles) through LDS
    constexpr int EROW = 36;  // floats per pixel row (32 + 4 pad)
    float *ep = reinterpret_cast<float *>(smem) + wave * (32 * EROW);
    const int chunk = lane & 3;
    const int cch = co_base + cow + chunk * 8;
    floatx4 q0[2], q1[2], q2[2], q3[2];
    q0[0] = *reinterpret_cast<const floatx4 *>(p.p0 + cch);
    q0[1] = *reinterpret_cast<const floatx4 *>(p.p0 + cch + 4);
    if (p.mode != EPI_PRELU) {
        q1[0] = *reinterpret_cast<const floatx4 *>(p.p1 + cch);
        q1[1] = *reinterpret_cast<const floatx4 *>(p.p1 + cch + 4);
    }
    if (p.mode == EPI_BN_ADD_BN && p.out1 && !SEP) {
        q2[0] = *reinterpret_cast<const floatx4 *>(p.p2 + cch);
        q2[1] = *reinterpret_cast<const floatx4 *>(p.p2 + cch + 4);
        q3[0] = *reinterpret_cast<const floatx4 *>(p.p3 + cch);
        q3[1] = *reinterpret_cast<const floatx4 *>(p.p3 + cch + 4);
    }
    // pixel slots of a strip are CONTIGUOUS in the flattened (image, row, column) index: m = m0 + slot (no divisions)
    const long m0 = CPT ? (long)c_mlo : ((long)img0 * H + row0) * W;
    const long Mtot = (long)p.B * H * W;
    const float inv_wp = 1.0f / (float)Wp;
    auto slot_pixel = [&](int sl, long &m) -> bool {  // pixel slot -> flattened output pixel index; false for dead slots
        if constexpr (CPT) {
            m = m0 + sl;
            return m < Mtot;
        }
        if (linear) {
            const int rr = (int)(((float)sl + 0.5f) * inv_wp);  // exact for sl < 2^20
            const int cc = sl - rr * Wp;
            m = m0 + rr * W + cc;
            return strip_ok && rr < R && row0 + rr < H && cc < W && m < Mtot;
        }
        m = m0 + sl;
        return strip_ok && sl < n_valid && m < Mtot;
    };
    if constexpr (SEP) {  // IR-SE: the whole SE tail here (frt_se_device.h)
        const int per_img = R * W;  // (compact slot enumeration when the strip holds more than one image)
        se_tail_epilogue<NT, (NT == 4 ? 2 : 1)>(p, acc, ep, reinterpret_cast<float *>(smem + 4 * 32 * EROW * 4), strip % strips_per_img, strips_per_img,
                                               n_co_tiles, img0, n_img, H * W, co_base, cow, [&](int sl, long &m, int &il) -> bool {
                                                   il = (NT == 4 && !linear && sl >= per_img) ? 1 : 0;
                                                   return slot_pixel(sl, m);
                                               });
        return;
    }
    half8 sc8[NT][2];
    if (p.mode == EPI_BN_ADD_BN) {  // stride 1: the shortcut has the output's geometry; all 14 loads in flight before the transposes
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int sl = j * 32 + (lane >> 2) + 16 * it;
                long m;
                const bool ok = slot_pixel(sl, m);
                sc8[j][it] = *reinterpret_cast<const half8 *>(p.sc + (ok ? m : 0) * p.Cout + cch);
            }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            *reinterpret_cast<floatx4 *>(ep + r * EROW + 8 * g + 4 * hi) = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int px = (lane >> 2) + 16 * it;
            const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8);
            const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8 + 4);
            const int sl = j * 32 + px;
            long m;
            if (!slot_pixel(sl, m)) continue;
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.mode == EPI_PRELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * q0[e >> 2][e & 3];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * q0[e >> 2][e & 3] + q1[e >> 2][e & 3];
            }
            if (p.mode == EPI_BN_ADD_BN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)sc8[j][it][e];
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
            *reinterpret_cast<half8 *>(p.out0 + m * p.Cout + cch) = o;
            if (p.mode == EPI_BN_ADD_BN && p.out1) {
                half8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[e] * q2[e >> 2][e & 3] + q3[e >> 2][e & 3]);
                *reinterpret_cast<half8 *>(p.out1 + m * p.Cout + cch) = z;
            }
        }
    }
}

template <int PPS, int PT, int NW, bool SINGLE, int ABL = 0, bool PAIR = false, int NT = 7, int BFD = 2, int WR = 3, bool SEP = false>
__global__ __launch_bounds__(256, BFD == 1 ? 2 : 1) void conv_patch_kernel(ConvMfmaArgs p, int R, int n_img, int linear) {
    conv_patch_body<PPS, PT, NW, SINGLE, ABL, PAIR, NT, BFD, WR, SEP, false>(p, R, n_img, linear);
}

// compact strips (see CPT above): NT pixel tiles = NT*32 consecutive pixels per strip; npr = rows of the stacked-image patch window
template <int NT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_patchc_kernel(ConvMfmaArgs p, int npr) {
    conv_patch_body<10, 1, 5, false, ABL, false, NT, 1, 3, false, true>(p, npr, 1, 0);
}

// ---------------------------------------------------------------- input layer: conv3x3 3->64 + BN + PReLU (+ unit-0 leading BN)
// 0.3 % of the FLOPs, K = 27: plain VALU.  8 lanes share one pixel, each lane owns 8 of the 64 output channels, so a pixel's
// 128-byte NHWC row is written by 8 consecutive lanes (fully coalesced 16-byte stores); the 27 taps are the same address for
// those 8 lanes (one broadcast fetch).  Weights [27][64] + the five per-channel vectors live in LDS; a lane reads its 8
// channels with two ds_read_b128 per tap (8 distinct 32-byte segments per wave: conflict-free).
__global__ __launch_bounds__(256) void arc_input_kernel(ArcInputArgs a) {
    __shared__ __attribute__((aligned(16))) float sw[27 * 64 + 5 * 64];
    for (int i = threadIdx.x; i < 27 * 64; i += 256) sw[i] = a.w[i];
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        sw[27 * 64 + c] = a.s0[c];
        sw[27 * 64 + 64 + c] = a.b0[c];
        sw[27 * 64 + 128 + c] = a.slope[c];
        sw[27 * 64 + 192 + c] = a.s1[c];
        sw[27 * 64 + 256 + c] = a.b1[c];
    }
    __syncthreads();
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const long gp = gt >> 3;
    const int cb = (int)(gt & 7) * 8;
    const int HW = a.H * a.W;
    if (gp >= (long)a.F * HW) return;
    const int f = (int)(gp / HW), pix = (int)(gp - (long)f * HW);
    const int oh = pix / a.W, ow = pix - oh * a.W;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci)  // not unrolled: otherwise all 54 weight reads are hoisted and the kernel needs 250 VGPRs
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh - 1 + kh, iw = ow - 1 + kw;
                const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                const float x = ok ? a.x[((long)f * 3 + ci) * HW + ih * a.W + iw] : 0.f;
                const float *wk = sw + (ci * 9 + kh * 3 + kw) * 64 + cb;
                const floatx4 w0 = *reinterpret_cast<const floatx4 *>(wk);
                const floatx4 w1 = *reinterpret_cast<const floatx4 *>(wk + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = fmaf(x, w0[c], acc[c]);
                    acc[4 + c] = fmaf(x, w1[c], acc[4 + c]);
                }
            }
    const float *pv = sw + 27 * 64 + cb;
    half8 y8, z8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float v = acc[c] * pv[c] + pv[64 + c];
        v = v > 0.f ? v : v * pv[128 + c];
        y8[c] = (half_t)v;
        z8[c] = (half_t)(v * pv[192 + c] + pv[256 + c]);
    }
    // y (the raw activation) is only ever read as unit 0's shortcut, MaxPool2d(1, 2) = the even (row, column) positions: write
    // just those, as a dense [F][H/2][W/2][64] tensor (saves 3/4 of a 205 MB write at F = 128)
    if (((oh | ow) & 1) == 0) *reinterpret_cast<half8 *>(a.y + (((long)f * (a.H >> 1) + (oh >> 1)) * (a.W >> 1) + (ow >> 1)) * 64 + cb) = y8;
    *reinterpret_cast<half8 *>(a.z + gp * 64 + cb) = z8;
}

// ---------------------------------------------------------------- Linear split-K reduce + bias + BatchNorm1d + L2 normalise
__global__ __launch_bounds__(512) void fc_finalize_kernel(const float *__restrict__ partial, int splits, int F, const float *__restrict__ bias,
                                                          const float *__restrict__ s, const float *__restrict__ b,
                                                          const int *__restrict__ valid, float *__restrict__ out) {
    const int f = blockIdx.x, o = threadIdx.x;
    float v = 0.f;
    if (splits == 49) {  // the recogniser's 7x7 slices: all 49 loads in flight, then added in slice order (128 blocks cannot hide a dependent load chain)
        float t[49];
#pragma unroll
        for (int k = 0; k < 49; ++k) t[k] = partial[((long)k * F + f) * 512 + o];
#pragma unroll
        for (int k = 0; k < 49; ++k) v += t[k];
    } else {
        for (int k = 0; k < splits; ++k) v += partial[((long)k * F + f) * 512 + o];
    }
    v = (v + bias[o]) * s[o] + b[o];
    float sq = v * v;
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    __shared__ float red[8];
    if ((o & 63) == 0) red[o >> 6] = sq;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot += red[w];
    const float nrm = fmaxf(sqrtf(tot), 1e-12f);  // F.normalize: x / max(||x||_2, eps), model_irse.py:171
    const bool ok = valid == nullptr || valid[f] != 0;
    out[(long)f * 512 + o] = ok ? v / nrm : 0.f;
}

// ---------------------------------------------------------------- SE tail (IR-SE): model_irse.py:22-45
// Pooling + gate in one launch.  grid (SE_SPLIT, F): every block sums its pixel range per channel; the block that arrives LAST for a
// face (device-scope counter; the partial sums travel as device-scope stores / loads) adds the SE_SPLIT partial sums in range order and runs the two tiny FC layers - so the result does
// not depend on which block that is.  (A separate gate kernel cost 5 - 11 us + a dependent launch per unit, 24 units per pass; pool +
// gate in ONE block per face had measured 18.6 us against 6.0 + 5.0: 128 blocks walking 100 KB each are latency-bound.  Folding the
// apply pass in as well - every block waits for its face's gate, then scales its own pixel range - measured 25 us against
// 10 + 13: the wait costs what the dependent launch did.)
__global__ __launch_bounds__(256) void se_pool_gate_kernel(const half_t *__restrict__ res, int HW, int C, int F, float *__restrict__ partial,
                                                           const float *__restrict__ w1, const float *__restrict__ w2, float *__restrict__ gate,
                                                           int *__restrict__ counter) {
    // thread = (channel octet, pixel lane): 16-byte loads, consecutive threads read one pixel's contiguous NHWC row
    __shared__ float red[256 * 8];
    __shared__ int s_last;
    const int f = blockIdx.y, part = blockIdx.x;
    const int C8 = C >> 3, NP = 256 / C8;          // C in {64, 128, 256, 512}: C8 <= 64
    const int oct = threadIdx.x % C8, pl = threadIdx.x / C8;
    const int i0 = (int)((long)HW * part / SE_SPLIT), i1 = (int)((long)HW * (part + 1) / SE_SPLIT);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pl < NP) {
        const half_t *p = res + (long)f * HW * C + oct * 8;
        for (int i = i0 + pl; i < i1; i += NP) {
            const half8 v = *reinterpret_cast<const half8 *>(p + (long)i * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {  // channel c = octet * 8 + e: sum over the NP pixel lanes in lane order
        const int o = c >> 3, e = c & 7;
        float t = 0.f;
        for (int q = 0; q < NP; ++q) t += red[(q * C8 + o) * 8 + e];
        // device-scope store (written through this XCD's L2): the block that gathers the partial sums may sit on another XCD
        __hip_atomic_store(&partial[((long)part * F + f) * C + c], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every store above has been acknowledged before this block announces its arrival.  NOT __threadfence(): a device-scope fence
    // writes back / invalidates the whole L2 (measured: 50 us per launch instead of 6)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(&counter[f], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == SE_SPLIT - 1;
        if (s_last) __hip_atomic_store(&counter[f], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    float *sp = red, *shid = red + 512;
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int q = 0; q < SE_SPLIT; ++q) t += __hip_atomic_load(&partial[((long)q * F + f) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sp[c] = t / (float)HW;
    }
    __syncthreads();
    se_fc_gate(sp, shid, w1, w2, C, f, gate);
}
__global__ __launch_bounds__(256) void se_apply_kernel(SeArgs a) {
    // thread = 8 channels of one pixel
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const int C8 = a.C / 8;
    const long total = (long)a.F * a.H * a.W * C8;
    if (g >= total) return;
    const int c = (int)(g % C8) * 8;
    const long m = g / C8;
    const int f = (int)(m / (a.H * a.W));
    const int rem = (int)(m - (long)f * a.H * a.W);
    const int oh = rem / a.W, ow = rem - oh * a.W;
    const half8 r8 = *reinterpret_cast<const half8 *>(a.res + m * a.C + c);
    const half8 s8 = *reinterpret_cast<const half8 *>(a.sc + (((long)f * a.sc_h + oh * a.sc_stride) * a.sc_w + ow * a.sc_stride) * a.C + c);
    half8 y8, z8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = (float)r8[e] * a.gate[(long)f * a.C + c + e] + (float)s8[e];
        y8[e] = (half_t)v;
        z8[e] = (half_t)(v * a.s1[c + e] + a.b1[c + e]);
    }
    *reinterpret_cast<half8 *>(a.y + m * a.C + c) = y8;
    *reinterpret_cast<half8 *>(a.z + m * a.C + c) = z8;
}

template <int WCO, int WPX>
void launch_conv_t(const ConvMfmaArgs &a, hipStream_t s) {
    constexpr int BCO = WCO * 64, BPX = WPX * 64;
    const size_t lds = (size_t)2 * (BCO + BPX) * 64 * sizeof(half_t);
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<WCO, WPX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int M = a.B * a.Ho * a.Wo;
    const int px_tiles = (M + BPX - 1) / BPX;
    dim3 grid(px_tiles * (a.Cout / BCO), 1, a.splits);
    hipLaunchKernelGGL((conv_mfma_kernel<WCO, WPX>), grid, dim3(256), lds, s, a);
}

template <int WCO, int WPX, int NSTAGE, int ABL = 0>
void launch_glds_t(const ConvMfmaArgs &a, hipStream_t s) {
    constexpr int BCO = WCO * 64, BPX = WPX * 64;
    const size_t lds = (size_t)NSTAGE * (BCO + BPX) * 64 * sizeof(half_t);
    static_assert(NSTAGE * (BCO + BPX) * 64 * 2 >= 4 * 32 * 68 * 4, "ring must hold the epilogue transpose buffer");
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_glds_kernel<WCO, WPX, NSTAGE, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    }
    const int M = a.B * a.Ho * a.Wo;
    const int px_tiles = (M + BPX - 1) / BPX;
    dim3 grid(px_tiles * (a.Cout / BCO), 1, a.splits);
    hipLaunchKernelGGL((conv_glds_kernel<WCO, WPX, NSTAGE, ABL>), grid, dim3(256), lds, s, a);
}

// strip geometry for the patch kernel; returns false when the layer is not eligible.  nt = pixel tiles per strip (1, 2, 4 or 7: the
// instantiations below).  The strip height follows the BATCH: a strip as tall as fits 7 tiles gives the best weight-fragment reuse, but
// the grids are then sized for >= 128 faces (14x14x256 at 32 faces: 64 workgroups on 256 CUs, and a 32-face pass cost 66 % of a
// 128-face one).  With fewer faces the strips get shorter until the launch has about one workgroup per CU again.
bool patch_geometry(const ConvMfmaArgs &a, int &R, int &n_img, int &pps, bool &single, int &nt) {
    // pair mode measured: 56x56 layers 91-105 -> 87-95 us, but the 112x112 layer 306 -> 348 us (it is HBM-bound: 205 MB in, 205 MB
    // out, and a 2-row strip re-reads its halo rows twice) - so the 112x112 layer stays on the im2col LDS-DMA kernel
    const bool pair = a.Cout == 64 && a.Cin == 64 && a.H <= 56;
    if (!a.wf) return false;  // the strip kernel streams the fragment-ordered weight copy
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || (a.Cout % 128 && !pair) || a.Cin % 64 || a.splits != 1 || a.H != a.W) return false;
    if (a.mode == EPI_PARTIAL) return false;
    if ((a.mode == EPI_BN_ADD_BN || a.mode == EPI_BN_SE) && !(a.sc_stride == 1 && a.sc_h == a.Ho && a.sc_w == a.Wo)) return false;
    single = a.Cin == 64;
    const int co_tiles = pair ? 1 : a.Cout / 128;
    static const bool small_ok = !(frt_tuning_env("FRT_CONV_SMALL_BATCH") && frt_tuning_env("FRT_CONV_SMALL_BATCH")[0] == '0');
    constexpr int kWant = 224;  // workgroups that count as "fills the 256 CUs" (128 / 64 measured at 16 / 32 / 64 faces: 0.89 / 1.20 / 1.68 ms per pass become 0.89 / 1.28 / 1.87 and 1.10 / 1.56 / 1.88, profiles/r03/r03r_kwant.txt)
    auto tiles_for = [](int px) { return px <= 32 ? 1 : (px <= 64 ? 2 : (px <= 128 ? 4 : (px <= 224 ? 7 : 0))); };
    auto slots_of = [&](int r, int ni) { return (ni * (r + 2) * (a.W + 2) * 9 + 255) / 256; };
    auto fits = [&](int r, int ni, int t) {  // LDS budget of the instantiation that serves t tiles (see launch_conv_mfma)
        const int sl = slots_of(r, ni);
        if (pair) return t == 7 && ni * (r + 2) * (a.W + 2) * 9 <= 34 * 128;
        if (single) return t == 7 && sl <= 15;
        return t == 7 ? sl <= 12 : (t == 4 ? sl <= 10 : sl <= 5);
    };
    R = 0;
    n_img = 1;
    nt = 0;
    if (a.H * a.W <= 56) {
        // whole small images per strip: 2 images (98 pixels, 4 tiles) put 7x7x512 at 64 strips x 4 cout tiles = 256 workgroups for 128
        // faces (4 images / 7 tiles would leave 128); below 128 faces one image per strip (2 tiles)
        static const int small_nt = frt_tuning_env("FRT_CONV_SMALL_NT") ? atoi(frt_tuning_env("FRT_CONV_SMALL_NT")) : 4;
        for (int ni : {(small_nt * 32) / (a.H * a.W), 1}) {
            if (ni < 1) continue;
            const int t = tiles_for(ni * a.H * a.W);
            if (!t || !fits(a.H, ni, t)) continue;
            if (!R || (small_ok && ((a.B + n_img - 1) / n_img) * co_tiles < kWant)) {
                R = a.H;
                n_img = ni;
                nt = t;
            }
        }
        if (!R) return false;
    } else {
        static const int lim14 = frt_tuning_env("FRT_CONV_NT4_14") ? 128 : 224;  // experiment: half-image strips (4 tiles) on the 14x14 layers
        const int lim = a.H == 14 ? lim14 : 224;
        static const bool ragged_ok = !(frt_tuning_env("FRT_CONV_RAGGED") && frt_tuning_env("FRT_CONV_RAGGED")[0] == '0');
        for (int d = a.H; d >= 1; --d) {  // tallest strip first
            if (d * a.W > lim) continue;
            const int n_str = (a.H + d - 1) / d;
            if (a.H % d) {
                // Ragged last strip (rows past the image are padding in the patch and dead in the epilogue): only the short two-tile strips
                // of a medium batch, in the padded enumeration, with at most 1/7 of the rows wasted - 14x14 in strips of 4 rows puts 24 - 48
                // faces at ONE round of two-tile workgroups where two-row strips take two rounds of one-tile workgroups, each of which
                // streams the same 590 KB of weights (pass of 28 / 32 / 40 / 48 / 55 faces: 1.15 / 1.22 / 1.54 / 1.60 / 1.84 -> 1.08 / 1.12 / 1.42 / 1.49 /
                // 1.57 ms, profiles/r03/r03z_ragged.txt)
                if (!ragged_ok || !small_ok || (n_str * d - a.H) * 7 > a.H || tiles_for(d * (a.W + 2)) != 2 || d * (a.W + 2) > 64) continue;
                if (a.B * n_str * co_tiles < kWant) continue;  // (never the last resort: the divisor strips cover that)
            }
            // slots are enumerated over the padded row width when that still fits the same number of tiles (conflict-free LDS reads)
            int t = tiles_for(d * (a.W + 2));
            if (!t || t != tiles_for(d * a.W)) t = tiles_for(d * a.W);
            if (!t || !fits(d, 1, t)) continue;
            if (d * a.W * 10 < t * 32 * 7) continue;  // more than 30 % dead pixel slots
            R = d;
            nt = t;
            if (!small_ok || a.B * n_str * co_tiles >= kWant) break;  // (else: keep shortening; the shortest eligible strip stays)
        }
        if (!R) return false;
    }
    pps = slots_of(R, n_img);  // DMA slots (1 KB per wave each) the patch image needs
    return true;
}

template <int PPS, int PT, int NW, bool SINGLE, int ABL = 0, bool PAIR = false, int NT = 7, int BFD = 2, int WR = 3, bool SEP = false>
void launch_patch_t(const ConvMfmaArgs &a, int R, int n_img, hipStream_t s) {
    const size_t lds = PAIR ? (size_t)2 * 34 * 2048 : (size_t)(SINGLE ? 1 : 2) * PT * PPS * 4096;  // patch buffers only (weights live in registers)
    static_assert(PAIR || ((SINGLE ? 1 : 2) * PT * PPS * 4096 <= 160 * 1024 && PT * PPS * 4096 >= 4 * 32 * 36 * 4), "LDS budget / epilogue scratch");
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_patch_kernel<PPS, PT, NW, SINGLE, ABL, PAIR, NT, BFD, WR, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    }
    const int strips = ((a.B + n_img - 1) / n_img) * ((a.H + R - 1) / R);
    dim3 grid(PAIR ? (strips + 1) / 2 : strips * (a.Cout / 128));
    const int linear = (n_img == 1 && R * (a.W + 2) <= NT * 32) ? 1 : 0;
    hipLaunchKernelGGL((conv_patch_kernel<PPS, PT, NW, SINGLE, ABL, PAIR, NT, BFD, WR, SEP>), grid, dim3(256), lds, s, a, R, n_img, linear);
}

// compact strips: rows of the stacked-image patch window the tallest strip of the launch needs (top halo + the rows its pixels touch, zero
// separators included + bottom halo), or 0 when the layer / batch is not served by conv_patchc_kernel<NT>
int compact_patch_rows(const ConvMfmaArgs &a, int nt) {
    const int P = a.H * a.W, S = nt * 32;
    const long M = (long)a.B * P;
    int worst = 0;
    // the (image, row) phase of a strip's first pixel repeats every lcm(P, S) pixels: walk one period (or the whole batch if shorter)
    long period = (long)P * S;
    for (long m = 0; m < M && m < period; m += S) {
        const long last = (m + S - 1 < M ? m + S - 1 : M - 1);
        const int sr0 = (int)(m / P) * (a.H + 1) + (int)(m % P) / a.W, sr1 = (int)(last / P) * (a.H + 1) + (int)(last % P) / a.W;
        worst = sr1 - sr0 + 3 > worst ? sr1 - sr0 + 3 : worst;
    }
    return worst;
}

template <int NT, int ABL = 0>
void launch_patchc_t(const ConvMfmaArgs &a, int npr, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 10 * 4096;
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_patchc_kernel<NT, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long M = (long)a.B * a.H * a.W;
    const int strips = (int)((M + NT * 32 - 1) / (NT * 32));
    hipLaunchKernelGGL((conv_patchc_kernel<NT, ABL>), dim3(strips * (a.Cout / 128)), dim3(256), lds, s, a, npr);
}

int conv_impl() {  // FRT_CONV_IMPL: 1 = v1 register-staged, 2 = LDS-DMA 2-stage (default: 64 KB ring, 2 workgroups per CU), 3 = LDS-DMA 3-stage
    static int impl = -1;
    if (impl < 0) {
        const char *e = frt_tuning_env("FRT_CONV_IMPL");
        impl = e ? atoi(e) : 2;
        if (impl < 1 || impl > 3) impl = 2;
    }
    return impl;
}

}  // namespace

// Which kernel symbol a launch resolves to (also the profiling label, so bench.py / rocprofv3 can be matched by name).
enum { CV_V1_22, CV_V1_14, CV_G2_22, CV_G2_14, CV_G3_22, CV_G3_14, CV_P_PAIR, CV_P_SINGLE, CV_P_255, CV_P_264, CV_P_255_NT4, CV_P_NT2, CV_P_NT1, CV_PC_7, CV_PC_4 };
static int conv_variant(const ConvMfmaArgs &a, int &R, int &n_img) {
    const int impl = conv_impl();
    static const int use_patch = frt_tuning_env("FRT_CONV_PATCH") ? atoi(frt_tuning_env("FRT_CONV_PATCH")) : 1;
    int slots, nt;
    bool single;
    if (impl >= 2 && use_patch && patch_geometry(a, R, n_img, slots, single, nt)) {
        if (a.Cout == 64) return CV_P_PAIR;   // pair mode: 2 strips x 68 KB patch
        if (single) return CV_P_SINGLE;       // 15 slots (60 KB)
        if (nt == 1) return CV_P_NT1;         // short strips for small batches: 2 x 20 KB patch buffers
        if (nt == 2) return CV_P_NT2;
        // compact strips (round 6) wherever the full-batch geometry leaves dead pixel slots: whole 14x14 images in 7 tiles (196 of 224 slots
        // live) -> 8 images per 7 strips; two 7x7 images in 4 tiles (98 of 128) -> 128 images per 49 strips.  Not for the fused SE tail (it
        // pools per image inside a strip) - conv_se_fused asks with the SE scratch set.
        static const bool compact_on = !(frt_tuning_env("FRT_CONV_COMPACT") && frt_tuning_env("FRT_CONV_COMPACT")[0] == '0');
        const bool epi_ok = a.mode == EPI_PRELU || a.mode == EPI_BN || (a.mode == EPI_BN_ADD_BN && !a.se_pool);
        if (compact_on && epi_ok && slots <= 10 && ((nt == 7 && n_img == 1 && R == a.H && a.H * a.W < 224) || (nt == 4 && n_img == 2 && 2 * a.H * a.W < 128))) {
            const int npr = compact_patch_rows(a, nt);
            if (npr > 0 && (npr * a.W + 1) * 9 <= 10 * 256) {
                R = npr;
                return nt == 7 ? CV_PC_7 : CV_PC_4;
            }
        }
        if (nt == 4) return CV_P_255_NT4;     // 4 pixel tiles per strip (small maps)
        return slots <= 10 ? CV_P_255 : CV_P_264;  // 2 x 40 KB / 2 x 48 KB patch buffers
    }
    const bool wide = a.Cout % 128 == 0;
    if (impl == 1) return wide ? CV_V1_22 : CV_V1_14;
    if (impl == 2) return wide ? CV_G2_22 : CV_G2_14;
    return wide ? CV_G3_22 : CV_G3_14;
}

const char *conv_kernel_label(const ConvMfmaArgs &a) {
    static const char *names[] = {"conv_mfma_kernel<2, 2>", "conv_mfma_kernel<1, 4>", "conv_glds_kernel<2, 2, 2, 0>", "conv_glds_kernel<1, 4, 2, 0>",
                                  "conv_glds_kernel<2, 2, 3, 0>", "conv_glds_kernel<1, 4, 3, 0>", "conv_patch_kernel<3, 5, 5, true, 0, true, 7, 2, 3>",
                                  "conv_patch_kernel<3, 5, 5, true, 0, false, 7, 1, 3>", "conv_patch_kernel<10, 1, 5, false, 0, false, 7, 1, 3>",
                                  "conv_patch_kernel<2, 6, 4, false, 0, false, 7, 2, 3>", "conv_patch_kernel<10, 1, 5, false, 0, false, 4, 1, 3>",
                                  "conv_patch_kernel<5, 1, 5, false, 0, false, 2, 1, 3>", "conv_patch_kernel<5, 1, 5, false, 0, false, 1, 1, 3>",
                                  "conv_patchc_kernel<7, 0>", "conv_patchc_kernel<4, 0>"};
    if (conv_small_applies(a)) return a.mode == EPI_BN_ADD_BN && a.scx ? "conv_small_kernel<true>" : "conv_small_kernel<false>";
    if (conv_ks_applies(a)) return "conv_ks_kernel";
    if (const char *l2 = conv_s2_label(a)) return l2;
    if (conv64_applies(a))
        return a.mode == EPI_PRELU ? "conv64_kernel<0, 0>" : (a.mode == EPI_BN ? "conv64_kernel<1, 0>" : "conv64_kernel<2, 0>");
    int R, n_img;
    const int variant = conv_variant(a, R, n_img);
    const char *base = names[variant];
    if (!strncmp(base, "conv_patch_kernel", 17)) {  // the strip kernel's symbol carries a tenth argument: SE tail in the epilogue or not
        static thread_local char buf[96];
        snprintf(buf, sizeof(buf), "%.*s, %s>", (int)strlen(base) - 1, base, a.mode == EPI_BN_SE ? "true" : "false");
        return buf;
    }
    return base;
}

// IR-SE: can the launch of `a` (conv2 of a unit described as EPI_BN_ADD_BN, se_* scratch set) run the whole SE tail in its epilogue
// (mode EPI_BN_SE)?  Only the strip kernel's main variant with row-range strips of single images can; everything else writes the BN
// output and leaves the tail to launch_se.
// The fused tail makes the (at most SE_SPLIT x 4) workgroups of one face wait for each other inside a launch, so they must be able to be
// resident together whatever else runs: checked once per device from the runtime's own occupancy figures for the two strip-kernel
// instantiations that carry the tail (a device on which fewer than 16 of their workgroups fit - a much smaller part, a broken LDS opt-in -
// gets the stand-alone tail instead of a hand-over that could starve).
static bool se_fused_fits_device() {
    static int ok[FRT_MAX_DEVICES] = {};  // 0 unknown, 1 yes, -1 no
    int d = 0;
    (void)hipGetDevice(&d);
    d &= FRT_MAX_DEVICES - 1;
    if (!ok[d]) {
        hipDeviceProp_t prop;
        int per_cu7 = 0, per_cu4 = 0;
        const size_t lds7 = (size_t)2 * 10 * 4096, lds4 = (size_t)2 * 10 * 4096;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_patch_kernel<10, 1, 5, false, 0, false, 7, 1, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds7);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_patch_kernel<10, 1, 5, false, 0, false, 4, 1, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        const bool q = hipGetDeviceProperties(&prop, d) == hipSuccess &&
                       hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu7, conv_patch_kernel<10, 1, 5, false, 0, false, 7, 1, 3, true>, 256, lds7) == hipSuccess &&
                       hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, conv_patch_kernel<10, 1, 5, false, 0, false, 4, 1, 3, true>, 256, lds4) == hipSuccess;
        ok[d] = (q && (long)per_cu7 * prop.multiProcessorCount >= 16 && (long)per_cu4 * prop.multiProcessorCount >= 16) ? 1 : -1;
    }
    return ok[d] > 0;
}

bool conv_se_fused(const ConvMfmaArgs &a0) {
    ConvMfmaArgs a = a0;
    a.mode = EPI_BN_ADD_BN;  // same eligibility as the plain unit tail (shortcut with the output's geometry)
    if (!a.se_pool || !a.sc || !a.out1 || conv_small_applies(a) || conv_ks_applies(a) || conv64_applies(a)) return false;
    if (!se_fused_fits_device()) return false;
    if (conv_s2_applies(a)) return conv_s2_se_fused(a);
    int R = 0, n_img = 0;
    const int v = conv_variant(a, R, n_img);
    if (v == CV_P_255) return n_img == 1 && a.H / R <= SE_SPLIT;               // row ranges of one image
    if (v == CV_P_255_NT4) return n_img == 2 ? R == a.H : a.H / R <= SE_SPLIT;  // two whole small images, or row ranges of one
    return false;
}

void launch_conv_mfma(const ConvMfmaArgs &a, hipStream_t s) {
    if (launch_conv_small(a, s)) return;  // a small batch's few pixel tiles: one (32 couts x 32 pixels) unit per workgroup (kernels_arc_small.hip)
    if (launch_conv_ks(a, s)) return;     // medium batches at 14x14x256 / 7x7x512: one 32-cout block x a strip, K split over the waves (kernels_arc_ks.hip)
    if (launch_conv64(a, s)) return;  // dedicated 64 -> 64 stride-1 kernel (kernels_arc_c64.hip)
    if (launch_conv_s2(a, s)) return;  // stride-2 strip kernel on de-interleaved phase planes (kernels_arc_s2.hip)
    int R = 0, n_img = 0;
    const int v = conv_variant(a, R, n_img);
#ifdef FRT_ABLATE
    static const int abl = frt_tuning_env("FRT_CONV_ABLATE") ? atoi(frt_tuning_env("FRT_CONV_ABLATE")) : 0;  // timing experiments only (make TUNING=1)
#endif
    switch (v) {
        case CV_P_PAIR: return launch_patch_t<3, 5, 5, true, 0, true>(a, R, n_img, s);
        case CV_P_SINGLE: return launch_patch_t<3, 5, 5, true, 0, false, 7, 1>(a, R, n_img, s);
        case CV_P_255:
#ifdef FRT_ABLATE
            if (abl == 1) return launch_patch_t<2, 5, 5, false, 1>(a, R, n_img, s);
            if (abl == 2) return launch_patch_t<2, 5, 5, false, 2>(a, R, n_img, s);
            if (abl == 4) return launch_patch_t<2, 5, 5, false, 4>(a, R, n_img, s);
            if (abl == 5) return launch_patch_t<2, 5, 5, false, 5>(a, R, n_img, s);
            if (abl == 6) return launch_patch_t<2, 5, 5, false, 6>(a, R, n_img, s);
            if (abl == 7) return launch_patch_t<2, 5, 5, false, 7>(a, R, n_img, s);
            if (abl == 8) return launch_patch_t<2, 5, 5, false, 8>(a, R, n_img, s);
            if (abl == 9) return launch_patch_t<2, 5, 5, false, 9>(a, R, n_img, s);
            if (abl == 18) return launch_patch_t<5, 2, 5, false, 0, false, 7, 1>(a, R, n_img, s);
            if (abl == 13) return launch_patch_t<2, 5, 5, false, 13, false, 7, 1>(a, R, n_img, s);
            if (abl == 14) return launch_patch_t<2, 5, 5, false, 14, false, 7, 1>(a, R, n_img, s);
            if (abl == 15) return launch_patch_t<2, 5, 5, false, 1, false, 7, 1>(a, R, n_img, s);
            if (abl == 12) return launch_patch_t<2, 5, 5, false, 0, false, 7, 2, 9>(a, R, n_img, s);
            if (abl == 11) return launch_patch_t<2, 5, 5, false>(a, R, n_img, s);  // two-deep B ring, one wave per SIMD
            if (abl == 19) return launch_patch_t<2, 5, 5, false, 0, false, 7, 1>(a, R, n_img, s);  // patch pieces spread over taps 0-4
            if (abl == 20) return launch_patch_t<10, 1, 5, false, 20, false, 7, 1>(a, R, n_img, s);
            if (abl == 22) return launch_patch_t<10, 1, 5, false, 2, false, 7, 1>(a, R, n_img, s);
            if (abl == 24) return launch_patch_t<10, 1, 5, false, 4, false, 7, 1>(a, R, n_img, s);
            if (abl == 29) return launch_patch_t<10, 1, 5, false, 9, false, 7, 1>(a, R, n_img, s);
#endif
            if (a.mode == EPI_BN_SE)  // IR-SE conv2 with the whole SE tail in the epilogue (the caller checked conv_se_fused)
                return launch_patch_t<10, 1, 5, false, 0, false, 7, 1, 3, true>(a, R, n_img, s);
            // (Round 3, built, measured and parked in tools/experiments/conv_patch2_two_cout_blocks_per_wave.hip: a wave owning TWO cout
            //  blocks x half the pixel tiles, so that a B fragment read from LDS feeds two MFMAs - half the LDS bytes per MFMA at 256
            //  registers, parity tests green.  40.9 -> 43.2 us per launch, pipelined step 3.274 -> 3.348 ms (profiles/r03/r03g_patch2_*):
            //  the 4 + 3 split of 7 tiles puts 8 MFMA slots per kk step on the critical SIMDs, and the K loop was never LDS-bound - it
            //  runs at 0.94 of the rate the part sustains for its instruction mix.)
            return launch_patch_t<10, 1, 5, false, 0, false, 7, 1>(a, R, n_img, s);
        case CV_P_264: return launch_patch_t<2, 6, 4, false>(a, R, n_img, s);
        case CV_PC_7:
#ifdef FRT_ABLATE
            if (abl == 22) return launch_patchc_t<7, 2>(a, R, s);
            if (abl == 24) return launch_patchc_t<7, 4>(a, R, s);
            if (abl == 25) return launch_patchc_t<7, 5>(a, R, s);
#endif
            return launch_patchc_t<7>(a, R, s);
        case CV_PC_4: return launch_patchc_t<4>(a, R, s);
        case CV_P_255_NT4:
#ifdef FRT_ABLATE
            if (abl == 11) return launch_patch_t<2, 5, 5, false, 0, false, 4>(a, R, n_img, s);
            if (abl == 19) return launch_patch_t<2, 5, 5, false, 0, false, 4, 1>(a, R, n_img, s);
#endif
            if (a.mode == EPI_BN_SE) return launch_patch_t<10, 1, 5, false, 0, false, 4, 1, 3, true>(a, R, n_img, s);  // (conv_se_fused)
            return launch_patch_t<10, 1, 5, false, 0, false, 4, 1>(a, R, n_img, s);
        // (Round 3, measured and not kept: a 9-deep weight ring for these short-strip variants - 1 or 2 accumulator tiles leave the
        //  registers for it.  4 / 16 / 32 faces: 12.4 -> 12.0, 14.0 -> 13.5, 18.7 -> 18.9 us per launch, batch-1 call 1.286 -> 1.280 ms
        //  (profiles/r03/r03f_small_batch_wr.txt): a small-batch launch is prologue + four chunk hand-overs + epilogue + dispatch, not
        //  weight latency.)
        case CV_P_NT2:
#ifdef FRT_ABLATE
            if (abl == 1) return launch_patch_t<5, 1, 5, false, 1, false, 2, 1>(a, R, n_img, s);
            if (abl == 2) return launch_patch_t<5, 1, 5, false, 2, false, 2, 1>(a, R, n_img, s);
            if (abl == 5) return launch_patch_t<5, 1, 5, false, 5, false, 2, 1>(a, R, n_img, s);
            if (abl == 6) return launch_patch_t<5, 1, 5, false, 6, false, 2, 1>(a, R, n_img, s);
#endif
            return launch_patch_t<5, 1, 5, false, 0, false, 2, 1>(a, R, n_img, s);
        case CV_P_NT1:
#ifdef FRT_ABLATE
            if (abl == 1) return launch_patch_t<5, 1, 5, false, 1, false, 1, 1>(a, R, n_img, s);
            if (abl == 2) return launch_patch_t<5, 1, 5, false, 2, false, 1, 1>(a, R, n_img, s);
            if (abl == 5) return launch_patch_t<5, 1, 5, false, 5, false, 1, 1>(a, R, n_img, s);
            if (abl == 6) return launch_patch_t<5, 1, 5, false, 6, false, 1, 1>(a, R, n_img, s);
#endif
            return launch_patch_t<5, 1, 5, false, 0, false, 1, 1>(a, R, n_img, s);
        case CV_V1_22: return launch_conv_t<2, 2>(a, s);
        case CV_V1_14: return launch_conv_t<1, 4>(a, s);
        case CV_G2_22:
#ifdef FRT_ABLATE
            if (abl == 1) return launch_glds_t<2, 2, 2, 1>(a, s);
            if (abl == 2) return launch_glds_t<2, 2, 2, 2>(a, s);
#endif
            return launch_glds_t<2, 2, 2>(a, s);
        case CV_G2_14: return launch_glds_t<1, 4, 2>(a, s);
        case CV_G3_22: return launch_glds_t<2, 2, 3>(a, s);
        default: return launch_glds_t<1, 4, 3>(a, s);
    }
}

void launch_arc_input(const ArcInputArgs &a, hipStream_t s) {
    static const bool use_mfma = [] {
        const char *e = frt_tuning_env("FRT_ARC_INPUT_MFMA");
        return !(e && e[0] == '0');
    }();
    if (use_mfma && launch_arc_input_mfma(a, s)) return;
    const long total = (long)a.F * a.H * a.W * 8;  // 8 lanes per pixel
    hipLaunchKernelGGL(arc_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

void launch_fc_finalize(const float *partial, int splits, int F, const float *bias, const float *s, const float *b, const int *valid,
                        float *out, hipStream_t st) {
    hipLaunchKernelGGL(fc_finalize_kernel, dim3(F), dim3(512), 0, st, partial, splits, F, bias, s, b, valid, out);
}

void launch_se(const SeArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(se_pool_gate_kernel, dim3(SE_SPLIT, a.F), dim3(256), 0, s, a.res, a.H * a.W, a.C, a.F, a.pool, a.w1, a.w2, a.gate, a.counter);
    const long total = (long)a.F * a.H * a.W * (a.C / 8);
    hipLaunchKernelGGL(se_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}
