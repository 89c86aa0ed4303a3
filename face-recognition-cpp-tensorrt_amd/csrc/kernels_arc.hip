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
#include "frt_kernels.h"

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
            if (p.mode == EPI_BN_ADD_BN && p.out1) {
                half8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[e] * q2[e >> 2][e & 3] + q3[e >> 2][e & 3]);
                *reinterpret_cast<half8 *>(p.out1 + (long)m * p.Cout + c) = z;
            }
        }
    }
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
    *reinterpret_cast<half8 *>(a.y + gp * 64 + cb) = y8;
    *reinterpret_cast<half8 *>(a.z + gp * 64 + cb) = z8;
}

// ---------------------------------------------------------------- Linear split-K reduce + bias + BatchNorm1d + L2 normalise
__global__ __launch_bounds__(512) void fc_finalize_kernel(const float *__restrict__ partial, int splits, int F, const float *__restrict__ bias,
                                                          const float *__restrict__ s, const float *__restrict__ b,
                                                          const int *__restrict__ valid, float *__restrict__ out) {
    const int f = blockIdx.x, o = threadIdx.x;
    float v = 0.f;
    for (int k = 0; k < splits; ++k) v += partial[((long)k * F + f) * 512 + o];
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
__global__ __launch_bounds__(256) void se_pool_kernel(const half_t *__restrict__ res, int HW, int C, float *__restrict__ pool) {
    // grid (C/256 or 1, F); thread = channel; coalesced across channels (NHWC)
    const int c = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (c >= C) return;
    const half_t *p = res + (long)f * HW * C + c;
    float s = 0.f;
    for (int i = 0; i < HW; ++i) s += (float)p[(long)i * C];
    pool[(long)f * C + c] = s / (float)HW;
}
__global__ __launch_bounds__(256) void se_gate_kernel(const float *__restrict__ pool, const float *__restrict__ w1, const float *__restrict__ w2,
                                                      int C, float *__restrict__ gate) {
    // one block per face
    extern __shared__ float sh[];  // [C] pooled + [C/16] hidden
    const int f = blockIdx.x, R = C / 16;
    float *sp = sh, *shid = sh + C;
    for (int c = threadIdx.x; c < C; c += 256) sp[c] = pool[(long)f * C + c];
    __syncthreads();
    for (int h = threadIdx.x; h < R; h += 256) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(w1[(long)h * C + c], sp[c], a);
        shid[h] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int h = 0; h < R; ++h) a = fmaf(w2[(long)c * R + h], shid[h], a);
        gate[(long)f * C + c] = 1.f / (1.f + expf(-a));
    }
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
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<WCO, WPX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
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
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_glds_kernel<WCO, WPX, NSTAGE, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_done = true;
    }
    const int M = a.B * a.Ho * a.Wo;
    const int px_tiles = (M + BPX - 1) / BPX;
    dim3 grid(px_tiles * (a.Cout / BCO), 1, a.splits);
    hipLaunchKernelGGL((conv_glds_kernel<WCO, WPX, NSTAGE, ABL>), grid, dim3(256), lds, s, a);
}

int conv_impl() {  // FRT_CONV_IMPL: 1 = v1 register-staged, 2 = LDS-DMA 2-stage (default: 64 KB ring, 2 workgroups per CU), 3 = LDS-DMA 3-stage
    static int impl = -1;
    if (impl < 0) {
        const char *e = getenv("FRT_CONV_IMPL");
        impl = e ? atoi(e) : 2;
        if (impl < 1 || impl > 3) impl = 2;
    }
    return impl;
}

}  // namespace

void launch_conv_mfma(const ConvMfmaArgs &a, hipStream_t s) {
    const int impl = conv_impl();
    const bool wide = a.Cout % 128 == 0;
    if (impl == 1) {
        if (wide) launch_conv_t<2, 2>(a, s);
        else launch_conv_t<1, 4>(a, s);
    } else if (impl == 2) {
        static const int abl = getenv("FRT_CONV_ABLATE") ? atoi(getenv("FRT_CONV_ABLATE")) : 0;  // timing experiments only
        if (wide && abl == 1) return launch_glds_t<2, 2, 2, 1>(a, s);
        if (wide && abl == 2) return launch_glds_t<2, 2, 2, 2>(a, s);
        if (wide) launch_glds_t<2, 2, 2>(a, s);
        else launch_glds_t<1, 4, 2>(a, s);
    } else {
        if (wide) launch_glds_t<2, 2, 3>(a, s);
        else launch_glds_t<1, 4, 3>(a, s);
    }
}

void launch_arc_input(const ArcInputArgs &a, hipStream_t s) {
    const long total = (long)a.F * a.H * a.W * 8;  // 8 lanes per pixel
    hipLaunchKernelGGL(arc_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

void launch_fc_finalize(const float *partial, int splits, int F, const float *bias, const float *s, const float *b, const int *valid,
                        float *out, hipStream_t st) {
    hipLaunchKernelGGL(fc_finalize_kernel, dim3(F), dim3(512), 0, st, partial, splits, F, bias, s, b, valid, out);
}

void launch_se(const SeArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(se_pool_kernel, dim3((a.C + 255) / 256, a.F), dim3(256), 0, s, a.res, a.H * a.W, a.C, a.pool);
    hipLaunchKernelGGL(se_gate_kernel, dim3(a.F), dim3(256), (a.C + a.C / 16) * sizeof(float), s, a.pool, a.w1, a.w2, a.C, a.gate);
    const long total = (long)a.F * a.H * a.W * (a.C / 8);
    hipLaunchKernelGGL(se_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}
