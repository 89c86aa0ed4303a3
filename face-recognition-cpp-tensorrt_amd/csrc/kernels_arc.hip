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

// ---------------------------------------------------------------- input layer: conv3x3 3->64 + BN + PReLU (+ unit-0 leading BN)
// 0.3 % of the FLOPs, K = 27: plain VALU with scalar-path weights, one thread per pixel, fp32 planar in, fp16 NHWC out.
__global__ __launch_bounds__(256) void arc_input_kernel(ArcInputArgs a) {
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    const int HW = a.H * a.W;
    if (gp >= (long)a.F * HW) return;
    const int f = (int)(gp / HW), pix = (int)(gp - (long)f * HW);
    const int oh = pix / a.W, ow = pix - oh * a.W;
    float xin[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh - 1 + kh, iw = ow - 1 + kw;
                const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                xin[ci * 9 + kh * 3 + kw] = ok ? a.x[((long)f * 3 + ci) * HW + ih * a.W + iw] : 0.f;
            }
    half_t *yo = a.y + gp * 64, *zo = a.z + gp * 64;
#pragma unroll
    for (int cb = 0; cb < 64; cb += 8) {
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = fmaf(xin[k], a.w[k * 64 + cb + c], acc[c]);
        half8 y8, z8;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = acc[c] * a.s0[cb + c] + a.b0[cb + c];
            v = v > 0.f ? v : v * a.slope[cb + c];
            y8[c] = (half_t)v;
            z8[c] = (half_t)(v * a.s1[cb + c] + a.b1[cb + c]);
        }
        *reinterpret_cast<half8 *>(yo + cb) = y8;
        *reinterpret_cast<half8 *>(zo + cb) = z8;
    }
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

}  // namespace

void launch_conv_mfma(const ConvMfmaArgs &a, hipStream_t s) {
    if (a.Cout % 128 == 0)
        launch_conv_t<2, 2>(a, s);
    else
        launch_conv_t<1, 4>(a, s);
}

void launch_arc_input(const ArcInputArgs &a, hipStream_t s) {
    const long total = (long)a.F * a.H * a.W;
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
