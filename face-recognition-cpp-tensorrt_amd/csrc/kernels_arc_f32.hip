// ArcFace IR-50 / IR-SE-50 in fp32 end to end (round 5): the recogniser mode BASELINE configs[1] names ("640x640 batch=1 ... fp32").
//
// Arithmetic spec: /root/reference/conversion/arcface/model_irse.py:11-173 (Backbone.forward, bottleneck_IR / _IR_SE, SEModule); the reference
// itself runs the network as a TensorRT fp16 engine (conversion/arcface/torch2trt.py:42-43), this mode is the accuracy reference of this build:
// fp32 activations (NHWC), fp32 weights, every product on v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain) or v_fma_f32 - no fp16 anywhere.
// It is a SEPARATE, simple path for a handful of faces per call (frt_embedder_set_precision(e, 1)); the fp16-MFMA kernels of kernels_arc*.hip
// stay the default and the throughput path.
//
//   conv32_kernel   implicit GEMM, one workgroup = 32 output pixels x 32 output channels, K = ks*ks*Cin split over the four waves in units of 8
//                   channels of one tap; lane (r, hi) loads 4 consecutive channels (offset 4 hi) of weight row r / pixel r as one float4 and
//                   issues four MFMAs whose k pairs are (c, c + 4) - any pairing is fine as long as A and B agree.  Optional prologue: the
//                   unit's LEADING BatchNorm applied to the in-image pixels on load (zero padding stays zero: SURVEY App. C.9 - in fp32 there
//                   is no reason to materialise that tensor).  The four partial tiles meet in LDS and are added in wave order.
//                   Epilogues: PReLU | BN | BN + shortcut (a tensor sampled with a stride: MaxPool2d(1, s) or the 1x1-conv shortcut's output).
//   input / fc / SE small direct kernels (3 -> 64 conv + BN + PReLU; Linear 25088 -> 512 over the NHWC flatten with output_layer.0's BN applied
//                   on load; SE gate + apply).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "frt_kernels.h"

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---------------------------------------------------------------- input layer: conv3x3(3 -> 64) + BN + PReLU, planar in, NHWC out
__global__ __launch_bounds__(256) void arc32_input_kernel(const float *__restrict__ x, const float *__restrict__ w /*[27][64]*/, const float *__restrict__ s0,
                                                          const float *__restrict__ b0, const float *__restrict__ slope, float *__restrict__ y, int F) {
    __shared__ float ws[27 * 64];
    for (int i = threadIdx.x; i < 27 * 64; i += 256) ws[i] = w[i];
    __syncthreads();
    const long g = (long)blockIdx.x * 256 + threadIdx.x;  // (pixel, channel quad)
    const long total = (long)F * 112 * 112 * 16;
    if (g >= total) return;
    const int cq = (int)(g & 15);
    const long pix = g >> 4;
    const int f = (int)(pix / (112 * 112)), p = (int)(pix - (long)f * 112 * 112), oy = p / 112, ox = p - oy * 112;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ci = 0; ci < 3; ++ci)
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                const int iy = oy - 1 + kh, ix = ox - 1 + kw;
                const float v = (iy >= 0 && iy < 112 && ix >= 0 && ix < 112) ? x[((long)f * 3 + ci) * 112 * 112 + iy * 112 + ix] : 0.f;
                const float *wr = ws + (ci * 9 + kh * 3 + kw) * 64 + cq * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(v, wr[e], acc[e]);
            }
    floatx4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cq * 4 + e;
        const float v = acc[e] * s0[c] + b0[c];
        o[e] = v > 0.f ? v : v * slope[c];
    }
    *reinterpret_cast<floatx4 *>(y + pix * 64 + cq * 4) = o;
}

// ---------------------------------------------------------------- generic fp32 conv (3x3 pad 1 / 1x1 pad 0, stride 1 or 2), NHWC
struct Conv32 {
    const float *x;        // [F][H][W][Cin]
    const float *w;        // fragment-ordered copy of [Cout][ks*ks][Cin] (pack_conv32_weights)
    const float *ps, *pb;  // optional prologue BN per input channel (null: none)
    float *out;            // [F][Ho][Wo][Cout]
    int F, H, W, Cin, Ho, Wo, Cout, ks, stride, pad;
    int mode;              // 0: PReLU(p0)  1: BN(p0, p1)  2: BN(p0, p1) + shortcut
    const float *p0, *p1;
    const float *sc;       // [F][sc_h][sc_w][Cout], sampled at (oy * sc_stride, ox * sc_stride)
    int sc_h, sc_w, sc_stride;
};

// Round 6.  The first version (one workgroup = 32 pixels x 32 couts, K split over the four waves, every lane fetching one float4 of "its" weight
// row and one of "its" pixel straight into the MFMA operand registers, one unit ahead) ran at 0.13 of the fp32 matrix rate at 4 - 9 faces:
//   * one (weight, pixel) pair in flight per wave with less than one wave per SIMD: every step paid a whole L2 / HBM round trip (~ 0.7 us
//     against 0.1 us of MFMAs), plus two run-time integer divisions in front of every load;
//   * made D deep (a register ring, straight-line code, shifts for the divisions) it reached 0.21: the loads themselves were the limit - lane
//     (r, hi) reads 16 bytes of row / pixel r, so ONE wave-level load touches 32 different 128-byte lines (32 bytes of each): four times the
//     address / tag work per byte (the same finding as the fp16 path's row-major weight fragments, DESIGN A.1 round 2).
// Now both operands arrive as whole lines:
//   * weights come from a FRAGMENT-ORDERED copy packed by the host ([32-cout tile][K block of 32 channels of one tap][q][lane][4 floats]): a
//     wave-level load is one contiguous kilobyte;
//   * pixels are fetched 8 lanes per pixel (128 contiguous bytes = one line, 8 lines per wave-level load), get the unit's leading BatchNorm and
//     the zero padding applied in registers, and are transposed into MFMA operand order through a PRIVATE per-wave LDS tile (144-byte pixel
//     rows: conflict-free ds_read_b128; no workgroup barrier - LDS executes a wave's accesses in order);
//   * a wave's K share is whole 32-channel blocks (16 MFMAs each); R blocks are in flight in a register ring with compile-time slots, the code
//     is straight-line (loads unconditional from clamped addresses, validity bits applied at the LDS write), so every vmcnt is an exact count.
// The arithmetic per output is the old kernel's: the same products, per wave in channel order within a block, blocks in the same order, the
// four waves' partial sums added in wave order.
// K blocks in flight per wave: 4, or 3 with the prologue BatchNorm's extra quads (the ring then fits two waves per SIMD: the 112x112 and
// 56x56 layers launch 1 500 - 3 000 workgroups and want the second wave)

// (third finding of the round, after the loads were deep and line-shaped and the time had not moved: with ONE wave per SIMD the instruction stream is
//  in-order - an MFMA that waits for the matrix pipe blocks everything behind it, so the ~ 130 address / conversion instructions per block that
//  the compiler had placed behind the 16 MFMAs ran AFTER them, not under them: 1024 + ~ 1300 cycles per block.  Hence: 32-bit byte offsets against
//  uniform base pointers, 32-bit index divisions in prologue and epilogue, per-pixel tap-validity masks computed once, the LDS hand-over one
//  block ahead of its use, and the step's instruction order pinned by hand (one group per MFMA).
//  Where a 14x14x256 launch at 4 faces (200 workgroups, 21.5 us) goes - timing ablations, profiles/r06_conv32_ablation.txt: launch + prologue +
//  epilogue 4.3 us; the K loop WITHOUT its MFMAs 10.7 us (118 MB from L2 per launch - 32 x 32 tiles re-read every weight 25 times and every
//  pixel 8 x 9 times - at the ~ 11 TB/s the L2 -> CU path delivers); the 288 MFMAs of a wave 8.7 us at 30 ns each (tools/ubench/mfma_f32_chain:
//  the dependent chain costs nothing) - and the two ADD instead of overlapping, whatever the order.  Fewer bytes need larger tiles, larger tiles
//  leave fewer than 800 waves for 1 024 SIMDs at 4 faces: the mode stays a small-batch accuracy path.)
template <bool PRO, int ABL = 0>  // ABL (measurement builds): 1 = no K loop, 2 = K loop without the MFMAs
__global__ __launch_bounds__(256) void conv32_kernel(Conv32 a, int cb_log2) {
    constexpr int C32_R = PRO ? 3 : 4;
    __shared__ float red[4][32][33];
    __shared__ __attribute__((aligned(16))) float stage[4][2][32 * 36];  // [wave][buffer][pixel row of 36 floats: 32 channels + 16 bytes pad]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hi = lane >> 5;
    // (32-bit index arithmetic throughout: the first versions divided 64-bit pixel indices - ~ 200 instructions per division, ten of them per
    //  lane in front of the first load and in the epilogue: several microseconds of a 23 us launch)
    const unsigned HoWo = (unsigned)(a.Ho * a.Wo);
    const unsigned M = (unsigned)a.F * HoWo;
    const unsigned m0 = blockIdx.x * 32u;
    const int co0 = blockIdx.y * 32;
    const int CB = 1 << cb_log2;            // 32-channel blocks per tap
    const int NB = a.ks * a.ks * CB;        // K blocks of the layer
    const int NI = ABL == 1 ? 0 : (NB + 3) >> 2;           // this wave's iterations (block wave + 4 it; blocks >= NB are zero operands)
    // loader role: lane = (pixel sub-index lane >> 3, 16-byte chunk lane & 7); pixel of slot i: (lane >> 3) + 8 i.  Per pixel, once: the BYTE offset
    // of its tap (0, 0) / channel 4 ch (may be "negative": only ever used with a valid tap's offset added) and the 9 taps' validity bits
    const int lp = lane >> 3, ch = lane & 7;
    int xb0[4];
    unsigned vm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned m = m0 + lp + 8 * i;
        const bool ok = m < M;
        const unsigned mm = ok ? m : 0u;
        const unsigned fu = mm / HoWo, pu = mm - fu * HoWo, oyu = pu / (unsigned)a.Wo;
        const int f = (int)fu, oy = (int)oyu, ox = (int)(pu - oyu * (unsigned)a.Wo);
        const int y0 = oy * a.stride - a.pad, x0 = ox * a.stride - a.pad;
        xb0[i] = (((f * a.H + y0) * a.W + x0) * a.Cin + 4 * ch) * 4;
        unsigned bits = 0;
        for (int kh = 0; kh < a.ks; ++kh)
            for (int kw = 0; kw < a.ks; ++kw)
                if (ok && y0 + kh >= 0 && y0 + kh < a.H && x0 + kw >= 0 && x0 + kw < a.W) bits |= 1u << (kh * a.ks + kw);
        vm[i] = bits;
    }
    const char *xbase = reinterpret_cast<const char *>(a.x);
    const char *wbase = reinterpret_cast<const char *>(a.w) + (size_t)blockIdx.y * NB * 4096;  // this cout tile's fragments: [block][q][lane][4]
    const unsigned wlane = lane * 16, sblane = ch * 16;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    floatx4 wq[C32_R][4], xq[C32_R][4], sq[PRO ? C32_R : 1], bq[PRO ? C32_R : 1];
    unsigned okm = 0;  // 4 validity bits per ring slot
    // global loads of block `it` into ring slot S, in four parts (PART 0: weight fragments, 1 / 2: two pixel quads each, 3: BatchNorm quads)
    auto issue = [&](int it, auto slot_c, auto part_c) {
        constexpr int S = decltype(slot_c)::value, PART = decltype(part_c)::value;
        const int b = wave + 4 * it;      // wave-uniform from here to the offsets
        const bool bok = b < NB;
        const int bc = bok ? b : 0;
        const int tap = bc >> cb_log2, c0 = (bc & (CB - 1)) << 5;
        if constexpr (PART == 0 || PART == 4) {
            const char *wb = wbase + (size_t)bc * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q) wq[S][q] = *reinterpret_cast<const floatx4 *>(wb + wlane + q * 1024);
        }
        if constexpr (PART == 1 || PART == 2 || PART == 4) {
            const int kh = a.ks == 3 ? (tap * 11) >> 5 : 0, kw = tap - kh * a.ks;  // tap / 3 for tap < 9
            const unsigned tapoff = (unsigned)(((kh * a.W + kw) * a.Cin + c0) * 4);
            const unsigned tbit = bok ? 1u << tap : 0u;
#pragma unroll
            for (int i = (PART == 2 ? 2 : 0); i < (PART == 1 ? 2 : 4); ++i) {
                const bool ok = (vm[i] & tbit) != 0;
                const unsigned off = ok ? (unsigned)xb0[i] + tapoff : 0u;  // an invalid tap reads the tensor's first bytes (and is zeroed below)
                xq[S][i] = *reinterpret_cast<const floatx4 *>(xbase + off);
                okm = (okm & ~(1u << (4 * S + i))) | ((ok ? 1u : 0u) << (4 * S + i));
            }
        }
        if constexpr (PRO && (PART == 3 || PART == 4)) {
            sq[S] = *reinterpret_cast<const floatx4 *>(reinterpret_cast<const char *>(a.ps) + (unsigned)(c0 * 4) + sblane);
            bq[S] = *reinterpret_cast<const floatx4 *>(reinterpret_cast<const char *>(a.pb) + (unsigned)(c0 * 4) + sblane);
        }
    };
    // registers of slot S, pixel quad i -> (BatchNorm, zero padding) -> this wave's LDS tile `buf`
    auto to_lds = [&](auto slot_c, auto buf_c, int i) {
        constexpr int S = decltype(slot_c)::value;
        float *st = &stage[wave][decltype(buf_c)::value][0];
        floatx4 v = xq[S][i];
        const bool ok = (okm >> (4 * S + i)) & 1u;
        if constexpr (PRO) {  // leading BatchNorm on in-image pixels only: the conv's zero padding pads the NORMALISED tensor
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sq[S][e] + bq[S][e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
        *reinterpret_cast<floatx4 *>(st + (lp + 8 * i) * 36 + 4 * ch) = v;
    };
    floatx4 bf[2][4];  // B fragments: [block parity][q]
    auto from_lds = [&](auto buf_c, int q) {
        const float *st = &stage[wave][decltype(buf_c)::value][0];
        bf[decltype(buf_c)::value][q] = *reinterpret_cast<const floatx4 *>(st + r * 36 + 8 * q + 4 * hi);
    };
    // Block `it` (slot S, LDS buffer / fragment set S & 1).  The matrix pipe is busy 64 cycles per MFMA and this wave is alone on its SIMD, and
    // the instruction stream is in-order: a wait blocks the MFMAs behind it.  So the order is pinned by hand, one group per MFMA
    // (sched_barrier(0) after each): the global loads of block it + R and the LDS reads of block it + 1 go under the first MFMAs (they wait for
    // nothing), the pixel quads of block it + 2 - the only instructions that wait for memory - go to LDS under the LAST MFMAs, when their loads
    // have had three and a half blocks' time to arrive and twelve MFMAs are already queued in front of the wait.
    auto step = [&](int it, auto slot_c, auto par_c) {
        constexpr int S = decltype(slot_c)::value, P = decltype(par_c)::value;  // ring slot it % R, LDS buffer / fragment set it & 1
        floatx4 wa[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wa[q] = wq[S][q];
        static_for<0, 16>([&](auto gc) {
            constexpr int G = decltype(gc)::value;
            if (ABL == 2) asm volatile("" ::"v"(wa[G >> 2][G & 3]), "v"(bf[P][G >> 2][G & 3]));
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[G >> 2][G & 3], bf[P][G >> 2][G & 3], acc, 0, 0, 0);
            if constexpr (G < 4) issue(it + C32_R, slot_c, std::integral_constant<int, G>{});
            else if constexpr (G < 8) from_lds(std::integral_constant<int, P ^ 1>{}, G - 4);
            else if constexpr (G >= 11 && G < 15) to_lds(std::integral_constant<int, (S + 2) % C32_R>{}, std::integral_constant<int, P>{}, G - 11);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    static_for<0, C32_R>([&](auto ic) { issue(decltype(ic)::value, ic, std::integral_constant<int, 4>{}); });
    for (int i = 0; i < 4; ++i) to_lds(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, i);
    for (int i = 0; i < 4; ++i) to_lds(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, i);
    for (int q = 0; q < 4; ++q) from_lds(std::integral_constant<int, 0>{}, q);
    constexpr int UN = (C32_R & 1) ? 2 * C32_R : C32_R;  // slots and LDS buffers both at compile time: unrolled over their common period
    for (int it0 = 0; it0 < NI; it0 += UN)
        static_for<0, UN>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            step(it0 + I, std::integral_constant<int, I % C32_R>{}, std::integral_constant<int, I & 1>{});
        });
    // partial tile of this wave: acc[e] = (cout (e & 3) + 8 (e >> 2) + 4 hi, pixel r)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * hi][r] = acc[e];
    __syncthreads();
    // thread t: pixel t & 31, channels 4 (t >> 5) .. + 3; the four partials are added in wave order (deterministic)
    const int px = tid & 31, cg = tid >> 5;
    const unsigned mo = m0 + px;
    if (mo >= M) return;
    const unsigned fo = mo / HoWo, po = mo - fo * HoWo, yo = po / (unsigned)a.Wo, xo = po - yo * (unsigned)a.Wo;
    floatx4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int cl = 4 * cg + e, c = co0 + cl;
        float v = ((red[0][cl][px] + red[1][cl][px]) + red[2][cl][px]) + red[3][cl][px];
        if (a.mode == 0) {
            v = v > 0.f ? v : v * a.p0[c];
        } else {
            v = v * a.p0[c] + a.p1[c];
            if (a.mode == 2) v += a.sc[(((long)fo * a.sc_h + (long)yo * a.sc_stride) * a.sc_w + (long)xo * a.sc_stride) * a.Cout + c];
        }
        o[e] = v;
    }
    *reinterpret_cast<floatx4 *>(a.out + (size_t)mo * a.Cout + co0 + 4 * cg) = o;
}

// ---------------------------------------------------------------- Linear 25088 -> 512 over BN2d(y) flattened in NHWC order
// w [512][25088] with k = hw * 512 + c (re-ordered on the host from the reference's NCHW flatten c * 49 + hw, model_irse.py:11-13)
__global__ __launch_bounds__(256) void fc32_kernel(const float *__restrict__ y, const float *__restrict__ sn, const float *__restrict__ bn, const float *__restrict__ w,
                                                   float *__restrict__ out /*[F][512]*/, int F) {
    __shared__ float part[4][4];
    const int o0 = blockIdx.x * 4, f = blockIdx.y, tid = threadIdx.x;
    const float *yf = y + (long)f * 25088;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = tid * 4; k < 25088; k += 1024) {
        const int c = k & 511;
        floatx4 v = *reinterpret_cast<const floatx4 *>(yf + k);
        const floatx4 s = *reinterpret_cast<const floatx4 *>(sn + c), b = *reinterpret_cast<const floatx4 *>(bn + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * s[e] + b[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const floatx4 wv = *reinterpret_cast<const floatx4 *>(w + (long)(o0 + j) * 25088 + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j] = fmaf(v[e], wv[e], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_xor(acc[j], off);
        if ((tid & 63) == 0) part[tid >> 6][j] = acc[j];
    }
    __syncthreads();
    if (tid < 4) out[(long)f * 512 + o0 + tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
}

// ---------------------------------------------------------------- SE tail (IR-SE): gate[f][c] = sigmoid(W2 relu(W1 mean_hw(res))), out = res * gate + shortcut
__global__ __launch_bounds__(256) void se32_gate_kernel(const float *__restrict__ res, const float *__restrict__ w1 /*[C/16][C]*/, const float *__restrict__ w2 /*[C][C/16]*/,
                                                        float *__restrict__ gate, int HW, int C) {
    __shared__ float pool[512];
    __shared__ float hid[32];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float *rf = res + (long)f * HW * C;
    for (int c = tid; c < C; c += 256) {
        float sum = 0.f;
        for (int p = 0; p < HW; ++p) sum += rf[(long)p * C + c];
        pool[c] = sum / (float)HW;
    }
    __syncthreads();
    const int R = C / 16;
    if (tid < R) {
        float h = 0.f;
        for (int c = 0; c < C; ++c) h = fmaf(w1[(long)tid * C + c], pool[c], h);
        hid[tid] = fmaxf(h, 0.f);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float g = 0.f;
        for (int j = 0; j < R; ++j) g = fmaf(w2[(long)c * R + j], hid[j], g);
        gate[(long)f * C + c] = 1.f / (1.f + expf(-g));
    }
}
__global__ __launch_bounds__(256) void se32_apply_kernel(const float *__restrict__ res, const float *__restrict__ gate, const float *__restrict__ sc, float *__restrict__ out, int F,
                                                         int Ho, int Wo, int C, int sc_h, int sc_w, int sc_stride) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)F * Ho * Wo * C;
    if (g >= total) return;
    const int c = (int)(g % C);
    const long pix = g / C;
    const int f = (int)(pix / (Ho * Wo)), p = (int)(pix - (long)f * Ho * Wo), oy = p / Wo, ox = p - oy * Wo;
    out[g] = res[g] * gate[(long)f * C + c] + sc[(((long)f * sc_h + (long)oy * sc_stride) * sc_w + (long)ox * sc_stride) * C + c];
}

}  // namespace

void launch_arc32_input(const float *x, const float *w, const float *s0, const float *b0, const float *slope, float *y, int F, hipStream_t s) {
    const long total = (long)F * 112 * 112 * 16;
    hipLaunchKernelGGL(arc32_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, s0, b0, slope, y, F);
}
void launch_conv32(const Conv32Args &c, hipStream_t s) {
    Conv32 a{c.x, c.w, c.ps, c.pb, c.out, c.F, c.H, c.W, c.Cin, c.Ho, c.Wo, c.Cout, c.ks, c.stride, c.pad, c.mode, c.p0, c.p1, c.sc, c.sc_h, c.sc_w, c.sc_stride};
    const long M = (long)c.F * c.Ho * c.Wo;
    int lg = 0;
    while ((32 << lg) < c.Cin) ++lg;  // Cin / 32 is a power of two (64 ... 512 channels)
    const dim3 grid((unsigned)((M + 31) / 32), (unsigned)(c.Cout / 32));
#ifdef FRT_ABLATE
    static const int xp = frt_tuning_env("FRT_C32_ABLATE") ? atoi(frt_tuning_env("FRT_C32_ABLATE")) : 0;  // timing ablations (make TUNING=1): wrong results by design
    if (xp == 1) { if (c.ps) hipLaunchKernelGGL((conv32_kernel<true, 1>), grid, dim3(256), 0, s, a, lg); else hipLaunchKernelGGL((conv32_kernel<false, 1>), grid, dim3(256), 0, s, a, lg); return; }
    if (xp == 2) { if (c.ps) hipLaunchKernelGGL((conv32_kernel<true, 2>), grid, dim3(256), 0, s, a, lg); else hipLaunchKernelGGL((conv32_kernel<false, 2>), grid, dim3(256), 0, s, a, lg); return; }
#endif
    if (c.ps) hipLaunchKernelGGL((conv32_kernel<true>), grid, dim3(256), 0, s, a, lg);
    else hipLaunchKernelGGL((conv32_kernel<false>), grid, dim3(256), 0, s, a, lg);
}
// host side of the fragment order conv32_kernel reads: w [Cout][ks*ks][Cin] -> [Cout / 32][ks*ks * Cin / 32][q = 0..3][lane = (r, hi)][e = 0..3]
// with element = w[32 t + r][tap][32 cb + 8 q + 4 hi + e]
void pack_conv32_weights(const float *w, int cout, int taps, int cin, float *out) {
    const int CB = cin / 32, NB = taps * CB;
    for (int t = 0; t < cout / 32; ++t)
        for (int b = 0; b < NB; ++b) {
            const int tap = b / CB, cb = b - tap * CB;
            for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 31, hi = lane >> 5;
                    for (int e = 0; e < 4; ++e)
                        out[(((size_t)t * NB + b) * 4 + q) * 256 + lane * 4 + e] = w[((size_t)(32 * t + r) * taps + tap) * cin + 32 * cb + 8 * q + 4 * hi + e];
                }
        }
}
void launch_fc32(const float *y, const float *sn, const float *bn, const float *w, float *out, int F, hipStream_t s) {
    hipLaunchKernelGGL(fc32_kernel, dim3(128, (unsigned)F), dim3(256), 0, s, y, sn, bn, w, out, F);
}
void launch_se32(const float *res, const float *w1, const float *w2, float *gate, const float *sc, float *out, int F, int Ho, int Wo, int C, int sc_h, int sc_w, int sc_stride,
                 hipStream_t s) {
    hipLaunchKernelGGL(se32_gate_kernel, dim3((unsigned)F), dim3(256), 0, s, res, w1, w2, gate, Ho * Wo, C);
    const long total = (long)F * Ho * Wo * C;
    hipLaunchKernelGGL(se32_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, res, gate, sc, out, F, Ho, Wo, C, sc_h, sc_w, sc_stride);
}
