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
#include "frt_kernels.h"

namespace {

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
    const float *w;        // [Cout][ks*ks][Cin]
    const float *ps, *pb;  // optional prologue BN per input channel (null: none)
    float *out;            // [F][Ho][Wo][Cout]
    int F, H, W, Cin, Ho, Wo, Cout, ks, stride, pad;
    int mode;              // 0: PReLU(p0)  1: BN(p0, p1)  2: BN(p0, p1) + shortcut
    const float *p0, *p1;
    const float *sc;       // [F][sc_h][sc_w][Cout], sampled at (oy * sc_stride, ox * sc_stride)
    int sc_h, sc_w, sc_stride;
};

__global__ __launch_bounds__(256) void conv32_kernel(Conv32 a) {
    __shared__ float red[4][32][33];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hi = lane >> 5;
    const long M = (long)a.F * a.Ho * a.Wo;
    const long m0 = (long)blockIdx.x * 32;
    const int co0 = blockIdx.y * 32;
    // this lane's pixel (B operand) and weight row (A operand)
    const long m = m0 + r;
    const bool mok = m < M;
    const long mm = mok ? m : 0;
    const int f = (int)(mm / (a.Ho * a.Wo)), p = (int)(mm - (long)f * a.Ho * a.Wo), oy = p / a.Wo, ox = p - oy * a.Wo;
    const float *wrow = a.w + (long)(co0 + r) * a.ks * a.ks * a.Cin + 4 * hi;
    const float *xf = a.x + (long)f * a.H * a.W * a.Cin + 4 * hi;
    const int cpt = a.Cin >> 3;             // 8-channel units per tap
    const int U = a.ks * a.ks * cpt;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    floatx4 wa, xb, wn = {0.f, 0.f, 0.f, 0.f}, xn = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](int u, floatx4 &wv, floatx4 &xv) {
        const int tap = u / cpt, c0 = (u - tap * cpt) * 8;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
        wv = *reinterpret_cast<const floatx4 *>(wrow + (long)tap * a.Cin + c0);
        const int iy = oy * a.stride - a.pad + kh, ix = ox * a.stride - a.pad + kw;
        const bool ok = mok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        xv = floatx4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            xv = *reinterpret_cast<const floatx4 *>(xf + ((long)iy * a.W + ix) * a.Cin + c0);
            if (a.ps) {  // leading BatchNorm on in-image pixels only: the conv's zero padding pads the NORMALISED tensor
                const floatx4 s = *reinterpret_cast<const floatx4 *>(a.ps + c0 + 4 * hi), b = *reinterpret_cast<const floatx4 *>(a.pb + c0 + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = xv[e] * s[e] + b[e];
            }
        }
    };
    int u = wave;
    if (u < U) load(u, wn, xn);
    for (; u < U; u += 4) {
        wa = wn;
        xb = xn;
        if (u + 4 < U) load(u + 4, wn, xn);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[e], xb[e], acc, 0, 0, 0);
    }
    // partial tile of this wave: acc[e] = (cout (e & 3) + 8 (e >> 2) + 4 hi, pixel r)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * hi][r] = acc[e];
    __syncthreads();
    // thread t: pixel t & 31, channels 4 (t >> 5) .. + 3; the four partials are added in wave order (deterministic)
    const int px = tid & 31, cg = tid >> 5;
    const long mo = m0 + px;
    if (mo >= M) return;
    const int fo = (int)(mo / (a.Ho * a.Wo)), po = (int)(mo - (long)fo * a.Ho * a.Wo), yo = po / a.Wo, xo = po - yo * a.Wo;
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
    *reinterpret_cast<floatx4 *>(a.out + mo * a.Cout + co0 + 4 * cg) = o;
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
    hipLaunchKernelGGL(conv32_kernel, dim3((unsigned)((M + 31) / 32), (unsigned)(c.Cout / 32)), dim3(256), 0, s, a);
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
