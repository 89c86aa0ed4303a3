// RetinaFace-mobilenet0.25 forward (fp32, NCHW) as hand-written bandwidth-oriented HIP kernels for gfx950.
//
// Arithmetic spec: /root/reference/conversion/retina/models/net.py:9-38 (conv_bn / conv_dw blocks), :40-66 (SSH), :68-98
// (FPN), :102-124 (MobileNetV1 stages) and retinaface_trim.py:14-35,107-127 (heads, concat, softmax).  In the reference
// this network runs inside a TensorRT engine (src/retinaface.cpp:141).
//
// The whole detector is ~35 flop/byte - far below the machine balance - so no matrix cores (north_star agrees); what matters
// is (i) every activation is read and written once, (ii) loads/stores are coalesced, (iii) weights never cost HBM traffic.
//   * layout is planar NCHW with one thread per output pixel: for every input channel the 64 lanes of a wave read 64
//     consecutive pixels (256 contiguous bytes), and write 64 consecutive pixels per output channel;
//   * weights are indexed by wave-uniform values only, so the compiler keeps them on the scalar path (s_load -> SGPR
//     operand of v_fma_f32): zero vector-memory traffic and zero VGPRs for weights;
//   * BatchNorm is folded into the preceding conv on the host (all detector BNs follow their conv); ReLU, the FPN
//     nearest-upsample+add, the SSH concat (+ its ReLU) and the head permute/softmax are fused into the producing kernel;
//   * a conv_dw block (depthwise 3x3 + BN + ReLU + pointwise 1x1 + BN + ReLU) is ONE kernel: the depthwise result lives
//     in a register and feeds CT pointwise accumulators, the intermediate tensor never exists.
#include "frt_kernels.h"

namespace {

// ---------------------------------------------------------------- fused depthwise3x3(+ReLU) -> pointwise1x1 (+ReLU) (+upsample-add)
template <int CT, bool HAS_DW>
__global__ __launch_bounds__(256) void dwpw_kernel(DwPwArgs a) {
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    const int HoWo = a.Ho * a.Wo;
    if (gp >= (long)a.B * HoWo) return;
    const int b = (int)(gp / HoWo), p = (int)(gp - (long)b * HoWo);
    const int oh = p / a.Wo, ow = p - oh * a.Wo;
    const int co0 = blockIdx.y * CT;
    const int HW = a.H * a.W;

    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;

    const float *inb = a.in + (long)b * a.Cin * HW;
    if (HAS_DW) {
        const int ih0 = oh * a.stride - 1, iw0 = ow * a.stride - 1;
        int off[9];
        bool ok[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = ih0 + kh, iw = iw0 + kw;
                ok[kh * 3 + kw] = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                off[kh * 3 + kw] = ih * a.W + iw;
            }
        for (int ci = 0; ci < a.Cin; ++ci) {
            const float *x = inb + (long)ci * HW;
            const float *wd = a.wd + ci * 9;
            float d = a.bd[ci];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = ok[t] ? x[off[t]] : 0.f;
                d = fmaf(v, wd[t], d);
            }
            d = fmaxf(d, 0.f);
            const float *wp = a.wp + (long)ci * a.Cout + co0;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = fmaf(d, wp[c], acc[c]);
        }
    } else {
        const int ip = (oh * a.stride) * a.W + ow * a.stride;
        for (int ci = 0; ci < a.Cin; ++ci) {
            const float d = inb[(long)ci * HW + ip];
            const float *wp = a.wp + (long)ci * a.Cout + co0;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = fmaf(d, wp[c], acc[c]);
        }
    }

    float *ob = a.out + ((long)b * a.Cout + co0) * HoWo + p;
    const float *addb = nullptr;
    if (a.add) {
        // F.interpolate(mode="nearest") to (Ho,Wo): src = min(floor(dst * (float)in/out), in-1)   (net.py:89,93)
        const float sh = (float)a.add_h / (float)a.Ho, sw = (float)a.add_w / (float)a.Wo;
        int ah = (int)floorf(oh * sh), aw = (int)floorf(ow * sw);
        ah = ah < a.add_h - 1 ? ah : a.add_h - 1;
        aw = aw < a.add_w - 1 ? aw : a.add_w - 1;
        addb = a.add + ((long)b * a.Cout + co0) * a.add_h * a.add_w + ah * a.add_w + aw;
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float v = acc[c] + a.bp[co0 + c];
        if (a.relu) v = fmaxf(v, 0.f);
        if (addb) v += addb[(long)c * a.add_h * a.add_w];
        ob[(long)c * HoWo] = v;
    }
}

// ---------------------------------------------------------------- dense 3x3 (pad 1) + bias (+ReLU), writes a channel slice
template <int CT>
__global__ __launch_bounds__(256) void conv3x3_kernel(Conv3Args a) {
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    const int HoWo = a.Ho * a.Wo;
    if (gp >= (long)a.B * HoWo) return;
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
            off[kh * 3 + kw] = ih * a.W + iw;
        }
    const float *inb = a.in + (long)b * a.Cin * HW;
    for (int ci = 0; ci < a.Cin; ++ci) {
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
    float *ob = a.out + ((long)b * a.out_ctotal + a.out_coff + co0) * HoWo + p;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float v = acc[c] + a.b[co0 + c];
        if (a.relu) v = fmaxf(v, 0.f);
        ob[(long)c * HoWo] = v;
    }
}

// ---------------------------------------------------------------- heads: 1x1 64->8 (bbox) and 64->4 (class) + softmax, NHWC order
__global__ __launch_bounds__(256) void heads_kernel(HeadArgs a) {
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
    for (int ci = 0; ci < a.C; ++ci) {
        const float x = inb[(long)ci * HW];
#pragma unroll
        for (int c = 0; c < 8; ++c) lb[c] = fmaf(x, a.wb[ci * 8 + c], lb[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) lc[c] = fmaf(x, a.wc[ci * 4 + c], lc[c]);
    }
    // anchor index = base + (i*W + j)*2 + l ; channels 4l..4l+3 / 2l..2l+1 belong to anchor l (retinaface_trim.py:22-24,33-35)
    const long an = (long)b * a.A + a.base + (long)p * 2;
    floatx4 o0 = {lb[0] + a.bb[0], lb[1] + a.bb[1], lb[2] + a.bb[2], lb[3] + a.bb[3]};
    floatx4 o1 = {lb[4] + a.bb[4], lb[5] + a.bb[5], lb[6] + a.bb[6], lb[7] + a.bb[7]};
    *reinterpret_cast<floatx4 *>(a.loc + an * 4) = o0;
    *reinterpret_cast<floatx4 *>(a.loc + an * 4 + 4) = o1;
    floatx4 cf;
#pragma unroll
    for (int l = 0; l < 2; ++l) {  // F.softmax(dim=-1) over the 2 classes (retinaface_trim.py:126): max-subtracted exp / sum
        const float c0 = lc[2 * l] + a.bc[2 * l], c1 = lc[2 * l + 1] + a.bc[2 * l + 1];
        const float m = fmaxf(c0, c1);
        const float e0 = expf(c0 - m), e1 = expf(c1 - m);
        const float s = e0 + e1;
        cf[2 * l] = e0 / s;
        cf[2 * l + 1] = e1 / s;
    }
    *reinterpret_cast<floatx4 *>(a.conf + an * 2) = cf;
}

template <int CT>
void launch_dwpw_t(const DwPwArgs &a, hipStream_t s) {
    const long total = (long)a.B * a.Ho * a.Wo;
    dim3 grid((unsigned)((total + 255) / 256), a.Cout / CT);
    if (a.wd)
        hipLaunchKernelGGL((dwpw_kernel<CT, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((dwpw_kernel<CT, false>), grid, dim3(256), 0, s, a);
}

}  // namespace

void launch_dwpw(const DwPwArgs &a, hipStream_t s) {
    // output-channel tile per thread: large enough to amortise the depthwise work, small enough to keep >= ~1000 waves in flight
    const long total = (long)a.B * a.Ho * a.Wo;
    if (a.Cout % 64 == 0 && total >= 256L * 1024)
        launch_dwpw_t<64>(a, s);
    else if (a.Cout % 32 == 0 && total >= 256L * 64)
        launch_dwpw_t<32>(a, s);
    else if (a.Cout % 16 == 0)
        launch_dwpw_t<16>(a, s);
    else
        launch_dwpw_t<8>(a, s);
}

void launch_conv3x3(const Conv3Args &a, hipStream_t s) {
    const long total = (long)a.B * a.Ho * a.Wo;
    const unsigned gx = (unsigned)((total + 255) / 256);
    if (a.Cout % 32 == 0 && total >= 256L * 256)
        hipLaunchKernelGGL((conv3x3_kernel<32>), dim3(gx, a.Cout / 32), dim3(256), 0, s, a);
    else if (a.Cout % 16 == 0)
        hipLaunchKernelGGL((conv3x3_kernel<16>), dim3(gx, a.Cout / 16), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv3x3_kernel<8>), dim3(gx, a.Cout / 8), dim3(256), 0, s, a);
}

void launch_heads(const HeadArgs &a, hipStream_t s) {
    const long total = (long)a.B * a.H * a.W;
    hipLaunchKernelGGL(heads_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}
