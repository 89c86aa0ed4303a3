// Baseline JPEG, device half (frame ingest / reply step; see frt_jpeg.hpp): everything that is per block or per pixel.
//
//   decode:  int16 coefficient blocks (natural order, as the host's Huffman decoder left them)
//            -> dequantise + 8x8 inverse DCT ("islow": the Loeffler-Ligtenberg-Moschytz integer transform with 13-bit constants,
//               two passes, the same scaling and rounding points as the IJG decoder OpenCV's cv::imdecode sits on)
//            -> component planes (u8) -> "fancy" triangle-filter chroma upsampling (h2v1 / h2v2) + YCbCr -> BGR -> u8 HWC frame.
//   encode:  u8 BGR crop -> YCbCr (fixed point, 16 fraction bits) -> 2x2 chroma box filter with the alternating 1,2 rounding bias
//            -> forward DCT (same transform family) -> quantisation (round half up on the magnitude) -> zigzag-ordered int16 blocks
//            for the host's Huffman encoder.
// All of it is integer arithmetic with fixed rounding points, so the bytes are DEFINED: tests compare them with PIL's libjpeg-turbo
// (the library behind cv::imdecode / cv::imencode in the reference, src/app.cpp:296,328).  HBM-bound, trivial next to the networks.
#include "frt_jpeg_dev.h"

namespace {

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373, F_1_175875602 = 9633,
              F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one 8-point inverse transform; in[] = the (dequantised) inputs 0..7, shift = descale amount, out written through `st`
template <typename Store>
__device__ __forceinline__ void idct8(const int *in, int shift, Store st) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * F_0_541196100;
    int tmp2 = z1 + z3 * (-F_1_847759065);
    int tmp3 = z1 + z2 * F_0_765366865;
    z2 = in[0];
    z3 = in[4];
    int tmp0 = (z2 + z3) << CONST_BITS;
    int tmp1 = (z2 - z3) << CONST_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336;
    tmp1 *= F_2_053119869;
    tmp2 *= F_3_072711026;
    tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223;
    z2 *= -F_2_562915447;
    z3 *= -F_1_961570560;
    z4 *= -F_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    st(0, descale(tmp10 + tmp3, shift));
    st(7, descale(tmp10 - tmp3, shift));
    st(1, descale(tmp11 + tmp2, shift));
    st(6, descale(tmp11 - tmp2, shift));
    st(2, descale(tmp12 + tmp1, shift));
    st(5, descale(tmp12 - tmp1, shift));
    st(3, descale(tmp13 + tmp0, shift));
    st(4, descale(tmp13 - tmp0, shift));
}

// thread = one 8x8 block.  grid.y = image * 3 + component.
__global__ __launch_bounds__(128) void jpeg_idct_kernel(const int16_t *__restrict__ coef, const JpegImageDesc *__restrict__ desc, uint8_t *__restrict__ planes) {
    const int img = blockIdx.y / 3, comp = blockIdx.y % 3;
    const JpegImageDesc &d = desc[img];
    if (comp >= d.ncomp) return;
    const int nb = d.bw[comp] * d.bh[comp];
    const int b = blockIdx.x * 128 + threadIdx.x;
    if (b >= nb) return;
    const int by = b / d.bw[comp], bx = b - by * d.bw[comp];
    const int16_t *c = coef + (d.coef_block0 + d.block0[comp] + (size_t)b) * 64;
    const uint16_t *q = d.q[comp];
    int ws[64];
    // pass 1: columns of the coefficient block (loaded row by row, 16 bytes at a time)
    short v[64];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int4 t = *reinterpret_cast<const int4 *>(c + r * 8);
        const short *ts = reinterpret_cast<const short *>(&t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[r * 8 + k] = ts[k];
    }
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int in[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)v[r * 8 + col] * (int)q[r * 8 + col];
        idct8(in, CONST_BITS - PASS1_BITS, [&](int r, int val) { ws[r * 8 + col] = val; });
    }
    // pass 2: rows; +128, clamp, 8 bytes per row
    uint8_t *out = planes + d.plane_off[comp] + ((size_t)by * 8) * d.pitch[comp] + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int in[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) in[k] = ws[r * 8 + k];
        uint8_t o[8];
        idct8(in, CONST_BITS + PASS1_BITS + 3, [&](int k, int val) { o[k] = (uint8_t)clamp255(val + 128); });
        *reinterpret_cast<uint2 *>(out + (size_t)r * d.pitch[comp]) = *reinterpret_cast<const uint2 *>(o);
    }
}

// chroma sample at full-resolution position (x, y) after "fancy" upsampling of plane p (dw x dh samples, pitch bytes per row)
__device__ __forceinline__ int chroma_at(const uint8_t *__restrict__ p, int pitch, int dw, int dh, int hs, int vs, int x, int y) {
    if (hs == 1 && vs == 1) return p[(size_t)y * pitch + x];
    const int c = x >> 1;
    if (dw <= 2) return p[(size_t)(vs == 2 ? (y >> 1) : y) * pitch + c];  // tiny images: plain replication (the fancy kernels need > 2 columns)
    if (vs == 1) {  // h2v1: 3/4 nearer + 1/4 farther, biases 1 (left-leaning) and 2 (right-leaning)
        const uint8_t *r = p + (size_t)y * pitch;
        if (x & 1) return c == dw - 1 ? r[c] : (3 * r[c] + r[c + 1] + 2) >> 2;
        return c == 0 ? r[c] : (3 * r[c] + r[c - 1] + 1) >> 2;
    }
    // h2v2: vertical 3:1 blend of the nearer / farther chroma row first (kept at x4 scale), then the horizontal 3:1 blend, biases 8 / 7
    const int i = y >> 1;
    int j = (y & 1) ? i + 1 : i - 1;
    j = j < 0 ? 0 : (j > dh - 1 ? dh - 1 : j);
    const uint8_t *r0 = p + (size_t)i * pitch, *r1 = p + (size_t)j * pitch;
    const int cur = 3 * r0[c] + r1[c];
    if (x & 1) {
        if (c == dw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + 3 * r0[c + 1] + r1[c + 1] + 7) >> 4;
    }
    if (c == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + 3 * r0[c - 1] + r1[c - 1] + 8) >> 4;
}

// thread = one output pixel; grid.z = image.  Writes tight u8 BGR rows into the image's slot of `out`.
__global__ __launch_bounds__(256) void jpeg_color_kernel(const uint8_t *__restrict__ planes, const JpegImageDesc *__restrict__ desc, uint8_t *__restrict__ out) {
    const JpegImageDesc &d = desc[blockIdx.z];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= d.width || y >= d.height) return;
    const int Y = planes[d.plane_off[0] + (size_t)y * d.pitch[0] + x];
    int r, g, b;
    if (d.ncomp == 1) {
        r = g = b = Y;
    } else {
        const int hs = d.hmax / d.h[1], vs = d.vmax / d.v[1];
        const int cb = chroma_at(planes + d.plane_off[1], d.pitch[1], d.dw[1], d.dh[1], hs, vs, x, y) - 128;
        const int cr = chroma_at(planes + d.plane_off[2], d.pitch[2], d.dw[2], d.dh[2], hs, vs, x, y) - 128;
        // 16-bit fixed point: 1.40200 -> 91881, 1.77200 -> 116130, 0.71414 -> 46802, 0.34414 -> 22554; half added before the shift
        r = clamp255(Y + ((91881 * cr + 32768) >> 16));
        b = clamp255(Y + ((116130 * cb + 32768) >> 16));
        g = clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    }
    uint8_t *o = out + d.out_off + ((size_t)y * d.width + x) * 3;
    o[0] = (uint8_t)b;
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)r;
}

// ---------------------------------------------------------------------------------------------------------------- encoder
template <typename Store>
__device__ __forceinline__ void fdct8(const int *d, bool pass1, Store st) {
    const int tmp0 = d[0] + d[7], tmp7 = d[0] - d[7], tmp1 = d[1] + d[6], tmp6 = d[1] - d[6];
    const int tmp2 = d[2] + d[5], tmp5 = d[2] - d[5], tmp3 = d[3] + d[4], tmp4 = d[3] - d[4];
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    const int sh = pass1 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
    if (pass1) {
        st(0, (tmp10 + tmp11) << PASS1_BITS);
        st(4, (tmp10 - tmp11) << PASS1_BITS);
    } else {
        st(0, descale(tmp10 + tmp11, PASS1_BITS));
        st(4, descale(tmp10 - tmp11, PASS1_BITS));
    }
    int z1 = (tmp12 + tmp13) * F_0_541196100;
    st(2, descale(z1 + tmp13 * F_0_765366865, sh));
    st(6, descale(z1 + tmp12 * (-F_1_847759065), sh));
    z1 = tmp4 + tmp7;
    int z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
    const int z5 = (z3 + z4) * F_1_175875602;
    const int t4 = tmp4 * F_0_298631336, t5 = tmp5 * F_2_053119869, t6 = tmp6 * F_3_072711026, t7 = tmp7 * F_1_501321110;
    z1 *= -F_0_899976223;
    z2 *= -F_2_562915447;
    z3 *= -F_1_961570560;
    z4 *= -F_0_390180644;
    z3 += z5;
    z4 += z5;
    st(7, descale(t4 + z1 + z3, sh));
    st(5, descale(t5 + z2 + z4, sh));
    st(3, descale(t6 + z2 + z3, sh));
    st(1, descale(t7 + z1 + z4, sh));
}

__device__ __forceinline__ void bgr_to_ycc(const uint8_t *p, int &y, int &cb, int &cr) {
    const int b = p[0], g = p[1], r = p[2];
    // 16-bit fixed point: 0.29900 19595, 0.58700 38470, 0.11400 7471 | 0.16874 11059, 0.33126 21709, 0.5 32768 | 0.41869 27439, 0.08131 5329
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

// thread = one 8x8 block of one component of one image (4:2:0).  grid.y = image; blocks of an image: Y [2*mcuy][2*mcux], Cb, Cr [mcuy][mcux].
// q: [2][64] natural order (luma, chroma).  out: zigzag-ordered int16 [n][blocks][64].
__global__ __launch_bounds__(64) void jpeg_encode_blocks_kernel(const uint8_t *__restrict__ bgr, int n, int rows, int cols, const uint16_t *__restrict__ q,
                                                                int16_t *__restrict__ out) {
    const int mcux = (cols + 15) >> 4, mcuy = (rows + 15) >> 4;
    const int yb = 4 * mcux * mcuy, cbk = mcux * mcuy, per_img = yb + 2 * cbk;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= per_img) return;
    const uint8_t *img = bgr + (size_t)blockIdx.y * rows * cols * 3;
    int comp, bx, by;
    if (b < yb) {
        comp = 0;
        by = b / (2 * mcux);
        bx = b - by * 2 * mcux;
    } else {
        comp = 1 + (b - yb) / cbk;
        const int k = (b - yb) % cbk;
        by = k / mcux;
        bx = k - by * mcux;
    }
    int ws[64];
    // samples (level-shifted) -> pass 1 over rows
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
        int d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int val;
            if (comp == 0) {
                const int yy = min(by * 8 + r, rows - 1), xx = min(bx * 8 + k, cols - 1);  // edge replication to whole MCUs
                int y, cb, cr;
                bgr_to_ycc(img + ((size_t)yy * cols + xx) * 3, y, cb, cr);
                val = y;
            } else {
                int sum = 0;
                // bottom padding of a chroma plane repeats its last real (downsampled) row; right padding repeats the last full-
                // resolution column before the box filter - the two edges are NOT symmetric in the IJG encoder
                const int cy = min(by * 8 + r, ((rows + 1) >> 1) - 1);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int yy = min(cy * 2 + dy, rows - 1), xx = min((bx * 8 + k) * 2 + dx, cols - 1);
                        int y, cb, cr;
                        bgr_to_ycc(img + ((size_t)yy * cols + xx) * 3, y, cb, cr);
                        sum += comp == 1 ? cb : cr;
                    }
                val = (sum + 1 + (k & 1)) >> 2;  // bias 1, 2, 1, 2, ... along the output row
            }
            d[k] = val - 128;
        }
        fdct8(d, true, [&](int k, int v) { ws[r * 8 + k] = v; });
    }
    const uint16_t *qt = q + (comp ? 64 : 0);
    int16_t *o = out + ((size_t)blockIdx.y * per_img + b) * 64;
    short res[64];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int d[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r] = ws[r * 8 + c];
        fdct8(d, false, [&](int r, int v) {
            // quantise: the transform output carries a factor 8; divide by 8*q with round-half-up on the magnitude
            const int qv = (int)qt[r * 8 + c] << 3;
            int t = v < 0 ? -v : v;
            t = (t + (qv >> 1)) / qv;
            res[r * 8 + c] = (short)(v < 0 ? -t : t);
        });
    }
    constexpr unsigned char zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
#pragma unroll
    for (int k = 0; k < 64; ++k) o[k] = res[zz[k]];
}

}  // namespace

void launch_jpeg_decode(const int16_t *coef, const JpegImageDesc *desc_dev, int n, int max_blocks_per_comp, int max_w, int max_h, uint8_t *planes,
                        uint8_t *out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks_per_comp + 127) / 128, n * 3), dim3(128), 0, s, coef, desc_dev, planes);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((max_w + 63) / 64, (max_h + 3) / 4, n), dim3(256), 0, s, planes, desc_dev, out);
}

void launch_jpeg_encode_blocks(const uint8_t *bgr, int n, int rows, int cols, const uint16_t *q_dev, int16_t *out, hipStream_t s) {
    if (n <= 0) return;
    const int mcux = (cols + 15) >> 4, mcuy = (rows + 15) >> 4;
    const int per_img = 6 * mcux * mcuy;
    hipLaunchKernelGGL(jpeg_encode_blocks_kernel, dim3((per_img + 63) / 64, n), dim3(64), 0, s, bgr, n, rows, cols, q_dev, out);
}
