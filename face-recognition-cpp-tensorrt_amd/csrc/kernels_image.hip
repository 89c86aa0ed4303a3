// Image pre-processing on the device (compiled with -ffp-contract=off: coefficient rounding must match oracle/imgops.c).
//
//   det_preprocess_kernel  <- RetinaFace::preprocess   (/root/reference/src/retinaface.cpp:106-136): letterbox with
//                             cv::resize INTER_LINEAR onto a 128-grey canvas, float conversion, minus (104,117,123) in
//                             BGR order, planar CHW.  One fused pass: u8 HWC in, fp32 planar out (the reference makes
//                             ~5 host passes and then uploads 4x the bytes).
//   crop_faces_kernel      <- getCroppedFaces          (/root/reference/src/arcface.cpp:3-17): ROI = cols [y1,y2) x
//                             rows [x1,x2), cv::resize INTER_CUBIC to 112x112, fused with preprocessFaces
//                             (arcface.cpp:116-129): BGR->RGB, (x-127.5)*0.0078125, planar CHW.
//
// The 8-bit resize arithmetic restates OpenCV 4.5's fixed-point path (11-bit coefficients, see oracle/imgops.c for the
// full statement and the "parity unpinned" note: OpenCV itself is not available in this image).
#include "frt_kernels.h"

namespace {

constexpr int COEF_SCALE = 2048;

__device__ __forceinline__ int floor_i(float v) {
    const int i = (int)v;
    return i - (v < (float)i);
}
__device__ __forceinline__ int sat_short_round(float v) {
    int r = (int)rintf(v);  // round-half-even == cvRound
    r = r > 32767 ? 32767 : r;
    r = r < -32768 ? -32768 : r;
    return r;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void det_preprocess_kernel(const uint8_t *__restrict__ frames, int frame_h, int frame_w, size_t row_stride, size_t frame_stride,
                                      int in_h, int in_w, int rw, int rh, int rx, int ry, float *__restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= in_h * in_w) return;
    const int oy = p / in_w, ox = p - oy * in_w;
    int v[3] = {128, 128, 128};
    const int dx = ox - rx, dy = oy - ry;
    if (dx >= 0 && dx < rw && dy >= 0 && dy < rh) {
        const uint8_t *src = frames + (size_t)f * frame_stride;
        if (rw == frame_w && rh == frame_h) {
            const uint8_t *s = src + (size_t)dy * row_stride + (size_t)dx * 3;
            v[0] = s[0];
            v[1] = s[1];
            v[2] = s[2];
        } else {
            const double scale_x = (double)frame_w / rw, scale_y = (double)frame_h / rh;
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = floor_i(fx);
            fx -= sx;
            if (sx < 0) {
                fx = 0;
                sx = 0;
            }
            if (sx >= frame_w - 1) {
                fx = 0;
                sx = frame_w - 1;
            }
            const int a0 = sat_short_round((1.f - fx) * COEF_SCALE), a1 = sat_short_round(fx * COEF_SCALE);
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            const int sy = floor_i(fy);
            fy -= sy;
            const int b0 = sat_short_round((1.f - fy) * COEF_SCALE), b1 = sat_short_round(fy * COEF_SCALE);
            const int y0 = clampi(sy, 0, frame_h - 1), y1 = clampi(sy + 1, 0, frame_h - 1);
            const int sx1 = sx + 1 < frame_w ? sx + 1 : frame_w - 1;
            const uint8_t *s0 = src + (size_t)y0 * row_stride, *s1 = src + (size_t)y1 * row_stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r0 = s0[sx * 3 + c] * a0 + s0[sx1 * 3 + c] * a1;
                const int r1 = s1[sx * 3 + c] * a0 + s1[sx1 * 3 + c] * a1;
                v[c] = ((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2) & 255;
            }
        }
    }
    const float mean[3] = {104.f, 117.f, 123.f};
    float *o = out + (size_t)f * 3 * in_h * in_w + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[(size_t)c * in_h * in_w] = (float)v[c] - mean[c];
}

// ---------------------------------------------------------------- frame ingest: cv::resize(img, img, Size(frameW, frameH)), app.cpp:301
// (default INTER_LINEAR, 8UC3).  Same fixed-point arithmetic as the letterbox resize above; the exact-2x case OpenCV redirects to
// its INTER_AREA fast path gives identical values ((a+b+c+d+2)>>2), so one code path covers it.
__global__ void resize_linear_u8_kernel(const uint8_t *__restrict__ src, int sh, int sw, size_t sstride, size_t sframe, uint8_t *__restrict__ dst,
                                        int dh, int dw, size_t dstride, size_t dframe) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= dh * dw) return;
    const int dy = p / dw, dx = p - dy * dw;
    const uint8_t *s = src + (size_t)f * sframe;
    uint8_t *o = dst + (size_t)f * dframe + (size_t)dy * dstride + (size_t)dx * 3;
    if (sh == dh && sw == dw) {
        const uint8_t *q = s + (size_t)dy * sstride + (size_t)dx * 3;
        o[0] = q[0];
        o[1] = q[1];
        o[2] = q[2];
        return;
    }
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = floor_i(fx);
    fx -= sx;
    if (sx < 0) {
        fx = 0;
        sx = 0;
    }
    if (sx >= sw - 1) {
        fx = 0;
        sx = sw - 1;
    }
    const int a0 = sat_short_round((1.f - fx) * COEF_SCALE), a1 = sat_short_round(fx * COEF_SCALE);
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = floor_i(fy);
    fy -= sy;
    const int b0 = sat_short_round((1.f - fy) * COEF_SCALE), b1 = sat_short_round(fy * COEF_SCALE);
    const int y0 = clampi(sy, 0, sh - 1), y1 = clampi(sy + 1, 0, sh - 1);
    const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
    const uint8_t *s0 = s + (size_t)y0 * sstride, *s1 = s + (size_t)y1 * sstride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int r0 = s0[sx * 3 + c] * a0 + s0[sx1 * 3 + c] * a1;
        const int r1 = s1[sx * 3 + c] * a0 + s1[sx1 * 3 + c] * a1;
        o[c] = (uint8_t)(((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2) & 255);
    }
}

__device__ __forceinline__ void cubic_coeffs(float x, int *c) {
    const float A = -0.75f;
    float w[4];
    w[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    w[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    w[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    w[3] = 1.f - w[0] - w[1] - w[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = sat_short_round(w[k] * COEF_SCALE);
}

__global__ void crop_faces_kernel(const uint8_t *__restrict__ frames, int frame_h, int frame_w, size_t row_stride, size_t frame_stride,
                                  const frt_bbox *__restrict__ boxes, const int *__restrict__ n_boxes, int max_faces, int frames_shared,
                                  int oh, int ow, uint8_t *__restrict__ crops, float *__restrict__ chw, int *__restrict__ valid) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= oh * ow) return;
    const int dy = p / ow, dx = p - dy * ow;
    const int frame = frames_shared ? 0 : f / max_faces;
    const frt_bbox b = boxes[f];
    const int rh = b.x2 - b.x1, rw = b.y2 - b.y1;  // rows = x, cols = y; far corner excluded (cv::Rect(Point,Point))
    bool ok = rh > 0 && rw > 0 && b.x1 >= 0 && b.y1 >= 0 && b.x2 <= frame_h && b.y2 <= frame_w;
    if (n_boxes) ok = ok && (f % max_faces) < n_boxes[f / max_faces];
    if (p == 0 && valid) valid[f] = ok ? 1 : 0;
    int v[3] = {0, 0, 0};
    if (ok) {
        const uint8_t *src = frames + (size_t)frame * frame_stride + (size_t)b.x1 * row_stride + (size_t)b.y1 * 3;
        if (rh == oh && rw == ow) {
            const uint8_t *s = src + (size_t)dy * row_stride + (size_t)dx * 3;
            v[0] = s[0];
            v[1] = s[1];
            v[2] = s[2];
        } else {
            const double scale_x = (double)rw / ow, scale_y = (double)rh / oh;
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            const int sx = floor_i(fx);
            fx -= sx;
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            const int sy = floor_i(fy);
            fy -= sy;
            int ca[4], cb[4], xi[4];
            cubic_coeffs(fx, ca);
            cubic_coeffs(fy, cb);
#pragma unroll
            for (int t = 0; t < 4; ++t) xi[t] = clampi(sx - 1 + t, 0, rw - 1) * 3;
            int acc[3] = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = clampi(sy - 1 + k, 0, rh - 1);
                const uint8_t *s = src + (size_t)yy * row_stride;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h = s[xi[0] + c] * ca[0] + s[xi[1] + c] * ca[1] + s[xi[2] + c] * ca[2] + s[xi[3] + c] * ca[3];
                    acc[c] += h * cb[k];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = clampi((acc[c] + (1 << 21)) >> 22, 0, 255);
        }
    }
    if (crops) {
        uint8_t *o = crops + ((size_t)f * oh * ow + p) * 3;
        o[0] = (uint8_t)v[0];
        o[1] = (uint8_t)v[1];
        o[2] = (uint8_t)v[2];
    }
    float *o = chw + (size_t)f * 3 * oh * ow + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[(size_t)c * oh * ow] = ok ? ((float)v[2 - c] - 127.5f) * 0.0078125f : 0.f;
}

__global__ void face_normalize_kernel(const uint8_t *__restrict__ crops, int hw, float *__restrict__ chw) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= hw) return;
    const uint8_t *s = crops + ((size_t)f * hw + p) * 3;
    float *o = chw + (size_t)f * 3 * hw + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[(size_t)c * hw] = ((float)s[2 - c] - 127.5f) * 0.0078125f;
}

}  // namespace

void launch_det_preprocess(const uint8_t *frames, int n, int frame_h, int frame_w, size_t row_stride, size_t frame_stride, int in_h,
                           int in_w, float *out, hipStream_t s) {
    // letterbox geometry, retinaface.cpp:111-122 (float scales, int truncation)
    const float scale_h = (float)in_h / frame_h, scale_w = (float)in_w / frame_w;
    int w, h, x, y;
    if (scale_h > scale_w) {
        w = in_w;
        h = (int)(scale_w * frame_h);
        x = 0;
        y = (in_h - h) / 2;
    } else {
        w = (int)(scale_h * frame_w);
        h = in_h;
        x = (in_w - w) / 2;
        y = 0;
    }
    dim3 grid((in_h * in_w + 255) / 256, n);
    hipLaunchKernelGGL(det_preprocess_kernel, grid, dim3(256), 0, s, frames, frame_h, frame_w, row_stride, frame_stride, in_h, in_w, w, h, x,
                       y, out);
}

void launch_resize_linear(const uint8_t *src, int n, int sh, int sw, size_t sstride, size_t sframe, uint8_t *dst, int dh, int dw, size_t dstride,
                          size_t dframe, hipStream_t s) {
    if (n <= 0) return;
    dim3 grid((dh * dw + 255) / 256, n);
    hipLaunchKernelGGL(resize_linear_u8_kernel, grid, dim3(256), 0, s, src, sh, sw, sstride, sframe, dst, dh, dw, dstride, dframe);
}

void launch_crop_faces(const uint8_t *frames, int frame_h, int frame_w, size_t row_stride, size_t frame_stride, const frt_bbox *boxes,
                       const int *n_boxes, int max_faces, int F, int frames_shared, int oh, int ow, uint8_t *crops, float *chw, int *valid,
                       hipStream_t s) {
    if (F <= 0) return;
    dim3 grid((oh * ow + 255) / 256, F);
    hipLaunchKernelGGL(crop_faces_kernel, grid, dim3(256), 0, s, frames, frame_h, frame_w, row_stride, frame_stride, boxes, n_boxes, max_faces,
                       frames_shared, oh, ow, crops, chw, valid);
}

// ---------------------------------------------------------------- 5-point alignment (optional mode; absent from the reference)
// Least-squares similarity (scale * rotation + translation; equals Umeyama's solution whenever that is not a reflection) from the
// 5 detected landmarks to the public ArcFace 112x112 template, then an inverse-mapped bilinear warp with zero border, fused
// with the recogniser's BGR->RGB / (x-127.5)/128 / planar normalisation.  Float arithmetic; restated in oracle/align.py.
namespace {
__constant__ float c_arc_template[10] = {38.2946f, 51.6963f, 73.5318f, 51.5014f, 56.0252f, 71.7366f, 41.5493f, 92.3655f, 70.7299f, 92.2041f};

__global__ __launch_bounds__(256) void align_faces_kernel(const uint8_t *__restrict__ frames, int frame_h, int frame_w, size_t row_stride,
                                                          size_t frame_stride, const float *__restrict__ landmarks, const int *__restrict__ n_boxes,
                                                          int max_faces, int frames_shared, uint8_t *__restrict__ crops, float *__restrict__ chw,
                                                          int *__restrict__ valid) {
    const int f = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int frame = frames_shared ? 0 : f / max_faces;
    bool ok = !n_boxes || (f % max_faces) < n_boxes[f / max_faces];
    // every thread derives the same 2x3 inverse transform (25 flops; cheaper than a broadcast)
    const float *lm = landmarks + (long)f * 10;
    float msx = 0.f, msy = 0.f, mdx = 0.f, mdy = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        msx += lm[2 * k];
        msy += lm[2 * k + 1];
        mdx += c_arc_template[2 * k];
        mdy += c_arc_template[2 * k + 1];
    }
    msx *= 0.2f; msy *= 0.2f; mdx *= 0.2f; mdy *= 0.2f;
    float sa = 0.f, sb = 0.f, den = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float sx = lm[2 * k] - msx, sy = lm[2 * k + 1] - msy;
        const float dx = c_arc_template[2 * k] - mdx, dy = c_arc_template[2 * k + 1] - mdy;
        sa += sx * dx + sy * dy;
        sb += sx * dy - sy * dx;
        den += sx * sx + sy * sy;
    }
    ok = ok && den > 1e-6f && (sa * sa + sb * sb) > 1e-12f;
    if (p == 0 && valid) valid[f] = ok ? 1 : 0;
    if (p >= 112 * 112) return;
    int v[3] = {0, 0, 0};
    if (ok) {
        const float a = sa / den, b = sb / den;            // dst = [a -b; b a] * src + t
        const float tx = mdx - (a * msx - b * msy), ty = mdy - (b * msx + a * msy);
        const float n2 = a * a + b * b;
        const float ia = a / n2, ib = b / n2;               // src = [ia ib; -ib ia] * (dst - t)
        const int oy = p / 112, ox = p - oy * 112;
        const float ux = (float)ox - tx, uy = (float)oy - ty;
        const float sxf = ia * ux + ib * uy, syf = -ib * ux + ia * uy;
        const float fx0 = floorf(sxf), fy0 = floorf(syf);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx = sxf - fx0, wy = syf - fy0;
        const uint8_t *src = frames + (size_t)frame * frame_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
                t4[q] = (xx >= 0 && xx < frame_w && yy >= 0 && yy < frame_h) ? (float)src[(size_t)yy * row_stride + (size_t)xx * 3 + c] : 0.f;
            }
            const float top = t4[0] + wx * (t4[1] - t4[0]), bot = t4[2] + wx * (t4[3] - t4[2]);
            const float val = top + wy * (bot - top);
            v[c] = clampi((int)floorf(val + 0.5f), 0, 255);
        }
    }
    if (crops) {
        uint8_t *o = crops + ((size_t)f * 112 * 112 + p) * 3;
        o[0] = (uint8_t)v[0];
        o[1] = (uint8_t)v[1];
        o[2] = (uint8_t)v[2];
    }
    float *o = chw + (size_t)f * 3 * 112 * 112 + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[(size_t)c * 112 * 112] = ok ? ((float)v[2 - c] - 127.5f) * 0.0078125f : 0.f;
}
}  // namespace

void launch_align_faces(const uint8_t *frames, int frame_h, int frame_w, size_t row_stride, size_t frame_stride, const float *landmarks,
                        const int *n_boxes, int max_faces, int F, int frames_shared, uint8_t *crops, float *chw, int *valid, hipStream_t s) {
    if (F <= 0) return;
    dim3 grid((112 * 112 + 255) / 256, F);
    hipLaunchKernelGGL(align_faces_kernel, grid, dim3(256), 0, s, frames, frame_h, frame_w, row_stride, frame_stride, landmarks, n_boxes, max_faces,
                       frames_shared, crops, chw, valid);
}

void launch_face_normalize(const uint8_t *crops, int F, int oh, int ow, float *chw, hipStream_t s) {
    if (F <= 0) return;
    dim3 grid((oh * ow + 255) / 256, F);
    hipLaunchKernelGGL(face_normalize_kernel, grid, dim3(256), 0, s, crops, oh * ow, chw);
}
