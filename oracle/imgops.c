/*
 * ORACLE (test infrastructure only).  CPU restatement of the reference's OpenCV image operations on the hot path:
 *
 *   det letterbox + normalise   <- /root/reference/src/retinaface.cpp:106-136  (cv::resize INTER_LINEAR, 128-grey canvas,
 *                                  convertTo CV_32F, minus (104,117,123), split -> planar BGR)
 *   face crop + resize          <- /root/reference/src/arcface.cpp:3-17        (cv::Rect ROI, cv::resize INTER_CUBIC)
 *   face normalise              <- /root/reference/src/arcface.cpp:105-129     (BGR2RGB, (x-127.5)*0.0078125, split)
 *
 * PARITY UNPINNED: the arithmetic lives in OpenCV 4.5.5 (third party, un-vendored: /root/reference/README.md:11,
 * app/CMakeLists.txt:8) which is absent from this image (no C++ headers, no cv2), and the reference has no tests for it.
 * This file restates OpenCV's published 8-bit resize algorithm (modules/imgproc/src/resize.cpp, 4.5.x):
 *   - pixel-centre mapping  src = (dst + 0.5) * scale - 0.5, scale = src_size / dst_size (double), no antialiasing;
 *   - coefficients computed in float, converted to short with scale 2^11 by round-to-nearest-even (cvRound);
 *   - INTER_LINEAR 8U: horizontal pass in int, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2;
 *     x taps reset to (sx, fx=0) outside the source, y taps clamp the row index instead;
 *   - INTER_CUBIC 8U (a = -0.75): 4x4 taps, indices clamped to the ROI (a Mat ROI is its own image for cv::resize),
 *     result = saturate_u8((sum + 2^21) >> 22);
 *   - equal source/destination size degenerates to a copy.
 * tests/test_oracle_imgops.py cross-checks it against torch.nn.functional.interpolate (same half-pixel / a=-0.75
 * convention, float arithmetic): agreement within +-1 LSB is required, exactness vs OpenCV is NOT claimed.
 *
 * Build with -ffp-contract=off (coefficients must round the same way on every machine and in the HIP kernels).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define COEF_BITS 11
#define COEF_SCALE (1 << COEF_BITS)

static inline int floor_i(float v) {
    int i = (int)v;
    return i - (v < (float)i);
}

static inline short sat_short_round(float v) {
    long r = lrintf(v); /* round-half-even in the default rounding mode == cvRound */
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return (short)r;
}

static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ---------------------------------------------------------------- INTER_LINEAR, 8UC3 */
void orc_resize_linear_u8c3(const uint8_t *src, int sh, int sw, size_t sstride, uint8_t *dst, int dh, int dw, size_t dstride) {
    if (sh == dh && sw == dw) {
        for (int y = 0; y < dh; ++y) memcpy(dst + (size_t)y * dstride, src + (size_t)y * sstride, (size_t)dw * 3);
        return;
    }
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *xa = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
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
        xofs[dx] = sx;
        xa[2 * dx] = sat_short_round((1.f - fx) * COEF_SCALE);
        xa[2 * dx + 1] = sat_short_round(fx * COEF_SCALE);
    }
    int *row0 = (int *)malloc(sizeof(int) * 3 * (size_t)dw), *row1 = (int *)malloc(sizeof(int) * 3 * (size_t)dw);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = floor_i(fy);
        fy -= sy;
        short b0 = sat_short_round((1.f - fy) * COEF_SCALE), b1 = sat_short_round(fy * COEF_SCALE);
        int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const uint8_t *s0 = src + (size_t)y0 * sstride, *s1 = src + (size_t)y1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            for (int c = 0; c < 3; ++c) {
                row0[dx * 3 + c] = s0[sx * 3 + c] * xa[2 * dx] + s0[sx1 * 3 + c] * xa[2 * dx + 1];
                row1[dx * 3 + c] = s1[sx * 3 + c] * xa[2 * dx] + s1[sx1 * 3 + c] * xa[2 * dx + 1];
            }
        }
        uint8_t *d = dst + (size_t)dy * dstride;
        for (int i = 0; i < dw * 3; ++i) d[i] = (uint8_t)((((b0 * (row0[i] >> 4)) >> 16) + ((b1 * (row1[i] >> 4)) >> 16) + 2) >> 2);
    }
    free(row0);
    free(row1);
    free(xofs);
    free(xa);
}

/* ---------------------------------------------------------------- INTER_CUBIC, 8UC3 */
static inline void cubic_coeffs(float x, float *c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

void orc_resize_cubic_u8c3(const uint8_t *src, int sh, int sw, size_t sstride, uint8_t *dst, int dh, int dw, size_t dstride) {
    if (sh == dh && sw == dw) {
        for (int y = 0; y < dh; ++y) memcpy(dst + (size_t)y * dstride, src + (size_t)y * sstride, (size_t)dw * 3);
        return;
    }
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *xa = (short *)malloc(sizeof(short) * 4 * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = floor_i(fx);
        fx -= sx;
        float c[4];
        cubic_coeffs(fx, c);
        xofs[dx] = sx;
        for (int k = 0; k < 4; ++k) xa[4 * dx + k] = sat_short_round(c[k] * COEF_SCALE);
    }
    int *rows = (int *)malloc(sizeof(int) * 4 * 3 * (size_t)dw);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = floor_i(fy);
        fy -= sy;
        float c[4];
        cubic_coeffs(fy, c);
        short b[4];
        for (int k = 0; k < 4; ++k) b[k] = sat_short_round(c[k] * COEF_SCALE);
        for (int k = 0; k < 4; ++k) {
            int yy = sy - 1 + k;
            yy = yy < 0 ? 0 : (yy > sh - 1 ? sh - 1 : yy);
            const uint8_t *s = src + (size_t)yy * sstride;
            int *r = rows + (size_t)k * 3 * dw;
            for (int dx = 0; dx < dw; ++dx) {
                int xi[4];
                for (int t = 0; t < 4; ++t) {
                    int xx = xofs[dx] - 1 + t;
                    xi[t] = xx < 0 ? 0 : (xx > sw - 1 ? sw - 1 : xx);
                }
                for (int ch = 0; ch < 3; ++ch)
                    r[dx * 3 + ch] = s[xi[0] * 3 + ch] * xa[4 * dx] + s[xi[1] * 3 + ch] * xa[4 * dx + 1] + s[xi[2] * 3 + ch] * xa[4 * dx + 2] +
                                     s[xi[3] * 3 + ch] * xa[4 * dx + 3];
            }
        }
        uint8_t *d = dst + (size_t)dy * dstride;
        for (int i = 0; i < dw * 3; ++i) {
            int v = rows[i] * b[0] + rows[3 * dw + i] * b[1] + rows[6 * dw + i] * b[2] + rows[9 * dw + i] * b[3];
            d[i] = sat_u8((v + (1 << 21)) >> 22);
        }
    }
    free(rows);
    free(xofs);
    free(xa);
}

/* ---------------------------------------------------------------- detector pre-processing, retinaface.cpp:106-136 */
/* frame: u8 BGR HWC frame_h x frame_w (row stride in bytes); out: float32 [3][in_h][in_w] planar BGR, mean-subtracted. */
void orc_det_preprocess(const uint8_t *frame, int frame_h, int frame_w, size_t stride, int in_h, int in_w, float *out) {
    float scale_h = (float)in_h / frame_h, scale_w = (float)in_w / frame_w;
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
    uint8_t *canvas = (uint8_t *)malloc((size_t)in_h * in_w * 3);
    memset(canvas, 128, (size_t)in_h * in_w * 3);
    orc_resize_linear_u8c3(frame, frame_h, frame_w, stride, canvas + ((size_t)y * in_w + x) * 3, h, w, (size_t)in_w * 3);
    const float mean[3] = {104.f, 117.f, 123.f};
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < in_h * in_w; ++i) out[(size_t)c * in_h * in_w + i] = (float)canvas[(size_t)i * 3 + c] - mean[c];
    free(canvas);
}

/* ---------------------------------------------------------------- getCroppedFaces, arcface.cpp:3-17 */
/* One box.  ROI = cols [y1,y2) x rows [x1,x2)  (cv::Rect(Point(y1,x1), Point(y2,x2)) excludes the far corner).
 * Returns 0, or -1 for an empty ROI (OpenCV would throw).  crop: u8 BGR [out_h][out_w][3]. */
int orc_crop_face(const uint8_t *frame, int frame_h, int frame_w, size_t stride, int x1, int y1, int x2, int y2, int out_h, int out_w,
                  uint8_t *crop) {
    int rh = x2 - x1, rw = y2 - y1;
    if (rh <= 0 || rw <= 0 || x1 < 0 || y1 < 0 || x2 > frame_h || y2 > frame_w) return -1;
    orc_resize_cubic_u8c3(frame + (size_t)x1 * stride + (size_t)y1 * 3, rh, rw, stride, crop, out_h, out_w, (size_t)out_w * 3);
    return 0;
}

/* ---------------------------------------------------------------- preprocessFace(s), arcface.cpp:105-129 */
/* crop u8 BGR [h][w][3] -> float32 planar RGB [3][h][w], (x - 127.5) * 0.0078125 */
void orc_face_normalize(const uint8_t *crop, int h, int w, float *out) {
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < h * w; ++i) out[(size_t)c * h * w + i] = ((float)crop[(size_t)i * 3 + (2 - c)] - 127.5f) * 0.0078125f;
}
