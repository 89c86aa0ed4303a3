/*
 * ORACLE (test infrastructure only - never linked into libfrt.so, never the thing measured).
 *
 * CPU restatement, in plain C, of the detector's host-side post-processing:
 *   anchors   <- /root/reference/src/retinaface.cpp:210-240  (create_anchor_retinaface)
 *   decode    <- /root/reference/src/retinaface.cpp:159-203  (threshold, decode, un-letterbox, clip)
 *   sort/nms  <- /root/reference/src/retinaface.cpp:204-208, 242-271
 *
 * PARITY UNPINNED BY EXECUTION: the reference translation unit needs NvInfer.h, cublasLt.h and OpenCV headers, none of
 * which exist in this image, so it cannot be compiled here without writing stand-ins (not allowed) and the reference
 * has no tests/golden vectors.  Anchors are pinned on the values SURVEY.md §8(c) recorded from the reference
 * (16 800 anchors at 640x640, first (0.00625,0.00625,0.015625,0.015625), last (0.975,0.975,0.4,0.4)); decode/NMS are
 * cross-checked by an independent NumPy restatement in tests/test_oracle_postproc.py.
 *
 * Arithmetic notes (all deliberate):
 *   - "x" is the ROW axis, "y" the COLUMN axis.
 *   - decode runs in double (literals 0.1 / 0.2, exp) and narrows to float per field; corner and un-letterbox
 *     expressions are float with int operands promoted; every assignment to an int field truncates toward zero.
 *   - build with -ffp-contract=off and no -march flags: the reference is baseline x86-64, no FMA.
 *   - score test is strict '>'; NMS suppresses on '>='; areas and intersections use the '+1' convention.
 *   - std::sort is unstable on equal scores; this restatement defines ties as "lower anchor index first".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t x1, y1, x2, y2;
    float score;
} orc_bbox;

typedef struct {
    float cx, cy, sx, sy;
} orc_anchor;

static const int k_min_sizes[3][2] = {{10, 20}, {32, 64}, {128, 256}};
static const float k_steps[3] = {8.f, 16.f, 32.f};

int orc_anchor_count(int w, int h) {
    int n = 0;
    for (int k = 0; k < 3; ++k) n += (int)ceil(h / k_steps[k]) * (int)ceil(w / k_steps[k]) * 2;
    return n;
}

/* retinaface.cpp:210-240 */
int orc_make_anchors(int w, int h, orc_anchor *out) {
    int n = 0;
    for (int k = 0; k < 3; ++k) {
        int fh = (int)ceil(h / k_steps[k]);
        int fw = (int)ceil(w / k_steps[k]);
        for (int i = 0; i < fh; ++i)
            for (int j = 0; j < fw; ++j)
                for (int l = 0; l < 2; ++l) {
                    orc_anchor a;
                    a.sx = (float)(k_min_sizes[k][l] * 1.0 / w);
                    a.sy = (float)(k_min_sizes[k][l] * 1.0 / h);
                    a.cx = (float)((j + 0.5) * k_steps[k] / w);
                    a.cy = (float)((i + 0.5) * k_steps[k] / h);
                    out[n++] = a;
                }
    }
    return n;
}

static int clipi(int v, int lo, int hi) { /* CLIP(a,min,max) = MAX(MIN(a,max),min), retinaface.h:9 */
    int t = v < hi ? v : hi;
    return t > lo ? t : lo;
}

typedef struct {
    orc_bbox b;
    int32_t idx;
} cand_t;

static int cand_cmp(const void *pa, const void *pb) {
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->b.score > b->b.score) return -1;
    if (a->b.score < b->b.score) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/*
 * One frame.  loc[A*4], conf[A*2] -> out[<=max_faces]; returns the number of boxes.  If cand_out != NULL it receives all
 * thresholded+decoded candidates in anchor order (cand_idx = anchor indices) and *n_cand their count.
 */
int orc_postprocess(const float *loc, const float *conf, int in_w, int in_h, int frame_w, int frame_h, float nms_thr,
                    float bbox_thr, int max_faces, orc_bbox *out, orc_bbox *cand_out, int32_t *cand_idx, int *n_cand) {
    const float scale_h = (float)in_h / frame_h; /* retinaface.cpp:21-22 */
    const float scale_w = (float)in_w / frame_w;
    int A = orc_anchor_count(in_w, in_h);
    orc_anchor *anc = (orc_anchor *)malloc(sizeof(orc_anchor) * (size_t)A);
    cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)A);
    orc_make_anchors(in_w, in_h, anc);
    int n = 0;
    for (int i = 0; i < A; ++i) {
        float score = conf[2 * i + 1];
        if (!(score > bbox_thr)) continue;
        const float *bb = loc + 4 * i;
        orc_anchor a = anc[i], d;
        d.cx = (float)(a.cx + bb[0] * 0.1 * a.sx);
        d.cy = (float)(a.cy + bb[1] * 0.1 * a.sy);
        d.sx = (float)(a.sx * exp(bb[2] * 0.2));
        d.sy = (float)(a.sy * exp(bb[3] * 0.2));
        orc_bbox r;
        r.y1 = (int)((d.cx - d.sx / 2) * in_w);
        r.x1 = (int)((d.cy - d.sy / 2) * in_h);
        r.y2 = (int)((d.cx + d.sx / 2) * in_w);
        r.x2 = (int)((d.cy + d.sy / 2) * in_h);
        if (scale_h > scale_w) {
            r.y1 = (int)(r.y1 / scale_w);
            r.y2 = (int)(r.y2 / scale_w);
            r.x1 = (int)((r.x1 - (in_h - scale_w * frame_h) / 2) / scale_w);
            r.x2 = (int)((r.x2 - (in_h - scale_w * frame_h) / 2) / scale_w);
        } else {
            r.y1 = (int)((r.y1 - (in_w - scale_h * frame_w) / 2) / scale_h);
            r.y2 = (int)((r.y2 - (in_w - scale_h * frame_w) / 2) / scale_h);
            r.x1 = (int)(r.x1 / scale_h);
            r.x2 = (int)(r.x2 / scale_h);
        }
        r.y1 = clipi(r.y1, 0, frame_w - 1);
        r.x1 = clipi(r.x1, 0, frame_h - 1);
        r.y2 = clipi(r.y2, 0, frame_w - 1);
        r.x2 = clipi(r.x2, 0, frame_h - 1);
        r.score = score;
        c[n].b = r;
        c[n].idx = i;
        ++n;
    }
    if (cand_out) {
        for (int i = 0; i < n; ++i) {
            cand_out[i] = c[i].b;
            if (cand_idx) cand_idx[i] = c[i].idx;
        }
    }
    if (n_cand) *n_cand = n;

    qsort(c, (size_t)n, sizeof(cand_t), cand_cmp);

    /* greedy NMS, retinaface.cpp:248-271 (erase == mark dead; order of survivors is unchanged) */
    float *area = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    char *dead = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; ++i) area[i] = (float)((c[i].b.x2 - c[i].b.x1 + 1) * (c[i].b.y2 - c[i].b.y1 + 1));
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        for (int j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            float xx1 = (float)(c[i].b.x1 > c[j].b.x1 ? c[i].b.x1 : c[j].b.x1);
            float yy1 = (float)(c[i].b.y1 > c[j].b.y1 ? c[i].b.y1 : c[j].b.y1);
            float xx2 = (float)(c[i].b.x2 < c[j].b.x2 ? c[i].b.x2 : c[j].b.x2);
            float yy2 = (float)(c[i].b.y2 < c[j].b.y2 ? c[i].b.y2 : c[j].b.y2);
            float w = xx2 - xx1 + 1;
            float h = yy2 - yy1 + 1;
            if (w < 0.f) w = 0.f;
            if (h < 0.f) h = 0.f;
            float inter = w * h;
            float ovr = inter / (area[i] + area[j] - inter);
            if (ovr >= nms_thr) dead[j] = 1;
        }
    }
    int m = 0;
    for (int i = 0; i < n && m < max_faces; ++i) /* cap AFTER nms, retinaface.cpp:206-207 */
        if (!dead[i]) out[m++] = c[i].b;
    free(area);
    free(dead);
    free(c);
    free(anc);
    return m;
}
