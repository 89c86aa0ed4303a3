// Detector post-processing on the device: analytic anchors + decode + threshold + compaction, then greedy NMS.
//
// Replaces RetinaFace::postprocessing / create_anchor_retinaface / nms (/root/reference/src/retinaface.cpp:154-271), which
// the reference runs on the host after a blocking D2H copy and which rebuilds all 16 800 anchors with push_back per frame.
//
// Exactness contract: box coordinates are ints obtained by truncating float expressions twice, so
// the arithmetic below reproduces the reference's precision mix operation by operation: priors and decode in double
// narrowed to float per field, corner / un-letterbox expressions in float with int operands promoted, IEEE division, no
// FMA contraction (this translation unit is compiled with -ffp-contract=off).  Strict '>' score test, '>=' suppression,
// '+1' areas, cap to max_faces AFTER the NMS.  std::sort is unstable on equal scores; here ties are defined as "lower
// anchor index first" (same rule as oracle/postproc.c).
//
// Greedy NMS followed by "keep the first K survivors" is computed as K rounds of {arg-max over the live candidates,
// suppress everything overlapping the winner}: identical output, no sort, O(K * n / 256) per frame.
#include "frt_kernels.h"

#include <limits.h>

namespace {

constexpr int CC_STRIDE = 32;
__constant__ int c_min_sizes[3][2] = {{10, 20}, {32, 64}, {128, 256}};
__constant__ float c_steps[3] = {8.f, 16.f, 32.f};

__device__ __forceinline__ int clipi(int v, int lo, int hi) {
    const int t = v < hi ? v : hi;
    return t > lo ? t : lo;
}

// Decode of ONE anchor (retinaface.cpp:161-187, 225-236): priors, box decode, corner truncation, un-letterbox, clip.
__device__ __forceinline__ frt_bbox decode_box(const DetGeom &g, int a, const float *__restrict__ loc_f, float score) {
    const int k = a >= g.base[2] ? 2 : (a >= g.base[1] ? 1 : 0);
    const int rel = a - g.base[k];
    const int l = rel & 1, cell = rel >> 1;
    const int i = cell / g.fw[k], j = cell - i * g.fw[k];
    const int w = g.in_w, h = g.in_h;
    const float step = c_steps[k];
    const int ms = c_min_sizes[k][l];
    // priors: double arithmetic narrowed to float (retinaface.cpp:230-233)
    const float asx = (float)(ms * 1.0 / w);
    const float asy = (float)(ms * 1.0 / h);
    const float acx = (float)((j + 0.5) * step / w);
    const float acy = (float)((i + 0.5) * step / h);

    const floatx4 bb = *reinterpret_cast<const floatx4 *>(loc_f + (long)a * 4);
    const float l0 = bb[0], l1 = bb[1], l2 = bb[2], l3 = bb[3];
    // decode (retinaface.cpp:166-169): double intermediates, float fields
    const float cx = (float)(acx + l0 * 0.1 * asx);
    const float cy = (float)(acy + l1 * 0.1 * asy);
    const float sx = (float)(asx * exp(l2 * 0.2));
    const float sy = (float)(asy * exp(l3 * 0.2));

    frt_bbox r;
    r.y1 = (int)((cx - sx / 2) * w);
    r.x1 = (int)((cy - sy / 2) * h);
    r.y2 = (int)((cx + sx / 2) * w);
    r.x2 = (int)((cy + sy / 2) * h);
    if (g.scale_h > g.scale_w) {
        r.y1 = (int)(r.y1 / g.scale_w);
        r.y2 = (int)(r.y2 / g.scale_w);
        r.x1 = (int)((r.x1 - (h - g.scale_w * g.frame_h) / 2) / g.scale_w);
        r.x2 = (int)((r.x2 - (h - g.scale_w * g.frame_h) / 2) / g.scale_w);
    } else {
        r.y1 = (int)((r.y1 - (w - g.scale_h * g.frame_w) / 2) / g.scale_h);
        r.y2 = (int)((r.y2 - (w - g.scale_h * g.frame_w) / 2) / g.scale_h);
        r.x1 = (int)(r.x1 / g.scale_h);
        r.x2 = (int)(r.x2 / g.scale_h);
    }
    r.y1 = clipi(r.y1, 0, g.frame_w - 1);
    r.x1 = clipi(r.x1, 0, g.frame_h - 1);
    r.y2 = clipi(r.y2, 0, g.frame_w - 1);
    r.x2 = clipi(r.x2, 0, g.frame_h - 1);
    r.score = score;
    return r;
}

// The per-frame candidate counters live CC_STRIDE ints (one 128-byte line) apart.  Round 5: as adjacent ints all ~ 4 000 returning atomics of
// a 32-frame call - from every XCD - queued on ONE cache line and decode_kernel took 26 us for 4 MB of input, whatever else it did (one atomic
// per candidate or per wave, box arithmetic in or out: 25.6 - 27.5 us); a line per frame lets the frames' atomics proceed side by side.
// Threshold + compaction: one thread per (frame, anchor).  The candidate list of a frame is an unordered set (the NMS picks by (score, anchor
// index), never by list position), so the compaction only has to be dense: a WAVE counts its candidates with one ballot, its first lane
// reserves that many slots with ONE atomic and every candidate takes the slot at its rank among the wave's candidates.  Round 5: this kernel
// only records (anchor, score); the box arithmetic - two double-precision exp and four double divisions per candidate - runs in nms_kernel on
// the DENSE list.  Here it ran under the threshold branch with ~ 1 % of the lanes alive: every wave that held a single candidate walked the
// whole double-precision path (27 us per 32 frames for ~ 5 000 candidates; the compaction atomics were never the cost).
__global__ __launch_bounds__(256) void decode_kernel(const float *__restrict__ conf, DetGeom g, Candidate *__restrict__ cand, int *__restrict__ cand_count) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const bool in_range = a < g.A;
    const float score = in_range ? conf[((long)f * g.A + a) * 2 + 1] : 0.f;
    const bool take = in_range && (score > g.bbox_thr);
    const unsigned long long mask = __ballot(take);
    if (mask == 0ull) return;  // (wave-uniform)
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(&cand_count[f * CC_STRIDE], __popcll(mask));
    base = __shfl(base, 0);
    if (!take) return;
    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    Candidate c;
    c.box.x1 = c.box.y1 = c.box.x2 = c.box.y2 = 0;
    c.box.score = score;
    c.anchor = a;
    cand[(long)f * g.A + pos] = c;
}

__device__ __forceinline__ bool cand_better(float s, int a, float bs, int ba) { return (s > bs) || (s == bs && a < ba); }

__global__ __launch_bounds__(256) void nms_kernel(Candidate *__restrict__ cand_all, const float *__restrict__ loc, int *__restrict__ cand_count, DetGeom g,
                                                  uint8_t *__restrict__ dead_all, frt_bbox *__restrict__ out, int *__restrict__ n_out,
                                                  int *__restrict__ kept_anchor) {
    const int f = blockIdx.x, tid = threadIdx.x;
    Candidate *cand = cand_all + (long)f * g.A;
    uint8_t *dead = dead_all + (long)f * g.A;
    const int n = cand_count[f * CC_STRIDE];
    __shared__ float s_score[4];
    __shared__ int s_anchor[4];
    __shared__ int s_pos[4];
    __shared__ Candidate s_win;
    __shared__ int s_has;

    int kept = 0;
    if (n <= 256) {
        // The usual case (round 5): the frame's candidates fit one per thread - decoded into LDS once, the K rounds of {arg-max, suppress} then run
        // on registers and LDS only (the general loop below goes back to global memory for every candidate in every round: 10 us of dependent
        // L2 round trips per call, whatever the batch).  Same selection rule, same IoU arithmetic.
        __shared__ Candidate s_c[256];
        Candidate c;
        bool alive = tid < n;
        if (alive) {
            c = cand[tid];
            c.box = decode_box(g, c.anchor, loc + (long)f * g.A * 4, c.box.score);
            s_c[tid] = c;
        }
        const float area = alive ? (float)((c.box.x2 - c.box.x1 + 1) * (c.box.y2 - c.box.y1 + 1)) : 0.f;
        __syncthreads();
        for (; kept < g.max_faces; ++kept) {
            float bs = alive ? c.box.score : -INFINITY;
            int ba = alive ? c.anchor : INT_MAX, bp = alive ? tid : -1;
            for (int off = 32; off > 0; off >>= 1) {
                const float os = __shfl_xor(bs, off);
                const int oa = __shfl_xor(ba, off);
                const int op = __shfl_xor(bp, off);
                if (op >= 0 && (bp < 0 || cand_better(os, oa, bs, ba))) {
                    bs = os;
                    ba = oa;
                    bp = op;
                }
            }
            if ((tid & 63) == 0) {
                s_score[tid >> 6] = bs;
                s_anchor[tid >> 6] = ba;
                s_pos[tid >> 6] = bp;
            }
            __syncthreads();
            // every thread picks the winner of the four wave winners (same data, same rule: no second barrier needed for a broadcast)
            float ws = s_score[0];
            int wa = s_anchor[0], wp = s_pos[0];
            for (int w = 1; w < 4; ++w)
                if (s_pos[w] >= 0 && (wp < 0 || cand_better(s_score[w], s_anchor[w], ws, wa))) {
                    ws = s_score[w];
                    wa = s_anchor[w];
                    wp = s_pos[w];
                }
            if (wp < 0) break;  // (uniform)
            const frt_bbox wb = s_c[wp].box;
            if (tid == wp) {
                alive = false;
                out[(long)f * g.max_faces + kept] = wb;
                if (kept_anchor) kept_anchor[(long)f * g.max_faces + kept] = c.anchor;
            }
            if (alive) {
                const float warea = (float)((wb.x2 - wb.x1 + 1) * (wb.y2 - wb.y1 + 1));
                const frt_bbox b = c.box;
                const float xx1 = (float)(wb.x1 > b.x1 ? wb.x1 : b.x1);
                const float yy1 = (float)(wb.y1 > b.y1 ? wb.y1 : b.y1);
                const float xx2 = (float)(wb.x2 < b.x2 ? wb.x2 : b.x2);
                const float yy2 = (float)(wb.y2 < b.y2 ? wb.y2 : b.y2);
                float w = xx2 - xx1 + 1;
                float h = yy2 - yy1 + 1;
                if (w < 0.f) w = 0.f;
                if (h < 0.f) h = 0.f;
                const float inter = w * h;
                const float ovr = inter / (warea + area - inter);
                if (ovr >= g.nms_thr) alive = false;
            }
            __syncthreads();  // (s_score / s_anchor / s_pos are rewritten in the next round)
        }
    } else {
    for (int i = tid; i < n; i += 256) {  // boxes of the frame's candidates (decode_kernel left anchor + score)
        dead[i] = 0;
        cand[i].box = decode_box(g, cand[i].anchor, loc + (long)f * g.A * 4, cand[i].box.score);
    }
    __syncthreads();

    for (; kept < g.max_faces; ++kept) {
        float bs = -INFINITY;
        int ba = INT_MAX, bp = -1;
        for (int i = tid; i < n; i += 256) {
            if (dead[i]) continue;
            const float s = cand[i].box.score;
            const int a = cand[i].anchor;
            if (bp < 0 || cand_better(s, a, bs, ba)) {
                bs = s;
                ba = a;
                bp = i;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oa = __shfl_xor(ba, off);
            const int op = __shfl_xor(bp, off);
            if (op >= 0 && (bp < 0 || cand_better(os, oa, bs, ba))) {
                bs = os;
                ba = oa;
                bp = op;
            }
        }
        if ((tid & 63) == 0) {
            s_score[tid >> 6] = bs;
            s_anchor[tid >> 6] = ba;
            s_pos[tid >> 6] = bp;
        }
        __syncthreads();
        if (tid == 0) {
            float ws = s_score[0];
            int wa = s_anchor[0], wp = s_pos[0];
            for (int w = 1; w < 4; ++w)
                if (s_pos[w] >= 0 && (wp < 0 || cand_better(s_score[w], s_anchor[w], ws, wa))) {
                    ws = s_score[w];
                    wa = s_anchor[w];
                    wp = s_pos[w];
                }
            s_has = wp >= 0;
            if (wp >= 0) {
                s_win = cand[wp];
                dead[wp] = 1;
                out[(long)f * g.max_faces + kept] = cand[wp].box;
                if (kept_anchor) kept_anchor[(long)f * g.max_faces + kept] = cand[wp].anchor;
            }
        }
        __syncthreads();
        if (!s_has) break;
        const frt_bbox wb = s_win.box;
        const float warea = (float)((wb.x2 - wb.x1 + 1) * (wb.y2 - wb.y1 + 1));
        for (int i = tid; i < n; i += 256) {
            if (dead[i]) continue;
            const frt_bbox b = cand[i].box;
            const float area = (float)((b.x2 - b.x1 + 1) * (b.y2 - b.y1 + 1));
            const float xx1 = (float)(wb.x1 > b.x1 ? wb.x1 : b.x1);
            const float yy1 = (float)(wb.y1 > b.y1 ? wb.y1 : b.y1);
            const float xx2 = (float)(wb.x2 < b.x2 ? wb.x2 : b.x2);
            const float yy2 = (float)(wb.y2 < b.y2 ? wb.y2 : b.y2);
            float w = xx2 - xx1 + 1;
            float h = yy2 - yy1 + 1;
            if (w < 0.f) w = 0.f;
            if (h < 0.f) h = 0.f;
            const float inter = w * h;
            const float ovr = inter / (warea + area - inter);
            if (ovr >= g.nms_thr) dead[i] = 1;
        }
        __syncthreads();
    }
    }
    if (tid == 0) {
        n_out[f] = kept;
        cand_count[f * CC_STRIDE] = 0;  // every thread read it before the first barrier; decode_kernel of the next call counts from zero (no memset launch)
    }
    // zero the unused slots so the results buffer is deterministic
    for (int i = kept + tid; i < g.max_faces; i += 256) {
        frt_bbox z;
        z.x1 = z.y1 = z.x2 = z.y2 = 0;
        z.score = 0.f;
        out[(long)f * g.max_faces + i] = z;
    }
}

}  // namespace

void launch_decode(const float *loc, const float *conf, int n_frames, const DetGeom &g, Candidate *cand, int *cand_count, hipStream_t s) {
    // cand_count is zero here: allocated zeroed, and nms_kernel (always launched after this kernel) resets the entries it consumed
    dim3 grid((g.A + 255) / 256, n_frames);
    (void)loc;  // (read by nms_kernel since round 5)
    hipLaunchKernelGGL(decode_kernel, grid, dim3(256), 0, s, conf, g, cand, cand_count);
}

void launch_nms(Candidate *cand, const float *loc, int *cand_count, int n_frames, const DetGeom &g, uint8_t *dead, frt_bbox *out, int *n_out,
                int *kept_anchor, hipStream_t s) {
    hipLaunchKernelGGL(nms_kernel, dim3(n_frames), dim3(256), 0, s, cand, loc, cand_count, g, dead, out, n_out, kept_anchor);
}

// ---------------------------------------------------------------- landmarks of the kept boxes (optional alignment mode)
// The reference exports the detector WITHOUT the landmark head and has no landmark decode (src/retinaface.cpp:58-60,
// conversion/retina/torch2trt.py:7-9), so this follows the upstream RetinaFace formula for the head defined in
// conversion/retina/models/retinaface.py:37-46:  point_k = prior_centre + pre[2k..2k+1] * variance[0] * prior_size, then the
// same un-letterboxing as the boxes - in float, without the reference's int truncation.  Parity unpinned (oracle/align.py).
namespace {
__global__ void landmark_decode_kernel(const float *__restrict__ ldm, const int *__restrict__ kept_anchor, const int *__restrict__ n_out,
                                       DetGeom g, float *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (t >= g.max_faces * 5) return;
    const int kbox = t / 5, k = t - kbox * 5;
    float x = 0.f, y = 0.f;
    if (kbox < n_out[f]) {
        const int a = kept_anchor[(long)f * g.max_faces + kbox];
        const int lv = a >= g.base[2] ? 2 : (a >= g.base[1] ? 1 : 0);
        const int rel = a - g.base[lv];
        const int l = rel & 1, cell = rel >> 1;
        const int i = cell / g.fw[lv], j = cell - i * g.fw[lv];
        const float step = c_steps[lv];
        const float asx = (float)c_min_sizes[lv][l] / (float)g.in_w, asy = (float)c_min_sizes[lv][l] / (float)g.in_h;
        const float acx = ((float)j + 0.5f) * step / (float)g.in_w, acy = ((float)i + 0.5f) * step / (float)g.in_h;
        const float *p = ldm + ((long)f * g.A + a) * 10 + 2 * k;
        x = (acx + p[0] * 0.1f * asx) * (float)g.in_w;
        y = (acy + p[1] * 0.1f * asy) * (float)g.in_h;
        if (g.scale_h > g.scale_w) {
            x = x / g.scale_w;
            y = (y - ((float)g.in_h - g.scale_w * (float)g.frame_h) / 2.f) / g.scale_w;
        } else {
            x = (x - ((float)g.in_w - g.scale_h * (float)g.frame_w) / 2.f) / g.scale_h;
            y = y / g.scale_h;
        }
    }
    float *o = out + ((long)f * g.max_faces + kbox) * 10 + 2 * k;
    o[0] = x;
    o[1] = y;
}
}  // namespace

void launch_landmark_decode(const float *ldm, const int *kept_anchor, const int *n_out, int n_frames, const DetGeom &g, float *out, hipStream_t s) {
    dim3 grid((g.max_faces * 5 + 63) / 64, n_frames);
    hipLaunchKernelGGL(landmark_decode_kernel, grid, dim3(64), 0, s, ldm, kept_anchor, n_out, g, out);
}

// ---------------------------------------------------------------- per-face result records of the batched pipeline
namespace {
__global__ void pack_results_kernel(const frt_bbox *__restrict__ boxes, const int *__restrict__ n_boxes, const int *__restrict__ valid,
                                    const int32_t *__restrict__ idx, const float *__restrict__ sim, int max_faces, int F,
                                    frt_face_result *__restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    frt_face_result r;
    r.frame = f / max_faces;
    const bool used = (f % max_faces) < n_boxes[r.frame];
    // an unused slot reads as all zeros (score 0 - a detection's score is above the threshold): "box present, ROI empty" (valid 0, score > 0)
    // stays distinguishable from "no box", which the request coalescer needs to hand findFace its exact box list
    r.box = used ? boxes[f] : frt_bbox{0, 0, 0, 0, 0.f};
    const bool ok = used && valid[f] != 0;
    r.valid = ok ? 1 : 0;
    r.match_idx = (ok && idx) ? idx[f] : -1;
    r.match_sim = (ok && sim) ? sim[f] : 0.f;
    out[f] = r;
}
}  // namespace

void launch_pack_results(const frt_bbox *boxes, const int *n_boxes, const int *valid, const int32_t *idx, const float *sim, int max_faces,
                         int F, frt_face_result *out, hipStream_t s) {
    hipLaunchKernelGGL(pack_results_kernel, dim3((F + 255) / 256), dim3(256), 0, s, boxes, n_boxes, valid, idx, sim, max_faces, F, out);
}
