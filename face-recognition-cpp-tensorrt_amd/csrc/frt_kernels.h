// Internal declarations shared by the HIP translation units of libfrt.so (gfx950 only; no other targets, no shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/frt.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// A/B switches of the kernels (environment variables, read once per process).  They exist for measurements
// and are compiled in only with `make TUNING=1` (-DFRT_TUNING -DFRT_ABLATE); the default library ignores them and contains none
// of the timing-ablation kernel variants (which produce wrong results by design).
#ifdef FRT_TUNING
#include <stdlib.h>
inline const char *frt_tuning_env(const char *name) { return getenv(name); }
#else
inline const char *frt_tuning_env(const char *) { return nullptr; }
#endif

// One-time per-DEVICE setup of a kernel (hipFuncSetAttribute for > 64 KB of dynamic LDS is per device, not per process).
constexpr int FRT_MAX_DEVICES = 32;
inline bool frt_first_use_on_device(bool (&done)[FRT_MAX_DEVICES]) {
    int d = 0;
    (void)hipGetDevice(&d);
    d &= FRT_MAX_DEVICES - 1;
    if (done[d]) return false;
    done[d] = true;
    return true;
}

// ---------------------------------------------------------------- utilities (kernels_util.hip)
void launch_spin(double microseconds, hipStream_t s);  // one wave busy-waiting on the constant device clock
// sustained matrix-core rate probe (kernels_util.hip); returns the flop of the launch
double launch_mfma_probe(int mix, int n_wg, int iters, const void *src, float *out, hipStream_t s);

// ---------------------------------------------------------------- match (kernels_match.hip)
struct MatchPartial {  // one per (workgroup, query)
    float sim;
    int32_t idx;
};
// top-1: gallery [N][512] fp32, queries [F][512] fp32 -> idx[F], sim[F].  partial: scratch [grid][F].
void launch_match_top1(const float *gallery, int N, int D, const float *queries, int F, MatchPartial *partial, int partial_blocks,
                       int32_t *idx_out, float *sim_out, int row_offset, hipStream_t s);
int match_top1_blocks(int N, int F);
// screened top-1: fp16 shadow gallery + coarse MFMA pass + exact re-rank of the few tiles that can hold the maximum
constexpr int FRT_MATCH_CTL_WORDS = 32 + 64 * 32;
struct ScreenScratch {
    half_t *q16;       // [F][D]
    float *tilemax;    // [F][tiles][sub], sub = 1 or 4 coarse maxima per 128-row tile
    int *tile_flags;   // [tiles]
    int *tile_list;    // [tiles]
    int *count;        // [1], == tile_flags + tiles (cleared together)
    float *segmax;     // [F][16] per-segment maxima of tilemax (selection step 1)
    // fast top-1 path (round 3): null -> the tile-list path
    float *wgmax;                 // [coarse workgroups][F] maxima of a workgroup's coarse entries
    void *pairs;                  // (query, tile) candidate pairs
    int pair_cap;
    int *ctl;                     // [FRT_MATCH_CTL_WORDS] control words: overflow flag, per-sub-list pair counts, one 128-byte line each (kernels_match.hip)
    unsigned long long *qkey;     // [F] packed (similarity, ~row) winners of the scalar re-rank
    // int8 shadow gallery (round 4; fast path, D = 512, fp32-stored galleries): null -> the fp16 shadow is scanned
    const uint8_t *g8;            // fragment-ordered biased bytes (value + 128), gallery8_bytes(N, D)
    const float *g8_scale;        // [tiles * 128] per-row scale (row = scale * int8 row + error)
    float gerr;                   // max over rows of || row - scale * int8 row ||
};
size_t gallery8_bytes(int N, int D);
// fp32 rows -> int8 shadow (+ per-row scales, largest quantisation error norm^2 and largest row norm^2 as float bit patterns, atomicMax)
void launch_gallery_shadow8(const float *gallery, int N, int D, uint8_t *g8, float *scale, int *max_err2_bits, int *max_norm2_bits, hipStream_t s);
void launch_gallery_shadow(const float *gallery, int N, int D, half_t *g16, int *max_norm2_bits, hipStream_t s);
void launch_match_top1_screened(const float *gallery, const half_t *g16, int N, int D, const float *queries, int F, float gmax_norm,
                                const ScreenScratch &w, MatchPartial *partial, int partial_blocks, int32_t *idx_out, float *sim_out,
                                int row_offset, hipStream_t s);
// exact top-k [F][k] (k passes of the top-1 search over the rows behind the previous winner; screened galleries scan coarsely once)
void launch_match_topk(const float *gallery, const half_t *g16, int N, int D, const float *queries, int F, int k, bool screen, float gmax_norm,
                       const ScreenScratch &w, float *kth_scratch, MatchPartial *partial, int partial_blocks, int32_t *idx_out, float *sim_out,
                       int row_offset, hipStream_t s);
int match_topk_max();
void launch_half_to_float(const half_t *in, long n, float *out, hipStream_t s);
void launch_float_to_half(const float *in, long n, half_t *out, hipStream_t s);
void launch_merge_topk(const int32_t *idx_all, const float *sim_all, int shards, int n, int k, int32_t *idx_out, float *sim_out, hipStream_t s);
// full matrix: out[F][N]
void launch_match_full(const float *gallery, int N, int D, const float *queries, int F, float *out, hipStream_t s);
// fp16-STORED gallery (BASELINE config 5): same kernels, rows widened exactly to fp32 while they are staged.  For the screened
// top-1 pass gallery == nullptr and g16 = the stored rows.
void launch_match_top1_h(const half_t *g16, int N, int D, const float *queries, int F, MatchPartial *partial, int partial_blocks,
                         int32_t *idx_out, float *sim_out, int row_offset, hipStream_t s);
void launch_match_full_h(const half_t *g16, int N, int D, const float *queries, int F, float *out, hipStream_t s);
void launch_gallery_norm16(const half_t *g16, int N, int D, int *max_norm2_bits, hipStream_t s);
// The fp16 gallery (shadow or stored) is kept in MFMA-fragment order and padded to whole 128-row tiles (see kernels_match.hip):
size_t gallery16_elems(int N, int D);
bool match_screen_supported(int D);  // D the coarse kernel is instantiated for
// fp32 rows [n_rows][D] (first row = global row row0, a multiple of 128) -> their place in the fp16 gallery (which must be zero-filled first)
void launch_rows_to_half(const float *in, long row0, long n_rows, int D, half_t *g16, hipStream_t s);

// ---------------------------------------------------------------- post-processing (kernels_post.hip)
struct DetGeom {
    int in_w, in_h, frame_w, frame_h;
    int fw[3], fh[3];      // feature-map sizes per level (ceil(dim/step))
    int base[3];           // first anchor index per level
    int A;                 // anchors per frame
    float scale_w, scale_h;
    float nms_thr, bbox_thr;
    int max_faces;
};
struct Candidate {
    frt_bbox box;
    int32_t anchor;
};
void launch_decode(const float *loc, const float *conf, int n_frames, const DetGeom &g, Candidate *cand, int *cand_count, hipStream_t s);
void launch_nms(Candidate *cand, const float *loc, int *cand_count, int n_frames, const DetGeom &g, uint8_t *dead, frt_bbox *out, int *n_out,
                int *kept_anchor, hipStream_t s);
// landmarks of the kept boxes: raw head output ldm [B][A][10] + kept anchors [B][K] -> frame coordinates (x0,y0,...,x4,y4) [B][K][10]
void launch_landmark_decode(const float *ldm, const int *kept_anchor, const int *n_out, int n_frames, const DetGeom &g, float *out, hipStream_t s);
// 5-point similarity alignment (ArcFace 112x112 template) + bilinear warp, fused with the recogniser's normalisation
void launch_align_faces(const uint8_t *frames, int frame_h, int frame_w, size_t row_stride, size_t frame_stride, const float *landmarks,
                        const int *n_boxes, int max_faces, int F, int frames_shared, uint8_t *crops, float *chw, int *valid, hipStream_t s);

// ---------------------------------------------------------------- image ops (kernels_image.hip)
// u8 BGR frames [n][frame_h][frame_w][3] (row_stride / frame_stride in bytes) -> fp32 planar [n][3][in_h][in_w]
void launch_det_preprocess(const uint8_t *frames, int n, int frame_h, int frame_w, size_t row_stride, size_t frame_stride, int in_h,
                           int in_w, float *out, hipStream_t s);
// per face slot f: box = boxes[f]; valid iff f%max_faces < n_boxes[f/max_faces] (n_boxes == nullptr: all valid) and ROI non-empty.
// writes u8 BGR crops [F][oh][ow][3] (may be null), fp32 planar RGB normalised [F][3][oh][ow], valid flags [F].
void launch_crop_faces(const uint8_t *frames, int frame_h, int frame_w, size_t row_stride, size_t frame_stride, const frt_bbox *boxes,
                       const int *n_boxes, int max_faces, int F, int frames_shared, int oh, int ow, uint8_t *crops, float *chw, int *valid,
                       hipStream_t s);
// n frames u8 HWC [sh][sw][3] -> [dh][dw][3], cv::resize INTER_LINEAR semantics (frame ingest, app.cpp:301)
void launch_resize_linear(const uint8_t *src, int n, int sh, int sw, size_t sstride, size_t sframe, uint8_t *dst, int dh, int dw, size_t dstride,
                          size_t dframe, hipStream_t s);
void launch_face_normalize(const uint8_t *crops, int F, int oh, int ow, float *chw, hipStream_t s);

// ---------------------------------------------------------------- detector network (kernels_det.hip), fp32 NCHW
struct DwPwArgs {
    const float *in; float *out;
    const float *wd, *bd;   // depthwise [Cin][9], bias [Cin] (BN folded); null wd -> plain 1x1 conv
    const float *wp, *bp;   // pointwise, TRANSPOSED [Cin][Cout], bias [Cout]
    const float *add;       // optional tensor to add after ReLU, nearest-upsampled from [B][Cout][add_h][add_w]
    int add_h, add_w;
    int B, Cin, H, W, Cout, Ho, Wo, stride, relu;
    float *tmp;             // scratch [B][Cin][Ho][Wo] for the split depthwise -> pointwise path (null: always fused)
    const float *wd12;      // depthwise weights packed [Cin][12] = 9 taps, bias, 2 pad (matrix-core kernel); null: scalar kernels only
    const half_t *wph;      // pointwise weights as fp16 hi/lo split [Cout][Cin/16][hi16 | lo16] (Cin % 16 == 0); null: fp32 MFMA path
    const float *wdp;       // depthwise weights of channel pairs [Cin/2][10][2] (kernels_det_wave.hip); null: that kernel does not apply
    const half_t *wpf;      // the split pointwise weights in fragment order [Cin/16][Cout/32][hi|lo][64][8] (kernels_det_wave.hip)
    const float *wdt;       // the same depthwise weights tap-major [10][Cin/2][2] (tap 9 = bias; kernels_det_stem.hip); Cin <= 16 only
    const float *stem;      // 8 -> 16 block only: the gathered weights of the fused stem kernel (det_stem_pack, kernels_det_stem.hip); null: not fused
    const float *zeros;     // dwpw_wave_zero_bytes() of zeros (kernels_det_wave.hip: the source of input rows outside the image)
};
size_t dwpw_wave_zero_bytes();
bool launch_dwpw_wave(const DwPwArgs &a, hipStream_t s);  // one wave = 64 pixels x all channels (round 4); false: shape not covered
void launch_dwpw(const DwPwArgs &a, hipStream_t s);
bool launch_dwpw_mfma(const DwPwArgs &a, hipStream_t s);  // false: shape not covered, use the scalar kernels
struct Conv3Args {
    const float *in; float *out;
    const float *w, *b;     // [Cin][9][Cout] (transposed), bias [Cout]
    int B, Cin, H, W, Cout, Ho, Wo, stride, relu;
    int out_ctotal, out_coff;  // write into channels [coff, coff+Cout) of a [B][ctotal][Ho][Wo] tensor
    // matrix-core path (kernels_det_conv3.hip); all optional
    const float *wm;           // host-packed weights [9][Cin/wm_kc][wm_cpad][wm_kc] (zero rows beyond Cout); null: scalar kernel only
    int wm_kc, wm_cpad;
    const half_t *wh;          // fp16 hi/lo split weights [Cin/16][9][64][hi16|lo16] (kernels_det_conv3h.hip, Cin == 64); null: n/a
    float *out2;               // channels >= split go to out2 (channel co - split of a [B][out2_ctotal][Ho][Wo] tensor, + out2_coff)
    int split, out2_ctotal, out2_coff;
};
bool det_mfma_enabled();       // env FRT_DET_MFMA=0 switches the detector back to the scalar kernels (A/B measurements)
void launch_conv3x3(const Conv3Args &a, hipStream_t s);
// the detector's first three layers in one kernel (kernels_det_stem.hip); false: not applicable, run them one by one
size_t det_stem_weight_floats();
void det_stem_pack(const Conv3Args &c, const DwPwArgs &d1, const DwPwArgs &d2, float *dst, hipStream_t s);
bool launch_det_stem(const uint8_t *frames, size_t row_stride, size_t frame_stride, const Conv3Args &c, const DwPwArgs &d1, const DwPwArgs &d2, hipStream_t s);
// first detector conv fed by the u8 frames directly (only valid when the letterbox is the identity); false: not applicable
bool launch_det_conv1_u8(const uint8_t *frames, size_t row_stride, size_t frame_stride, const Conv3Args &a, hipStream_t s);
// (A fused "stem" kernel - first conv + the two conv_dw blocks behind it with the intermediates in LDS - was tried and removed:
//  436-840 us against 302 us for the three separate kernels; with ~1150 weights it either spills SGPRs or hoists every LDS weight
//  read into 290 VGPRs, and the halo recompute plus LDS traffic eat the HBM saving.)
bool launch_conv3x3_split(const Conv3Args *a, int n, hipStream_t s);  // fp16 hi/lo split MFMA version (Cin == 64); false: n/a
bool launch_conv3x3_mfma(const Conv3Args *a, int n, hipStream_t s);   // false: shape not covered, use the scalar kernel
void launch_conv3x3_multi(const Conv3Args *a, int n, hipStream_t s);  // up to 3 same-Cout problems in one launch
struct HeadArgs {
    const float *in;        // [B][64][H][W]
    const float *wb, *bb;   // bbox head [64][8], [8]
    const float *wc, *bc;   // class head [64][4], [4]
    float *loc, *conf;      // [B][A][4], [B][A][2]
    int B, C, H, W, A, base;
    const float *wl, *bl;   // optional landmark head [64][20], [20] (null: trimmed network, the reference's default)
    float *ldm;             // [B][A][10]
};
void launch_heads(const HeadArgs &a, hipStream_t s);
void launch_heads_multi(const HeadArgs *a, int n, hipStream_t s);

// ---------------------------------------------------------------- recogniser network (kernels_arc.hip), fp16 NHWC + MFMA
enum { EPI_PRELU = 0, EPI_BN = 1, EPI_BN_ADD_BN = 2, EPI_PARTIAL = 3, EPI_BN_SE = 4 };  // EPI_BN_SE: BN -> SE gate -> + shortcut -> BN_next (IR-SE unit tail)
struct ConvMfmaArgs {
    const half_t *x;   // [B][H][W][Cin]
    const half_t *w;   // [Cout][ks*ks*Cin]
    const half_t *wf;  // optional fragment-ordered copy for the strip kernel: [Cout/32][Cin/64][9][4][64 lanes][8] (3x3, Cin % 64 == 0); null: none
    const half_t *wf2; // same for the stride-2 strip kernel (kernels_arc_s2.hip): taps in ITS step order 0,2,6,8,4,1,7,3,5; null: none
    int B, H, W, Cin, Ho, Wo, Cout, ks, stride, pad;
    int mode;
    const float *p0, *p1, *p2, *p3;
    const half_t *sc;  // shortcut tensor [B][sc_h][sc_w][Cout], sampled at (oh*sc_stride, ow*sc_stride)
    int sc_h, sc_w, sc_stride;
    // fused 1x1 stride-2 shortcut CONVOLUTION (stride-2 strip kernel, mode EPI_BN_ADD_BN, instead of `sc`): out = BN(conv3x3(x)) + BNsc(conv1x1_s2(scx))
    const half_t *scx;   // the unit's raw input [B][H][W][Csc]; null: shortcut tensor `sc`
    const half_t *wscf;  // 1x1 weights in fragment order [Cout/32][Csc/64][4 kk][64 lanes][8]
    const float *psc0, *psc1;  // the shortcut's folded BatchNorm
    int Csc;
    half_t *out0, *out1;
    float *outf;       // EPI_PARTIAL: [splits][M][Cout]
    int splits;
    const half_t *zeros;  // >= 16 bytes of zeros (source of padded taps for the LDS-DMA path)
    // IR-SE unit tail inside conv2's epilogue (mode EPI_BN_SE; only when conv_se_fused(args) - the strip kernel's main variant): y (out0) =
    // BN(conv) * gate + sc, z (out1) = y * p2 + p3, gate from the image-wide channel means through fc1 [C/16][C] / fc2 [C][C/16].
    float *se_pool;      // scratch [4][B][Cout] partial channel sums
    const float *se_w1, *se_w2;
    int *se_counter;     // [>= B] arrival counters, zero between launches; the gate-ready flags sit se_flag_off ints behind
    int se_flag_off;
    int se_epoch;        // launch number (> 0, different from every earlier launch's on this scratch): what the flags are set to
    int *se_error;       // error word in mapped host memory: set (to the launch number) when a hand-over wait timed out
};
bool conv_se_fused(const ConvMfmaArgs &a);  // a: the unit's conv2 described as EPI_BN_ADD_BN + the se_* scratch
void launch_conv_mfma(const ConvMfmaArgs &a, hipStream_t s);
bool conv_s2_applies(const ConvMfmaArgs &a);                // kernels_arc_s2.hip: 3x3 stride 2, Cout % 128 == 0 or 64 -> 64 at 112 -> 56
bool launch_conv_s2(const ConvMfmaArgs &a, hipStream_t s);
bool conv_s2_se_fused(const ConvMfmaArgs &a);  // IR-SE tail in the stride-2 strip kernel's epilogue (see conv_se_fused)
const char *conv_s2_label(const ConvMfmaArgs &a);
bool conv_small_applies(const ConvMfmaArgs &a);              // kernels_arc_small.hip: 3x3 convs of a small batch (few pixel tiles)
bool launch_conv_small(const ConvMfmaArgs &a, hipStream_t s);
bool conv_ks_applies(const ConvMfmaArgs &a);                 // kernels_arc_ks.hip: 3x3 stride 1 at 14x14x256 / 7x7x512, medium batches (K split over the waves)
bool launch_conv_ks(const ConvMfmaArgs &a, hipStream_t s);
bool conv64_applies(const ConvMfmaArgs &a);                 // kernels_arc_c64.hip: Cin = Cout = 64, 3x3, stride 1
bool launch_conv64(const ConvMfmaArgs &a, hipStream_t s);
const char *conv_kernel_label(const ConvMfmaArgs &a);  // kernel symbol (as rocprofv3 prints it) a launch resolves to
struct ArcInputArgs {
    const float *x;       // [F][3][112][112] planar RGB
    const float *w;       // [27][64]  (k = ci*9 + kh*3 + kw)
    const float *s0, *b0; // folded BN after the conv
    const float *slope;   // PReLU
    const float *s1, *b1; // next unit's leading BN
    half_t *y, *z;        // z [F][112][112][64]; y (shortcut of unit 0) only at even positions: [F][56][56][64]
    int F, H, W;
    const half_t *wh;     // matrix-core path: [64][32] fp16, k < 27: w * s0, k == 27: b0 (multiplies a constant 1), else 0
};
void launch_arc_input(const ArcInputArgs &a, hipStream_t s);
bool launch_arc_input_mfma(const ArcInputArgs &a, hipStream_t s);  // kernels_arc_input.hip; false: shape not covered
// output Linear 25088 -> 512 as 49 K-slices (kernels_arc_fc.hip): z [F][25088] fp16, wfrag = weights in MFMA-fragment order
// [512/32 output blocks][25088/16 k steps][64 lanes][8 halfs] (lane = (output row r, k half hi)), partial [49][F][512] fp32
void launch_fc_slices(const half_t *z, const half_t *wfrag, int F, float *partial, hipStream_t s);
// partial [splits][F][512] -> +bias -> BN1d -> L2 normalise -> out [F][512] fp32; rows with valid[f]==0 become zeros.
// fp32 end-to-end recogniser path (kernels_arc_f32.hip; frt_embedder_set_precision)
struct Conv32Args {
    const float *x, *w, *ps, *pb;  // w: fragment-ordered (pack_conv32_weights)
    float *out;
    int F, H, W, Cin, Ho, Wo, Cout, ks, stride, pad, mode;  // mode 0: PReLU(p0)  1: BN(p0, p1)  2: BN(p0, p1) + shortcut
    const float *p0, *p1, *sc;
    int sc_h, sc_w, sc_stride;
};
void launch_arc32_input(const float *x, const float *w, const float *s0, const float *b0, const float *slope, float *y, int F, hipStream_t s);
void launch_conv32(const Conv32Args &c, hipStream_t s);
void pack_conv32_weights(const float *w, int cout, int taps, int cin, float *out);  // host: [Cout][taps][Cin] -> the kernel's fragment order
void launch_fc32(const float *y, const float *sn, const float *bn, const float *w, float *out, int F, hipStream_t s);
void launch_se32(const float *res, const float *w1, const float *w2, float *gate, const float *sc, float *out, int F, int Ho, int Wo, int C, int sc_h, int sc_w, int sc_stride,
                 hipStream_t s);
void launch_fc_finalize(const float *partial, int splits, int F, const float *bias, const float *s, const float *b, const int *valid,
                        float *out, hipStream_t s_);
// SE tail (IR-SE): pool -> fc1 -> relu -> fc2 -> sigmoid -> scale, + shortcut, + next BN
struct SeArgs {
    const half_t *res;   // [F][H][W][C] = BN2(conv2)
    const float *w1;     // [C/16][C]
    const float *w2;     // [C][C/16]
    const half_t *sc; int sc_h, sc_w, sc_stride;
    const float *s1, *b1;
    half_t *y, *z;
    float *pool;         // scratch [4][F][C] (partial sums over pixel ranges)
    float *gate;         // scratch [F][C]
    int F, H, W, C;
    int *counter;        // [F] arrival counters of the pooling pass, zero between launches
};
void launch_se(const SeArgs &a, hipStream_t s);
