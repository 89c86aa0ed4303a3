/*
 * libfrt - MI355X-native (gfx950) detect -> crop -> embed -> match hot path; thin C ABI.
 *
 * This header is the drop-in boundary.  The reference has no C ABI: its boundary is three C++ classes
 * (RetinaFace / ArcFaceIR50 / MatMul) that sit directly on TensorRT + cuBLASLt + the CUDA runtime.  Every entry point
 * below names the reference member function (file:line under /root/reference) it replaces; the header-only C++ shells
 * in include/frt/{common,retinaface,arcface,matmul}.h rebuild the reference class surfaces on top of these calls.
 *
 * Conventions: plain pointers and sizes only; opaque handles; every function returns an frt_status (0 = ok) and
 * leaves a message retrievable with frt_last_error() (thread-local).  Host pointers unless the name ends in "_dev".
 * Threading (the reference objects are not thread-safe although the Crow server is multithreaded, src/app.cpp:367): every
 * object has a mutex around its entry points and object-level calls are synchronous; a pipeline borrows its three objects, takes
 * their mutexes while it enqueues, and leaves an event behind each stage that later object-level calls on the same object wait
 * for on the device - so object-level calls, frt_pipeline_run from several threads, and pipelined batches in flight may be
 * mixed freely.  Distinct objects may be used concurrently.
 *
 * Axis naming follows the reference: Bbox.x* are ROWS, Bbox.y* are COLUMNS (src/retinaface.cpp:165).
 */
#ifndef FRT_H
#define FRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum frt_status {
    FRT_OK = 0,
    FRT_ERR_INVALID = 1,     /* bad argument / shape                                   (reference: assert)            */
    FRT_ERR_NOT_FOUND = 2,   /* weight blob missing   (reference: throw std::logic_error("Cant find engine file"))   */
    FRT_ERR_FORMAT = 3,      /* weight blob malformed / wrong network kind                                           */
    FRT_ERR_DEVICE = 4,      /* HIP runtime failure   (reference: checkCudaStatus -> std::logic_error, common.cpp:43) */
    FRT_ERR_EMPTY = 5,       /* no faces / empty gallery (reference: throw const char*, src/arcface.cpp:198)          */
    FRT_ERR_EMPTY_ROI = 6,   /* zero-area crop rectangle (reference: cv::Exception from Mat::operator())             */
    FRT_ERR_CAPACITY = 7     /* more frames / faces than the object was created for                                  */
} frt_status;

/* Layout-identical to the reference's `struct Bbox` (src/common.h:13-16): 4 x int32 + float32 = 20 bytes. */
typedef struct frt_bbox {
    int32_t x1, y1, x2, y2; /* x = row, y = column */
    float score;
} frt_bbox;

typedef struct frt_detector frt_detector;
typedef struct frt_embedder frt_embedder;
typedef struct frt_matcher frt_matcher;
typedef struct frt_pipeline frt_pipeline;

const char *frt_last_error(void);
const char *frt_version(void);
/* Number of visible HIP devices (0 when none). */
int frt_device_count(void);
/* How long a blocking entry point (findFace, forward, frt_pipeline_wait ...) busy-polls for the device before it backs off: the wait spins
 * for `microseconds` (default 200, or FRT_WAIT_SPIN_US), then polls once per ~50 us sleep (the thread is off its core in between), then
 * parks in the interrupt wait.  Process-wide.  A server with many request threads keeps the default; a throughput driver that owns its
 * core may raise it (bench.py uses 50 000 and reports it).  Returns the previous value. */
long frt_set_wait_spin_us(long microseconds);
/* Measurement aid (no counterpart in the reference): what this device SUSTAINS on the recogniser's matrix-core instruction mix, in TFLOP/s.
 * Runs back-to-back v_mfma_f32_32x32x16_f16 on random fp16 operands on every SIMD for about `seconds` (0 < seconds <= 10) -
 * mix 0: MFMAs only; 1: + one ds_read_b128 per MFMA; 2: + the global loads of the dominant conv kernel's K loop as well - and returns
 * flop / elapsed.  The part is power-managed (1.4 kW), so this is well below the nominal 2.5 PFLOP/s; bench.py quotes the dominant
 * kernel against both (roofline.sustained_peak). */
int frt_probe_sustained_mfma(int device, int mix, double seconds, double *tflops_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Detector  ==  class RetinaFace (src/retinaface.h:18-23)
 * ------------------------------------------------------------------------------------------------------------------ */

/* RetinaFace::RetinaFace (src/retinaface.cpp:3-29) + loadEngine (:31-55) + preInference (:81-104).
 * weights_path: FRTW blob (kind 1) instead of a TensorRT engine.  in_c must be 3.  max_batch = frames per call. */
int frt_detector_create(const char *weights_path, int frame_w, int frame_h, int in_c, int in_h, int in_w, int max_batch,
                        int max_faces, float nms_threshold, float bbox_threshold, int device, frt_detector **out);
/* RetinaFace::~RetinaFace (src/retinaface.cpp:273-280) */
void frt_detector_destroy(frt_detector *d);
/* m_OUTPUT_SIZE_BASE (src/retinaface.cpp:13): anchors per frame */
int frt_detector_num_anchors(const frt_detector *d);
/* What the detector was created for (any out pointer may be NULL). */
int frt_detector_geometry(const frt_detector *d, int *frame_w, int *frame_h, int *max_batch, int *max_faces, int *device);

/* RetinaFace::findFace (src/retinaface.cpp:147-152).  bgr: u8 HWC frame of exactly frame_h x frame_w, row_stride bytes
 * per row.  out: capacity max_faces.  n_out: number of boxes written (score-descending, after NMS and cap). */
int frt_detector_find_faces(frt_detector *d, const uint8_t *bgr, int rows, int cols, size_t row_stride, frt_bbox *out, int *n_out);
/* New surface: n_frames <= max_batch contiguous frames; out[n_frames * max_faces], n_out[n_frames]. */
int frt_detector_find_faces_batch(frt_detector *d, const uint8_t *bgr, int n_frames, int rows, int cols, size_t row_stride,
                                  size_t frame_stride, frt_bbox *out, int *n_out);

/* RetinaFace::preprocess (src/retinaface.cpp:106-136): frame -> float32 planar [3][in_h][in_w] on the host. */
int frt_detector_preprocess(frt_detector *d, const uint8_t *bgr, int rows, int cols, size_t row_stride, float *chw_out);
/* RetinaFace::doInference (src/retinaface.cpp:138-145): bindings input_det -> output_det0 [batch][A][4],
 * output_det1 [batch][A][2] (softmaxed), conversion/retina/torch2trt.py:95-96. */
int frt_detector_infer(frt_detector *d, const float *chw, int batch, float *loc_out, float *conf_out);
/* RetinaFace::postprocessing (src/retinaface.cpp:154-208) for one frame of raw head outputs. */
int frt_detector_postprocess(frt_detector *d, const float *loc, const float *conf, frt_bbox *out, int *n_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Crop  ==  free function getCroppedFaces (src/arcface.h:17, src/arcface.cpp:3-17)
 * ------------------------------------------------------------------------------------------------------------------ */
/* crops_out: u8 BGR [n][out_h][out_w][3].  device < 0 -> current device.  Returns FRT_ERR_EMPTY_ROI if any box has
 * a zero-area or out-of-frame rectangle (nothing is written for that face; the others are still produced). */
int frt_crop_faces(const uint8_t *bgr, int rows, int cols, size_t row_stride, const frt_bbox *boxes, int n, int out_w, int out_h,
                   uint8_t *crops_out, int device);

/* ------------------------------------------------------------------------------------------------------------------
 * Embedder  ==  class ArcFaceIR50 minus the gallery (src/arcface.h:19-39)
 * ------------------------------------------------------------------------------------------------------------------ */

/* ArcFaceIR50::ArcFaceIR50 (src/arcface.cpp:21-43) + loadEngine (:45-69) + preInference (:88-103).
 * weights_path: FRTW blob kind 2 (IR-50, the reference's network) or kind 3 (IR-SE-50).  in_c,in_h,in_w must be
 * 3,112,112 and out_dim 512.  max_batch = faces per device launch (the reference default is 1, app/config.json:18). */
int frt_embedder_create(const char *weights_path, int in_c, int in_h, int in_w, int out_dim, int max_batch, int device,
                        frt_embedder **out);
void frt_embedder_destroy(frt_embedder *e);

/* IR-SE-50 only (no reference counterpart; the reference's engine is a black box): 1 (default, env FRT_SE_FUSED=0 turns it off) runs the
 * squeeze-and-excitation tail of a unit inside conv2's epilogue, where the workgroups of one face hand their partial channel sums
 * over through device-scope stores and a flag; 0 always uses the stand-alone pool + gate + apply launches (no cross-workgroup wait
 * anywhere).  Same arithmetic; the pooled sums are added in a different fixed order, so results agree to fp16 rounding of the gated activations (cosine >= 1 - 1e-5),
 * each mode bit-reproducible in itself.  A timed-out hand-over never kills the HIP context: the next synchronising call on the
 * embedder / pipeline returns FRT_ERR_DEVICE and this switch is the way back.  Takes effect for passes enqueued after the call; a
 * pipeline with hipGraph replay on must be told to re-capture (frt_pipeline_set_graph). */
int frt_embedder_set_se_fused(frt_embedder *e, int enable);
/* Arithmetic of the recogniser network for the passes enqueued after the call.  fp32 = 0 (default): fp16 activations and weights on the fp16
 * matrix cores with fp32 accumulation - what the reference's TensorRT engine is built with (conversion/arcface/torch2trt.py:42-43) and what
 * every throughput figure of this build is measured on (embeddings cosine-equal to the fp32 network within ~ 3e-6).  fp32 = 1: fp32
 * activations, fp32 weights, every product an exact fp32 fma (v_mfma_f32_32x32x2_f32 / v_fma_f32) - BASELINE configs[1]'s "fp32"; a simple,
 * separate path meant for a few faces per call (about 10x the time per face; 1 - cosine <= 1e-6 against the fp32 oracle).  The first call with
 * fp32 = 1 re-reads the weight blob the object was created from and uploads fp32 copies (175 MB).  Pipelines with hipGraph replay on must be
 * told to re-capture (frt_pipeline_set_graph). */
int frt_embedder_set_precision(frt_embedder *e, int fp32);

/* ArcFaceIR50::preprocessFace (src/arcface.cpp:105-114): u8 BGR [in_h][in_w][3] -> float32 planar RGB. */
int frt_embedder_preprocess_face(frt_embedder *e, const uint8_t *bgr_crop, float *chw_out);
/* ArcFaceIR50::doInference, both overloads (src/arcface.cpp:131-148): float32 [batch][3][112][112] -> [batch][512]
 * L2-normalised.  batch may exceed max_batch (processed in chunks). */
int frt_embedder_infer(frt_embedder *e, const float *chw, int batch, float *embeds_out);
/* ArcFaceIR50::forward (src/arcface.cpp:166-187) with the evident intent of the batched branch:
 * getCroppedFaces + preprocessFaces + inference.  embeds_out [n][512]; crops_out (may be NULL) u8 BGR [n][112][112][3]
 * (= CroppedFace.face, src/app.cpp:329). */
int frt_embedder_forward(frt_embedder *e, const uint8_t *bgr, int rows, int cols, size_t row_stride, const frt_bbox *boxes, int n,
                         float *embeds_out, uint8_t *crops_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Matcher  ==  class MatMul (src/matmul.h:6-21) + ArcFaceIR50::getOutputs argmax (src/arcface.cpp:203-217)
 * ------------------------------------------------------------------------------------------------------------------ */

/* MatMul::MatMul (src/matmul.cpp:3-7) */
int frt_matcher_create(int device, frt_matcher **out);
/* MatMul::~MatMul (src/matmul.cpp:79-91) */
void frt_matcher_destroy(frt_matcher *m);
/* MatMul::init (src/matmul.cpp:9-34): gallery A[num_row][num_col] row-major fp32, copied to the device.  Idempotent:
 * a second call frees the previous device copy (the reference leaks it on every /reload). */
int frt_matcher_init(frt_matcher *m, const float *gallery, int num_row, int num_col);
/* Storage precision of the gallery rows for the NEXT frt_matcher_init / frt_matcher_gallery_begin: 0 = fp32 rows (default; the
 * reference's MatMul), 1 = rows stored as fp16 on the device (BASELINE config 5, "fp16 embeddings": half the HBM bytes per scan
 * and per GPU).  With fp16 storage the similarities are DEFINED as the fp32 dot products with the fp16-rounded rows
 * (sum_k e[k] * float(half(g[k])), fp32 fmaf chain) - calculate / top1 / the pipeline all agree on them bit for bit. */
int frt_matcher_set_storage(frt_matcher *m, int fp16);
/* Screened top-1 on / off (default on).  Galleries of >= 32 768 rows answer top-1 calls with a coarse scan of a compact shadow copy
 * (int8 rows with a per-row scale for 512-column fp32 galleries, fp16 otherwise) followed by an EXACT fp32 re-rank of every row the
 * rigorous error bound cannot exclude - the result is bit-identical to the exact scan, its cost depends on the queries (a query that
 * matches keeps one candidate block, one that matches nothing a dozen).  on = 0 makes every call take the exact fp32 scan of the
 * whole gallery (MatMul::calculate's arithmetic, src/matmul.h:7-16, with the first-maximum epilogue of src/arcface.cpp:203-217 fused):
 * 4 * num_col * num_row bytes per call whatever the queries - the path's worst case, reported by bench.py as `match_worst_case`. */
int frt_matcher_set_screening(frt_matcher *m, int on);
/* A counter that changes whenever the gallery this matcher answers from changes (init / commit / row offset; also when its scratch buffers
 * move).  A caller that keeps (row index, similarity) pairs across calls compares it to know whether the indices still name the same rows
 * (include/frt/arcface.h: the coalesced fast path of featureMatching / getOutputs). */
unsigned frt_matcher_generation(frt_matcher *m);
/* Bytes of gallery data ONE top-1 call reads in the current mode (the coarse scan's shadow copy when screening is on and the gallery is
 * large enough for it, the stored rows otherwise; the exact re-rank's candidate rows come on top and depend on the queries). */
size_t frt_matcher_scan_bytes(frt_matcher *m);
/* Streaming gallery load == the loop of Database::getEmbeddings (src/db.cpp:316-346):
 *   initKnownEmbeds(n)            -> frt_matcher_gallery_begin(m, n, 512)
 *   addEmbedding(id, blob) x n    -> frt_matcher_gallery_append(m, blob, 1)     (blob = sqlite3_column_blob: raw little-endian
 *                                    float32[num_col]; any number of consecutive rows per call; copied before the call returns)
 *   initMatMul()                  -> frt_matcher_gallery_commit(m)
 * Rows are staged in pinned host chunks and uploaded asynchronously while the caller fetches the next ones; the previous gallery
 * stays searchable until commit swaps it (and frees it: the reference leaks both copies on every /reload).  commit with fewer
 * rows than reserved is fine (the count actually appended becomes num_row); appending more is FRT_ERR_CAPACITY. */
int frt_matcher_gallery_begin(frt_matcher *m, int row_capacity, int num_col);
int frt_matcher_gallery_append(frt_matcher *m, const void *rows, int n_rows);
int frt_matcher_gallery_commit(frt_matcher *m);
int frt_matcher_num_rows(const frt_matcher *m);
/* MatMul::calculate (src/matmul.cpp:36-77): outputs[i*num_row + j] = sum_k embeds[i][k] * gallery[j][k], fp32. */
int frt_matcher_calculate(frt_matcher *m, const float *embeds, int embed_count, float *outputs);
/* calculate and the first-maximum argmax of every row in ONE call: outputs (may be NULL: no matrix is materialised) as
 * frt_matcher_calculate, idx_out / sim_out as frt_matcher_top1 - the same accumulators, so idx_out[i] / sim_out[i] ARE std::max_element
 * over outputs row i, bit for bit.  The drop-in ArcFaceIR50 shell uses it so that featureMatching() + getOutputs() (src/arcface.cpp:189-217)
 * keep their signatures while the host-side O(F*N) scan disappears; with a pinned `outputs` (frt_pinned_alloc) the [F x N] copy is one DMA. */
int frt_matcher_calculate_top1(frt_matcher *m, const float *embeds, int embed_count, float *outputs, int32_t *idx_out, float *sim_out);
/* Page-locked host memory for buffers that cross PCIe on every call (the [F x N] matrix MatMul::calculate hands back is 4 MB per face at
 * N = 1M: pageable memory makes that copy a staged, synchronous one).  device: the device whose context registers it. */
int frt_pinned_alloc(size_t bytes, int device, void **out);
void frt_pinned_free(void *p);
/* Fused calculate + getOutputs argmax: idx_out[i] = FIRST j maximising the similarity (std::max_element semantics),
 * sim_out[i] = that similarity.  Never materialises the [n x num_row] matrix. */
int frt_matcher_top1(frt_matcher *m, const float *embeds, int embed_count, int32_t *idx_out, float *sim_out);
/* frt_matcher_top1 with everything resident in HBM, asynchronous on hip_stream (a hipStream_t as void*, NULL = default stream):
 * queries [n][num_col] fp32, idx_dev int32 [n], sim_dev fp32 [n].  The sharded-gallery path (dist.py) feeds it the RCCL
 * all-gathered embeddings without a host round trip. */
int frt_matcher_top1_dev(frt_matcher *m, const void *embeds_dev, int embed_count, void *idx_dev, void *sim_dev, void *hip_stream);
/* Sharded galleries: declare that local row 0 of this matcher is global row `row_offset`; top-1
 * indices (frt_matcher_top1, the pipeline's match_idx) are then global.  The full matrix of calculate() stays local. */
int frt_matcher_set_row_offset(frt_matcher *m, int row_offset);
/* Sharded-gallery variant: rows of this matcher are global rows [row_offset, row_offset+num_row).
 * Merges with an existing (idx, sim) pair per query, keeping the higher similarity and the LOWER global index on ties. */
int frt_merge_top1(int n, const int32_t *idx_a, const float *sim_a, const int32_t *idx_b, const float *sim_b, int32_t *idx_out,
                   float *sim_out);

/* Exact top-k (BASELINE configs[4]: "fp16 embeddings, RCCL top-k all-gather"; the reference itself only ever takes the first maximum,
 * src/arcface.cpp:203-217, which is k = 1).  idx_out / sim_out are [embed_count][k]: entry j of a query is the j-th row of its exact
 * ranking - higher similarity first, LOWER (global) index first among equal similarities, i.e. entry 0 is frt_matcher_top1's answer bit
 * for bit and entry j is "std::max_element over the rows not yet taken".  Fewer than k rows: idx -1, sim -inf in the unused slots.
 * 1 <= k <= 16. */
int frt_matcher_topk(frt_matcher *m, const float *embeds, int embed_count, int k, int32_t *idx_out, float *sim_out);
/* Device-resident form, asynchronous on hip_stream.  embeds_fp16 = 0: queries fp32 [n][num_col]; 1: queries IEEE fp16 [n][num_col] - what
 * the embedding exchange of configs[4] delivers (frt_embeds_to_half_dev + all-gather); they are widened exactly, so the similarities are
 * the fp32 dot products of the fp16-rounded embeddings with the stored rows. */
int frt_matcher_topk_dev(frt_matcher *m, const void *embeds_dev, int embeds_fp16, int embed_count, int k, void *idx_dev, void *sim_dev, void *hip_stream);
/* k-way merge of per-shard top-k lists with GLOBAL indices (frt_matcher_set_row_offset), idx_all / sim_all [shards][n][k] -> [n][k]:
 * same order rule as above (the lower global index wins a tie - the first-maximum rule carried across shards); idx < 0 = empty slot. */
int frt_merge_topk(int shards, int n, int k, const int32_t *idx_all, const float *sim_all, int32_t *idx_out, float *sim_out);
int frt_merge_topk_dev(int shards, int n, int k, const void *idx_all_dev, const void *sim_all_dev, void *idx_out_dev, void *sim_out_dev, void *hip_stream);
/* fp32 -> IEEE fp16 (round to nearest even) of n_values (a multiple of 8) device floats: the embeddings before they are exchanged. */
int frt_embeds_to_half_dev(const void *embeds_dev, size_t n_values, void *half_out_dev, void *hip_stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-GPU exchange.  The reference is one single-GPU C++ process (src/app.cpp:52-57, 367); north_star shards whole
 * frames over the GPUs of a node with an RCCL all-gather as the only exchange step.  These calls are that step for a C++ host: RCCL
 * (bound at run time from librccl.so.1) behind plain pointers.  One communicator per device - one process per GPU (get_unique_id on
 * rank 0, hand the 128 bytes to the other ranks by any means, frt_comm_create everywhere), or one process driving several devices
 * (frt_comm_create_all + one thread per device, or frt_comm_all_gather_multi from one thread).  Create communicators AFTER the
 * pipelines of the device (they own a stream; see frt_pipeline_check_overlap).
 * ------------------------------------------------------------------------------------------------------------------ */
#define FRT_COMM_ID_BYTES 128
typedef struct frt_comm frt_comm;
int frt_comm_get_unique_id(uint8_t *id_out /* [FRT_COMM_ID_BYTES] */);
/* RCCL's bootstrap calls (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll) block without a timeout of their own; frt_comm_get_unique_id,
 * frt_comm_create and frt_comm_create_all give up after this many seconds and return FRT_ERR_DEVICE with a message that says what to look at
 * (a rank that never called in, the interface RCCL chose).  Default 180 s (env FRT_COMM_BOOTSTRAP_TIMEOUT_S); <= 0: wait for ever.  Process-wide;
 * returns the previous value. */
double frt_comm_set_bootstrap_timeout(double seconds);
int frt_comm_create(const uint8_t *id, int rank, int world, int device, frt_comm **out);
int frt_comm_create_all(int n_devices, const int *devices, frt_comm **out /* [n_devices] */);
void frt_comm_destroy(frt_comm *c);
int frt_comm_rank(const frt_comm *c);
int frt_comm_world(const frt_comm *c);
/* the communicator's own exchange stream (a hipStream_t as void*) */
void *frt_comm_stream(frt_comm *c);
/* recv_dev [world][bytes_per_rank] <- every rank's send_dev [bytes_per_rank]; asynchronous on hip_stream (NULL: the communicator's
 * stream).  What travels: frt_face_result records (configs[3]), fp16 embeddings and (idx, sim) top-k lists (configs[4]). */
int frt_comm_all_gather(frt_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *hip_stream);
/* the same collective for n communicators of ONE process issued from one thread (ncclGroupStart / End around the n calls) */
int frt_comm_all_gather_multi(int n, frt_comm *const *comms, const void *const *send_dev, void *const *recv_dev, size_t bytes_per_rank,
                              void *const *hip_streams);
int frt_comm_sync(frt_comm *c);

/* ------------------------------------------------------------------------------------------------------------------
 * Batched device-resident pipeline (new surface; the /inference call stack of src/app.cpp:304-310 for B frames at once,
 * without host round trips between the stages).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct frt_face_result {
    frt_bbox box;
    int32_t frame;     /* frame index inside the batch                          */
    int32_t match_idx; /* gallery row of the best match (-1: no gallery)        */
    float match_sim;   /* its cosine similarity                                 */
    int32_t valid;     /* 0: slot unused (fewer than max_faces boxes: box is all zeros, score 0) or empty ROI (box kept, score > 0) */
} frt_face_result;

/* Borrows the three objects (they must outlive the pipeline and live on the same device).  max_frames <= detector
 * max_batch.  Face slots are [frame][max_faces]. */
int frt_pipeline_create(frt_detector *d, frt_embedder *e, frt_matcher *m, int max_frames, frt_pipeline **out);
void frt_pipeline_destroy(frt_pipeline *p);
/* frames: host u8 BGR, n_frames contiguous frames.  results[n_frames*max_faces]; embeds_out (may be NULL)
 * [n_frames*max_faces][512]. */
int frt_pipeline_run(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out);
/* Asynchronous form of frt_pipeline_run for callers that keep several batches in flight (the reference's shell is synchronous,
 * app.cpp:293-352; this is the entry point a multi-threaded or double-buffered caller binds instead).  submit() queues the
 * H2D copy of `frames` on the pipeline's copy stream, the stages, and the D2H copies of `results` / `embeds_out`, and returns a
 * ticket; wait(ticket) blocks until that batch's outputs are in host memory.  All three host buffers must stay valid and
 * untouched until wait() returns, and should be pinned (hipHostMalloc / hipHostRegister): with pageable memory the copies
 * degrade to synchronous ones.  Up to 4 batches may be in flight; a 5th submit() first waits for the oldest.  Tickets complete
 * in order.  frt_pipeline_run (below the same path: submit + wait) may be called from several threads at once. */
int frt_pipeline_submit(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out, long *ticket_out);
int frt_pipeline_wait(frt_pipeline *p, long ticket);
/* frt_pipeline_submit that also returns the faces' u8 BGR 112x112 crops (what CroppedFace.face holds after ArcFaceIR50::forward,
 * src/arcface.cpp:3-17; the reply step JPEG-encodes one of them, src/app.cpp:328): crops_out [n_frames*max_faces][112][112][3], may be
 * NULL.  Crops of unused / empty-ROI slots are unspecified. */
int frt_pipeline_submit_crops(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out, uint8_t *crops_out,
                              long *ticket_out);
/* Same with everything resident in HBM: frames_dev u8 [n_frames][rows][cols][3]; results_dev / embeds_dev device
 * buffers (embeds_dev may be NULL).  Asynchronous on the pipeline stream; frt_pipeline_sync() waits. */
int frt_pipeline_run_dev(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev);
/* Same, ordered behind the caller's producer: ready_event is a hipEvent_t (as void*) the caller recorded after the work that
 * writes frames_dev (an upload, frt_jpeg_decode_batch_dev, frt_resize_frames_dev ... on ANY stream); the stages wait for it on the
 * device.  This is the form to use when the frames are produced asynchronously: with software pipelining on, plain
 * frt_pipeline_run_dev does not wait for earlier work on the pipeline stream (see frt_pipeline_set_overlap). */
int frt_pipeline_run_dev_after(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev, void *ready_event);
/* Safe mode for callers that produce the frames on the pipeline stream itself: every frt_pipeline_run_dev call first records an
 * event on that stream and the stages wait for it.  Correct by construction, but the pipeline stream also carries the joins of the
 * previous calls, so consecutive calls no longer overlap - prefer frt_pipeline_run_dev_after with a separate upload stream. */
int frt_pipeline_set_input_sync(frt_pipeline *p, int enable);
/* Stream-overlap self-check.  The three stages only overlap across calls when the pipeline's stage streams - and the caller's stream,
 * which carries the joins - sit on different hardware queues (ROCm maps streams round-robin onto 4 queues per priority level; a queue
 * runs in order).  This call MEASURES it: one 150 us single-wave probe kernel per stream, started together; *ratio_out = elapsed /
 * 150 us, about 1 when they run side by side, about n when n of them share a queue.  Above 1.5 the call still returns FRT_OK but
 * frt_last_error() holds a warning (also printed to stderr once unless FRT_QUIET is set) naming the remedy.  The same check runs over
 * the stage streams alone inside frt_pipeline_create (about 1 ms); call this one after
 * frt_pipeline_set_stream and after every other stream of the process (RCCL, copy streams) exists.  Waits for work in flight. */
int frt_pipeline_check_overlap(frt_pipeline *p, float *ratio_out);
int frt_pipeline_sync(frt_pipeline *p);
/* Run on a caller-owned HIP stream (a hipStream_t passed as void*, e.g. PyTorch's current stream, so that RCCL collectives
 * issued by the caller are ordered after the pipeline without a host synchronisation).  NULL restores the private stream. */
int frt_pipeline_set_stream(frt_pipeline *p, void *hip_stream);
/* Software pipelining across calls (default on): detector of call b+1, crop + recogniser of
 * call b and match + pack of call b-1 run on three internal streams; the pipeline stream joins at the end of every call, so
 * results stay ordered on it exactly as if the call had run there.  With overlap on, the frames passed to
 * frt_pipeline_run_dev must already be valid when the call is made (the internal streams do not wait for earlier work on
 * the pipeline stream) and must stay unchanged until that call's results are complete on the pipeline stream. */
int frt_pipeline_set_overlap(frt_pipeline *p, int enable);
/* hipGraph replay of a call's ~150 launches (opt-in through this call; measured neutral on one GPU).  A call whose buffers, batch size
 * and mode repeat is captured on its second occurrence and replayed afterwards; callers that never repeat their buffers stay
 * on eager launches.  Automatically off while frt_profile_enable() records events. */
int frt_pipeline_set_graph(frt_pipeline *p, int enable);
/* Pairing of consecutive calls (no reference counterpart - the reference answers one request at a time, src/app.cpp:243-287).
 * Below ~ 64 faces a recogniser pass is a chain of launch latencies, not work: 16 faces cost 0.59 ms, 32 faces 0.92 ms, and one match call
 * scans the gallery once whatever the number of queries.  The crop + recogniser + match stages of up to four CONSECUTIVE calls can run as
 * ONE pass: a call's detector stage is queued at the call as always; its later stages are queued together with a later call's.
 *
 * enable = -1 (DEFAULT since round 6): ADAPTIVE, frt_pipeline_submit calls only.  A call is held back only while the pipeline is backed up
 *   AND at least five earlier tickets are still running - the GPU then has work queued for longer than the held call waits (a caller that
 *   keeps two to four calls in flight is latency-coupled to each of them: any holding measured 3 - 20 % slower there, so it gets none).  Held
 *   submits are MERGED: the next submits' frames join the held ones in one staging set (up to four tickets / the pipeline's capacity) and run
 *   as ONE call - one detector pass, one recogniser pass, one match call, per-ticket result downloads (frt_pipeline_merge_stats); beyond that
 *   a call's recogniser pass may still wait for the next call's.  A call that finds the pipeline idle is queued at once: a lone caller sees
 *   exactly the unpaired pipeline and its latency.  Held calls are released by the submit that fills them or finds fewer than five tickets
 *   running, by frt_pipeline_wait on one of their tickets (or on any ticket once the backlog is gone), by frt_pipeline_run_dev,
 *   frt_pipeline_sync and any frt_pipeline_set_*; the submit / wait contract is unchanged.  frt_pipeline_run_dev calls are never held in
 *   this mode (their contract is the join on the pipeline stream AT the call).  -3: the same without the merging of submits.
 * enable = -2: adaptive for frt_pipeline_run_dev calls too.  What changes for such a caller: the pipeline stream joins a held call's results
 *   at a LATER call (or at frt_pipeline_sync), not at the call itself, and the frames must stay valid until then.
 * enable = 0: off.  enable = 1 or 2: ALWAYS pairs; 3, 4: always groups of three / four (both boundaries; results complete when the group
 *   is - up to three calls later - or at a flush; 64 faces per pass is where a pass stops being a latency chain).
 *
 * In every mode: boxes, validity and matched rows are the unpaired pipeline's; the embeddings come from the recogniser kernels chosen for
 * the faces of ALL the calls of a pass together (tile shapes follow the number of faces per pass) and agree with the unpaired ones to fp16
 * rounding (cosine >= 1 - 1e-5; the same relation as between any two batch sizes, tests/test_gpu_embedder.py, tests/test_gpu_pipeline.py).
 * A call is only held when it can be grouped: at least twice its face slots must fit max_frames * max_faces of the pipeline and the
 * recogniser's max_batch (a pipeline created for exactly the frames a call carries - the benchmark's 32-frame step - never groups).
 * frt_pipeline_pairing_stats counts the recogniser passes that served several calls and those that served one. */
int frt_pipeline_set_pairing(frt_pipeline *p, int enable);
int frt_pipeline_pairing_stats(frt_pipeline *p, long *paired_passes_out, long *single_passes_out);
/* Adaptive mode, host boundary: a frt_pipeline_submit that finds the DETECTOR still busy with earlier calls is not queued at once - its frames are
 * uploaded and held, the next submit's frames join them (up to 4 tickets / the pipeline's capacity), and the held frames run as ONE call: one
 * detector pass, one recogniser pass, one match call, per-ticket downloads (a 4-frame detector pass costs 258 us of kernels, a 32-frame one
 * 809).  Released like a held pass (above); a submit that finds the detector idle is never held.  Counts the calls that carried several
 * tickets and the tickets they carried. */
int frt_pipeline_merge_stats(frt_pipeline *p, long *merged_calls_out, long *merged_tickets_out);
/* hipGraph replay (frt_pipeline_set_graph): stage graphs captured / replayed so far (a key is captured on its second sighting). */
int frt_pipeline_graph_stats(frt_pipeline *p, long *captured_out, long *replayed_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Request coalescing (opt-in).  The reference answers ONE frame per request (src/app.cpp:293-352) from a Crow server that runs
 * .multithreaded() (src/app.cpp:367); a one-frame call is 4 faces and leaves the device idle.  A coalescer gathers the frames of
 * requests that are being served at the same time into one pipeline batch: every request thread calls frt_coalescer_infer with its
 * frame and blocks; frames that arrive while earlier batches run (or within window_us of the first one when the device is idle)
 * travel together through frt_pipeline_submit; each caller gets its own frame's results.  Borrows the three objects like
 * frt_pipeline_create (m may be NULL or empty: no match).  max_frames <= the detector's max_batch.  Thread-safe.
 * The C++ shells use it behind RetinaFace::findFace / ArcFaceIR50::forward / featureMatching when asked to
 * (RetinaFace::coalesceWith, or FRT_COALESCE=<frames> in the environment).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct frt_coalescer frt_coalescer;
int frt_coalescer_create(frt_detector *d, frt_embedder *e, frt_matcher *m, int max_frames, int window_us, frt_coalescer **out);
void frt_coalescer_destroy(frt_coalescer *c);
/* One frame of the detector's frame size in, its max_faces result slots out (frt_face_result, frame = 0; slots [0, *n_boxes) hold the
 * boxes findFace returns, in its order); embeds_out (may be NULL) [max_faces][512].  Blocks until the frame's batch has completed. */
int frt_coalescer_infer(frt_coalescer *c, const uint8_t *bgr, int rows, int cols, size_t row_stride, frt_face_result *results, float *embeds_out,
                        int *n_boxes);
/* Same, and the faces' u8 BGR 112x112 crops (crops_out [max_faces][112][112][3], may be NULL) - everything ArcFaceIR50::forward leaves behind. */
int frt_coalescer_infer_crops(frt_coalescer *c, const uint8_t *bgr, int rows, int cols, size_t row_stride, frt_face_result *results, float *embeds_out,
                              uint8_t *crops_out, int *n_boxes);
/* Batches submitted and frames carried so far (frames / batches = the mean batch the load produced). */
int frt_coalescer_stats(frt_coalescer *c, long *batches_out, long *frames_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Frame ingest: the caller's `cv::resize(img, img, Size(frameWidth, frameHeight))` (src/app.cpp:166,301;
 * default INTER_LINEAR, 8UC3) on the device, bit-identical to the CPU restatement of OpenCV's fixed-point path; and the JPEG decode
 * in front of it / the JPEG + base64 reply behind the hot path (below).
 * ------------------------------------------------------------------------------------------------------------------ */
/* host in, host out; out: out_rows x out_cols x 3, tight rows */
int frt_resize_frame(const uint8_t *bgr, int rows, int cols, size_t row_stride, uint8_t *out, int out_rows, int out_cols, int device);
/* device in, device out, asynchronous on hip_stream (null = default stream): n frames of rows x cols -> tight out_rows x out_cols,
 * e.g. straight into the buffer handed to frt_pipeline_run_dev */
int frt_resize_frames_dev(const void *src_dev, int n, int rows, int cols, size_t row_stride, size_t frame_stride, void *dst_dev,
                          int out_rows, int out_cols, void *hip_stream);

/* JPEG decode in front of the resize (src/app.cpp:296: cv::imdecode(byte_vector, IMREAD_UNCHANGED)) and the reply step behind the
 * hot path (src/app.cpp:328-340: cv::imencode(".jpg", best crop) + base64).  OpenCV hands both to libjpeg with its defaults (integer
 * "islow" DCT, triangle-filter chroma upsampling, quality 95, 4:2:0, Annex-K Huffman tables) - all-integer algorithms, reproduced
 * here bit for bit: Huffman coding on a pool of host threads (it is serial per scan), dequantisation + IDCT + upsampling + colour
 * conversion (+ the resize) and colour conversion + downsampling + FDCT + quantisation on the device.  Supported streams: baseline /
 * extended-sequential / progressive (spectral selection + successive approximation, any scan script) Huffman, 8 bit, grey or YCbCr
 * with 4:4:4 / 4:2:2 / 4:2:0 sampling, restart intervals; arithmetic-coded, lossless, RGB-coded and CMYK streams return
 * FRT_ERR_FORMAT.  Grey images come out with the value replicated to B, G and R. */
typedef struct frt_jpeg_decoder frt_jpeg_decoder;
/* header only; any of the out pointers may be NULL */
int frt_jpeg_info(const uint8_t *data, size_t size, int *width, int *height, int *components);
/* n_threads <= 0: one host thread per image up to the machine's core count (max 64).  max_images = JPEGs per decode_batch call. */
int frt_jpeg_decoder_create(int max_images, int max_width, int max_height, int n_threads, int device, frt_jpeg_decoder **out);
void frt_jpeg_decoder_destroy(frt_jpeg_decoder *d);
/* n JPEG byte strings -> frames_dev u8 BGR [n][out_h][out_w][3] (device; e.g. the buffer handed to frt_pipeline_run_dev_after).  Images
 * whose size differs from out_w x out_h are resized like frt_resize_frames_dev (cv::resize default INTER_LINEAR).  The call returns when
 * the entropy decoding is done and the device work is queued on hip_stream (NULL: the decoder's own stream); the JPEG bytes may be
 * released then, frames_dev is complete once hip_stream has run. */
int frt_jpeg_decode_batch_dev(frt_jpeg_decoder *d, const uint8_t *const *data, const size_t *sizes, int n, void *frames_dev, int out_h, int out_w,
                              void *hip_stream);
/* one image, host in / host out, at its own size: bgr_out [height][width][3] */
int frt_jpeg_decode(frt_jpeg_decoder *d, const uint8_t *data, size_t size, uint8_t *bgr_out, size_t capacity, int *width, int *height);
/* host half only (tests, tools): entropy-decoded, NOT dequantised coefficient blocks [total_blocks][64] in natural order (coef_out may
 * be NULL) and the geometry: int32[219] = width, height, ncomp, hmax, vmax, 3 x (h, v, blocks_w, blocks_h, samples_w, samples_h, first
 * block), total_blocks, 3 x 64 quantiser steps (natural order). */
int frt_jpeg_read_coefficients(const uint8_t *data, size_t size, int16_t *coef_out, size_t coef_capacity_blocks, int32_t *geometry_out);
/* Reply step: n equally sized u8 BGR images [n][rows][cols][3] (host pointer, or device pointer with device_input = 1, e.g. the crops
 * of frt_embedder_forward) -> n JFIF streams, slot i at out + i * out_stride, length out_sizes[i].  quality as cv::imencode's
 * IMWRITE_JPEG_QUALITY (the reference uses the default, 95). */
int frt_jpeg_encode_batch(frt_jpeg_decoder *d, const void *bgr, int device_input, int n, int rows, int cols, int quality, uint8_t *out, size_t out_stride,
                          size_t *out_sizes);
/* Same, ordered behind an asynchronous producer of a DEVICE input: ready_event is a hipEvent_t (as void*, NULL = none) the caller
 * recorded after the work that writes `bgr` (a pipeline stage, another stream's forward ...); the codec's stream waits for it on the
 * device.  frt_jpeg_encode_batch itself only orders against work the caller has already synchronised. */
int frt_jpeg_encode_batch_after(frt_jpeg_decoder *d, const void *bgr, int device_input, int n, int rows, int cols, int quality, uint8_t *out,
                                size_t out_stride, size_t *out_sizes, void *ready_event);
/* host half only: quantised blocks in zigzag order (all luma blocks [2*mcuy][2*mcux], then Cb, then Cr [mcuy][mcux]) -> JFIF stream */
int frt_jpeg_write_jfif(int quality, int width, int height, const int16_t *coef_zigzag, uint8_t *out, size_t capacity, size_t *size_out);
/* crow::utility::base64encode (src/app.cpp:331): standard alphabet with '=' padding, NUL-terminated.  Returns the string length, or
 * the capacity needed (length + 1) when out is NULL or too small. */
size_t frt_base64_encode(const uint8_t *data, size_t size, char *out, size_t capacity);

/* ------------------------------------------------------------------------------------------------------------------
 * Optional 5-point alignment mode (default OFF).  The reference has none: it trims the landmark head off the detector
 * (conversion/retina/torch2trt.py:7-9, src/retinaface.cpp:58-60) and feeds the embedder a bbox crop + bicubic resize
 * (src/arcface.cpp:3-17).  These entry points exist for accuracy work only and need a detector blob exported WITH
 * LandmarkHead.* (conversion/retina/models/retinaface.py:37-46,117); their oracle (oracle/align.py) is "parity
 * unpinned" - there is no reference behaviour to compare with.
 *   landmarks layout: 10 floats per face = (x0,y0,...,x4,y4), x = column, y = row in FRAME pixels, order
 *   left eye, right eye, nose, left mouth corner, right mouth corner (upstream RetinaFace order). */
int frt_detector_has_landmarks(const frt_detector *d);
/* findFace + landmarks of the kept boxes.  landmarks_out: capacity max_faces*10. */
int frt_detector_find_faces_landmarks(frt_detector *d, const uint8_t *bgr, int rows, int cols, size_t row_stride, frt_bbox *out,
                                      float *landmarks_out, int *n_out);
/* doInference with the third output: ldm_out [batch][A][10] raw head values. */
int frt_detector_infer_landmarks(frt_detector *d, const float *chw, int batch, float *loc_out, float *conf_out, float *ldm_out);
/* Counterpart of frt_crop_faces: least-squares similarity from the 5 landmarks to the ArcFace 112x112 template, bilinear
 * warp with zero border.  crops_out: n x 112 x 112 x 3 BGR u8.  FRT_ERR_EMPTY_ROI for degenerate (coincident) landmarks. */
int frt_align_faces(const uint8_t *bgr, int rows, int cols, size_t row_stride, const float *landmarks, int n, uint8_t *crops_out,
                    int device);
/* Counterpart of frt_embedder_forward with aligned crops instead of bbox crops. */
int frt_embedder_forward_aligned(frt_embedder *e, const uint8_t *bgr, int rows, int cols, size_t row_stride, const float *landmarks,
                                 int n, float *embeds_out, uint8_t *crops_out);
/* Switch a pipeline between the reference's crop (0, default) and the aligned crop (1). */
int frt_pipeline_set_align(frt_pipeline *p, int enable);

/* ------------------------------------------------------------------------------------------------------------------
 * Profiling hooks (HIP events on the library's own stream; used by bench.py for the roofline object).
 * ------------------------------------------------------------------------------------------------------------------ */
/* kinds: 0 = off (drops the records), 1 = time every launch of the dominant kernel family (conv3x3 MFMA), 2 = time every stage,
 * -1 = pause: stop recording but keep the records (bench.py samples ONE step of its timed region: the events around every conv
 * launch cost 11 % of a step when left on for all of them).  Every call with kind >= 0 also makes sure a pool of events exists, so that
 * no event is created between the two records of a bracket (call it with 0 once before a timed region). */
int frt_profile_enable(int kind);
/* Drains recorded events.  names_out: '\n'-separated labels; returns the number of records written (<= cap). */
int frt_profile_collect(char *names_out, size_t names_cap, double *ms_out, double *work_out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* FRT_H */
