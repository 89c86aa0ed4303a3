// struct frt_detector: the object behind frt_detector_* (include/frt.h).  Internal header of libfrt.so.
#pragma once
#include "frt_internal.hpp"

struct frt_detector {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // The pipeline's detector stage keeps running on the pipeline's stream after frt_pipeline_run_dev / submit returned (the
    // object mutex is only held while work is enqueued).  Object-level entry points share d_input, the activations and the
    // candidate buffers with it: they order their stream behind the end of the last such stage (one event wait).
    hipEvent_t ev_busy = nullptr;
    bool busy = false;
    void wait_idle(hipStream_t s) {
        if (busy) HIPCHK(hipStreamWaitEvent(s, ev_busy, 0));
    }
    Arena arena;
    DetGeom g{};
    int max_batch = 1;
    struct Op {
        int type;  // 0 dwpw, 1 conv3x3 (n same-shaped problems, one per pyramid level), 2 heads (n levels), 3 fused conv3x3 pair
        int n;
        DwPwArgs dw;
        Conv3Args c3[3];
        HeadArgs hd[3];
    };
    float *d_tmp = nullptr;  // depthwise intermediate of the split conv_dw path
    float *d_wave_zeros = nullptr;  // zeros for dwpw_wave_kernel (input rows outside the image)
    std::vector<Op> ops;
    double flops_per_frame = 0;
    uint8_t *d_frames = nullptr;
    float *d_input = nullptr, *d_loc = nullptr, *d_conf = nullptr;
    Candidate *d_cand = nullptr;
    int *d_cand_count = nullptr, *d_nout = nullptr;
    uint8_t *d_dead = nullptr;
    frt_bbox *d_boxes = nullptr;
    // optional alignment mode: present only when the blob carries the LandmarkHead (the reference trims it away)
    bool has_landmarks = false;
    float *d_ldm = nullptr;        // raw head output [B][A][10]
    int *d_kept_anchor = nullptr;  // [B][max_faces]
    float *d_landmarks = nullptr;  // decoded, frame coordinates [B][max_faces][10]

    void build(const frt::Blob &b);
    void forward(int n, hipStream_t s, int first_op = 0);  // d_input -> d_loc/d_conf (first_op = 1: op 0 already ran)
    // preprocess + forward; when the letterbox is the identity the first conv reads the u8 frames and d_input is never written
    void forward_frames(const uint8_t *frames_dev, int n, size_t row_stride, size_t frame_stride, hipStream_t s);
    void postprocess(int n, hipStream_t s, frt_bbox *boxes_out = nullptr, int *nout_out = nullptr, float *landmarks_out = nullptr);  // d_loc/d_conf -> boxes (default: d_boxes/d_nout/d_landmarks)
    void preprocess(const uint8_t *frames_dev, int n, size_t row_stride, size_t frame_stride, hipStream_t s);
};

