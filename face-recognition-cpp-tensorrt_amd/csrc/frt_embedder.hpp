// struct frt_embedder: the object behind frt_embedder_* (include/frt.h).  Internal header of libfrt.so.
#pragma once
#include "frt_internal.hpp"

struct ArcUnit {
    int cin, depth, stride, h_in;  // input spatial size (square)
    half_t *w1 = nullptr, *w2 = nullptr, *wsc = nullptr;
    half_t *w1f = nullptr, *w2f = nullptr;  // fragment-ordered copies for the strip kernel (stride-1 3x3 convs)
    half_t *w2f2 = nullptr;                 // ... for the stride-2 strip kernel (conv2 of the first unit of a stage)
    half_t *wscf = nullptr;                 // 1x1 shortcut weights in fragment order (the stride-2 strip kernel computes the shortcut conv itself)
    float *prelu = nullptr, *s2 = nullptr, *b2 = nullptr, *ssc = nullptr, *bsc = nullptr;
    float *s2f32 = nullptr;              // closing BatchNorm's scale WITHOUT the load-time conditioning factor (the fp32 path multiplies the blob's own weights)
    float *sn = nullptr, *bn = nullptr;  // BatchNorm that consumes this unit's output (next unit's leading BN / output_layer.0)
    float *se_w1 = nullptr, *se_w2 = nullptr;
};

struct frt_embedder {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // end of the last pipeline recogniser pass on each activation set (see frt_detector::wait_idle)
    hipEvent_t ev_busy[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    void wait_idle(hipStream_t s) {
        for (int i = 0; i < 2; ++i)
            if (busy[i]) HIPCHK(hipStreamWaitEvent(s, ev_busy[i], 0));
    }
    Arena arena;
    int max_batch = 1;
    bool se = false;
    std::vector<ArcUnit> units;
    float *in_w, *in_s0, *in_b0, *in_slope, *in_s1, *in_b1;
    half_t *in_wh = nullptr;
    half_t *wfc;
    float *fc_bias, *bn_s, *bn_b;
    // activations
    float *d_in = nullptr;  // [max_batch][3][112][112]
    half_t *Y[2], *Z[2], *T, *SC, *RES = nullptr, *zeros = nullptr;
    float *fc_partial, *d_out, *se_pool = nullptr, *se_gate = nullptr;
    int se_epoch = 0;  // launch counter of the fused SE tails (their gate-ready flags carry the launch number)
    bool sc_fusion = true;       // IR-50: 1x1 stride-2 shortcut convs inside the stride-2 strip kernel (tuning build: FRT_SC_FUSED=0 restores the launches)
    bool se_fused = true;        // IR-SE: run the SE tail inside conv2's epilogue where the strip kernels allow it (FRT_SE_FUSED=0 /
                                 // frt_embedder_set_se_fused(e, 0): always the stand-alone pool + gate + apply launches)
    int *h_se_error = nullptr;   // error word of the fused tail's cross-workgroup hand-over (pinned, mapped; 0 = fine)
    int *d_se_error = nullptr;   // ... its device address
    void check_se_error() {      // after a host synchronisation: a timed-out hand-over must not pass as a result
        if (h_se_error && *reinterpret_cast<volatile int *>(h_se_error) != 0) {
            *h_se_error = 0;
            raise(FRT_ERR_DEVICE, "IR-SE: the fused SE tail's cross-workgroup hand-over timed out (embeddings of that pass are invalid); "
                                  "frt_embedder_set_se_fused(e, 0) selects the stand-alone tail");
        }
    }
    uint8_t *d_crops = nullptr;
    int *d_valid = nullptr;
    frt_bbox *d_boxes = nullptr;
    float *d_lm = nullptr;  // landmark staging of forward_aligned [max_batch][10]
    uint8_t *d_frame = nullptr;
    size_t frame_cap = 0;
    static constexpr int FC_SPLITS = 49;
    double flops_per_face = 0;

    // ---- fp32 end-to-end mode (frt_embedder_set_precision(e, 1); kernels_arc_f32.hip): its own weights and activation buffers, built on
    //      first use from the blob the object was created from
    struct F32Unit {
        float *w1 = nullptr, *w2 = nullptr, *wsc = nullptr;  // [Cout][tap][Cin] fp32 in conv32_kernel's fragment order
    };
    struct F32 {
        std::vector<void *> owned;   // device allocations of this mode
        std::vector<F32Unit> units;
        float *wfc = nullptr;        // [512][hw * 512 + c]
        float *A[2] = {nullptr, nullptr}, *T = nullptr, *SCb = nullptr, *RES = nullptr, *gate = nullptr, *fc_out = nullptr;
        int chunk = 0;               // faces per pass of this path
        hipEvent_t done = nullptr;   // end of the last pass: one activation set, so passes on different streams run one after the other
        bool busy = false;
    } f32;
    bool fp32_mode = false;
    std::string blob_path;
    void build_f32();
    void forward_f32(const float *chw_dev, int F, const int *valid_dev, float *out_dev, hipStream_t s);

    void build(const frt::Blob &b);
    // chw_dev [F][3][112][112] -> out_dev [F][512]; F <= max_batch
    void forward(const float *chw_dev, int F, const int *valid_dev, float *out_dev, hipStream_t s);
    // second set of activation buffers: lets the pipeline run the recogniser passes of two consecutive calls concurrently on two
    // streams (forward_alt).  Allocated on demand (288 GB of HBM: 1.2 GB more is not a concern).
    struct ActSet {
        half_t *Y[2], *Z[2], *T, *SC, *RES;
        float *fc_partial, *se_pool, *se_gate;
    } alt{};
    bool has_alt = false;
    void ensure_alt();
    void swap_alt() {
        std::swap(Y[0], alt.Y[0]); std::swap(Y[1], alt.Y[1]); std::swap(Z[0], alt.Z[0]); std::swap(Z[1], alt.Z[1]);
        std::swap(T, alt.T); std::swap(SC, alt.SC); std::swap(RES, alt.RES);
        std::swap(fc_partial, alt.fc_partial); std::swap(se_pool, alt.se_pool); std::swap(se_gate, alt.se_gate);
    }
    void forward_set(int set, const float *chw_dev, int F, const int *valid_dev, float *out_dev, hipStream_t s) {
        if (set) swap_alt();  // host-side pointer swap: the launches below capture the alternate buffers
        try {
            forward(chw_dev, F, valid_dev, out_dev, s);
        } catch (...) {
            if (set) swap_alt();
            throw;
        }
        if (set) swap_alt();
    }
};

