// Drop-in for /root/reference/src/matmul.h: class MatMul with the same public surface (src/matmul.h:17-21).
//   C = B x A^T : A = knownEmbeds [numRow x numCol] row-major (resident on the GPU), B = embeds [embedCount x numCol],
//   outputs[i*numRow + j] = <B_i, A_j>   (src/matmul.h:7-16)
#ifndef FRT_MATMUL_H
#define FRT_MATMUL_H

#include "common.h"

class MatMul {
  public:
    MatMul() : h_(nullptr), device_(0) {}
    explicit MatMul(int device) : h_(nullptr), device_(device) {}
    ~MatMul() { frt_matcher_destroy(h_); }
    MatMul(const MatMul &) = delete;
    MatMul &operator=(const MatMul &) = delete;

    void init(float *knownEmbeds, int numRow, int numCol) {
        ensure();
        checkFrtStatus(frt_matcher_init(h_, knownEmbeds, numRow, numCol));
    }
    void calculate(float *embeds, int embedCount, float *outputs) {
        ensure();
        checkFrtStatus(frt_matcher_calculate(h_, embeds, embedCount, outputs));
    }
    // Extension: calculate + the row-wise first maximum (what getOutputs computes from the matrix) in one call; outputs may be null
    void calculateTop1(float *embeds, int embedCount, float *outputs, int *idx, float *sim) {
        ensure();
        checkFrtStatus(frt_matcher_calculate_top1(h_, embeds, embedCount, outputs, idx, sim));
    }
    int device() const { return device_; }
    // Extension: streaming load (== initKnownEmbeds / addEmbedding x n / initMatMul without a host copy of the gallery): rows go
    // through pinned staging chunks to the device while the caller fetches the next ones (src/db.cpp:316-346)
    void galleryBegin(int rowCapacity, int numCol) {
        ensure();
        checkFrtStatus(frt_matcher_gallery_begin(h_, rowCapacity, numCol));
    }
    void galleryAppend(const float *rows, int n) { checkFrtStatus(frt_matcher_gallery_append(h_, rows, n)); }
    void galleryCommit() { checkFrtStatus(frt_matcher_gallery_commit(h_)); }
    int numRows() const { return frt_matcher_num_rows(h_); }
    // Extension: store the gallery rows as fp16 on the device (next init / galleryBegin); see frt_matcher_set_storage
    void setStorageFp16(bool on) {
        ensure();
        checkFrtStatus(frt_matcher_set_storage(h_, on ? 1 : 0));
    }
    // Extension: exact fp32 scan on every top-1 call instead of the screened one (same answers; see frt_matcher_set_screening)
    void setScreening(bool on) {
        ensure();
        checkFrtStatus(frt_matcher_set_screening(h_, on ? 1 : 0));
    }
    // Extension: fused argmax (what ArcFaceIR50::getOutputs computes from the full matrix), never materialises [n x numRow].
    void top1(float *embeds, int embedCount, int *idx, float *sim) {
        ensure();
        checkFrtStatus(frt_matcher_top1(h_, embeds, embedCount, idx, sim));
    }
    frt_matcher *handle() {
        ensure();
        return h_;
    }

  private:
    void ensure() {  // the reference constructs the cuBLASLt handle in the ctor; lazily here so a MatMul member costs nothing until used
        if (!h_) checkFrtStatus(frt_matcher_create(device_, &h_));
    }
    frt_matcher *h_;
    int device_;
};

using CosineSimilarityCalculator = MatMul;  // the name BASELINE.json:north_star uses for this class

#endif  // FRT_MATMUL_H
