// Device-side interface of the JPEG codec (kernels_jpeg.hip), shared with frt_jpeg_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct JpegImageDesc {   // one per image of a batch; lives in pinned host memory and is copied with the coefficients
    int width, height, ncomp, hmax, vmax;
    int h[3], v[3];        // sampling factors
    int bw[3], bh[3];      // blocks per row / column of the padded component plane
    int dw[3], dh[3];      // real (downsampled) plane size in samples
    int pitch[3];          // bytes per plane row = bw * 8
    uint32_t block0[3];    // first block of the component, relative to the image's first block
    uint64_t coef_block0;  // first block of the image in the batch's coefficient buffer
    uint64_t plane_off[3]; // byte offset of the component plane in the batch's plane buffer
    uint64_t out_off;      // byte offset of the image's tight BGR output
    uint16_t q[3][64];     // per-component quantisation table, natural order
};

void launch_jpeg_decode(const int16_t *coef, const JpegImageDesc *desc_dev, int n, int max_blocks_per_comp, int max_w, int max_h, uint8_t *planes,
                        uint8_t *out, hipStream_t s);
// n crops u8 BGR [rows][cols][3] (device) -> quantised zigzag blocks [n][6*mcux*mcuy][64]; q_dev: uint16 [2][64] natural order
void launch_jpeg_encode_blocks(const uint8_t *bgr, int n, int rows, int cols, const uint16_t *q_dev, int16_t *out, hipStream_t s);
