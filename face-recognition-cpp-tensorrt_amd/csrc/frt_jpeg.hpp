// Baseline JPEG codec, host half (frame ingest and reply step).
//
// The reference calls cv::imdecode(bytes, IMREAD_UNCHANGED) in front of the hot path (src/app.cpp:296) and cv::imencode(".jpg", crop) behind it
// (src/app.cpp:328); OpenCV 4.5.5 hands both to libjpeg(-turbo) with its defaults: JDCT_ISLOW, fancy upsampling, standard tables,
// quality 95, 4:2:0.  Those are all-integer algorithms (published in the Independent JPEG Group's jidctint / jfdctint / jdsample /
// jdcolor / jccolor / jcsample sources, restated here from their documented arithmetic), so the results below are defined bit for
// bit; tests pin them against PIL's libjpeg-turbo (tests/test_jpeg_*.py).
//
// Work split: entropy (Huffman) coding is serial per scan -> host threads, one image per task; everything per block / per pixel
// (dequantisation + IDCT + chroma upsampling + colour conversion, colour conversion + downsampling + FDCT + quantisation) -> device
// (kernels_jpeg.hip).  Coefficients cross PCIe as int16 blocks: the same 3 bytes per pixel at 4:2:0 as the raw BGR frame.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace frtjpeg {

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;       // blocks per row / column of the padded component plane (whole MCUs)
    int dw = 0, dh = 0;       // downsampled_width / height in samples: ceil(image * h / hmax)
    size_t block0 = 0;        // first block of this component in the coefficient buffer
};

struct Header {
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart_interval = 0;
    bool progressive = false;
    Component c[3];
    uint16_t q[4][64];        // quantisation tables, NATURAL (row-major) order
    bool qset[4] = {false, false, false, false};
    size_t total_blocks = 0;
    size_t scan_begin = 0;    // byte offset of the entropy-coded segment (sequential streams)
    size_t first_sos = 0;     // byte offset of the first SOS segment's length field (progressive streams: the scan loop starts here)
};

// 0 on success; error text in `err`.  Parses every marker segment up to the first SOS, including the Huffman tables.
struct HuffTable {
    bool set = false;
    uint8_t bits[17] = {0}, vals[256] = {0};
    // decoder view
    uint16_t fast[512];       // 9-bit lookup: (length << 8) | symbol, 0 = longer than 9 bits
    int32_t maxcode[18];
    int32_t valoff[17];
    void build();
};

struct Parsed {
    Header h;
    HuffTable dc[4], ac[4];
};

int parse(const uint8_t *data, size_t size, Parsed &out, std::string &err);
// Entropy-decodes the (single, interleaved or one-component) baseline scan into coef[total_blocks][64] (int16, natural order,
// NOT dequantised).  coef must be zero-filled by the caller.
int decode_scan(const uint8_t *data, size_t size, const Parsed &p, int16_t *coef, std::string &err);
// Progressive streams (SOF2; cv::imdecode accepts them, src/app.cpp:296): every scan of the file - DC first / refinement, AC first /
// refinement with end-of-band runs, spectral selection and successive approximation (ITU T.81 Annex G), Huffman tables redefined between
// scans, restart intervals - accumulated into the same coef[total_blocks][64] layout decode_scan produces; everything behind the entropy
// decoder (dequantisation, IDCT, upsampling, colour) is shared with the sequential path.  `p` is updated with the tables the scans define.
int decode_progressive(const uint8_t *data, size_t size, Parsed &p, int16_t *coef, std::string &err);
// either of the two, by p.h.progressive
int decode_coefficients(const uint8_t *data, size_t size, Parsed &p, int16_t *coef, std::string &err);

// ------------------------------------------------------------------------------------------------------------ encoder
struct EncTables {
    uint16_t q[2][64];        // natural order, quality-scaled, baseline-clamped
    // Huffman code / length per symbol for DC0, AC0, DC1, AC1
    uint16_t code[4][256];
    uint8_t len[4][256];
};
void make_enc_tables(int quality, EncTables &t);
// Writes SOI, JFIF APP0, DQT x2, SOF0 (3 components, 2x2 / 1x1 / 1x1), DHT x4, SOS, the entropy-coded scan and EOI.
// coef: quantised coefficients in ZIGZAG order; Y blocks [by][bx] of the padded luma plane first (bw = 2*mcux), then Cb, then Cr.
void write_jfif_420(const EncTables &t, int width, int height, const int16_t *coef, std::vector<uint8_t> &out);

std::string base64(const uint8_t *data, size_t n);

extern const uint8_t kNaturalOrder[64 + 16];  // zigzag index -> natural index (padded like libjpeg's table)

}  // namespace frtjpeg
