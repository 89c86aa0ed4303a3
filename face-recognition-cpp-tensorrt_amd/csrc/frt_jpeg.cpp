// Baseline JPEG: marker parsing, Huffman entropy decoding / encoding, JFIF writer, base64 (host half; see frt_jpeg.hpp).
#include "frt_jpeg.hpp"

#include <cstring>

namespace frtjpeg {

const uint8_t kNaturalOrder[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                        6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                        39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

// ---------------------------------------------------------------------------------------------------------------- tables
void HuffTable::build() {
    // canonical codes (ITU T.81 Annex C): codes of length l are consecutive, starting at (previous first code + count) << 1
    int code = 0, k = 0;
    for (int i = 0; i < 512; ++i) fast[i] = 0;
    for (int l = 1; l <= 16; ++l) {
        valoff[l] = k - code;
        for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
            if (l <= 9) {
                const int first = code << (9 - l), n = 1 << (9 - l);
                for (int j = 0; j < n; ++j) fast[first + j] = (uint16_t)((l << 8) | vals[k]);
            }
        }
        maxcode[l] = bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
}

namespace {

inline int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

int fail(std::string &err, const char *m) {
    err = m;
    return 3;
}

}  // namespace

int parse(const uint8_t *d, size_t n, Parsed &out, std::string &err) {
    Header &h = out.h;
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(err, "jpeg: no SOI marker");
    size_t p = 2;
    bool have_sof = false, saw_jfif = false, saw_adobe = false;
    int adobe_transform = -1;
    while (true) {
        if (p + 4 > n) return fail(err, "jpeg: truncated before SOS");
        if (d[p] != 0xFF) return fail(err, "jpeg: marker expected");
        while (p < n && d[p] == 0xFF) ++p;  // fill bytes
        if (p >= n) return fail(err, "jpeg: truncated");
        const int m = d[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;  // parameterless
        if (m == 0xD9) return fail(err, "jpeg: EOI before any scan");
        if (p + 2 > n) return fail(err, "jpeg: truncated segment");
        const int len = be16(d + p);
        if (len < 2 || p + len > n) return fail(err, "jpeg: bad segment length");
        const uint8_t *s = d + p + 2;
        const int sl = len - 2;
        if (m == 0xDB) {  // DQT
            int q = 0;
            while (q < sl) {
                const int pq = s[q] >> 4, tq = s[q] & 15;
                if (tq > 3 || pq > 1) return fail(err, "jpeg: bad DQT");
                const int need = 1 + 64 * (pq + 1);
                if (q + need > sl) return fail(err, "jpeg: short DQT");
                for (int i = 0; i < 64; ++i) h.q[tq][kNaturalOrder[i]] = pq ? (uint16_t)be16(s + q + 1 + 2 * i) : s[q + 1 + i];
                h.qset[tq] = true;
                q += need;
            }
        } else if (m == 0xC4) {  // DHT
            int q = 0;
            while (q < sl) {
                if (q + 17 > sl) return fail(err, "jpeg: short DHT");
                const int tc = s[q] >> 4, th = s[q] & 15;
                if (tc > 1 || th > 3) return fail(err, "jpeg: bad DHT");
                HuffTable &t = tc ? out.ac[th] : out.dc[th];
                int total = 0;
                t.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) {
                    t.bits[i] = s[q + i];
                    total += t.bits[i];
                }
                if (total > 256 || q + 17 + total > sl) return fail(err, "jpeg: bad DHT counts");
                // Kraft limit (libjpeg's "bad Huffman table"): with `code` = first unused code of length l, more than 2^l codes of
                // that length cannot exist.  An over-subscribed table would make build() index past fast[512] / hand out codes
                // that overlap - reject it before it is ever built.
                for (int l = 1, code = 0; l <= 16; ++l) {
                    code += t.bits[l];
                    if (code > (1 << l)) return fail(err, "jpeg: bad Huffman table (over-subscribed code lengths)");
                    code <<= 1;
                }
                std::memcpy(t.vals, s + q + 17, (size_t)total);
                t.build();
                t.set = true;
                q += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {  // SOF0 baseline, SOF1 extended sequential (Huffman), SOF2 progressive
            if (have_sof) return fail(err, "jpeg: two SOF markers");
            if (sl < 6) return fail(err, "jpeg: short SOF");
            if (s[0] != 8) return fail(err, "jpeg: only 8-bit precision is supported");
            h.progressive = m == 0xC2;
            h.height = be16(s + 1);
            h.width = be16(s + 3);
            h.ncomp = s[5];
            if (h.width < 1 || h.height < 1) return fail(err, "jpeg: empty image");
            if (h.ncomp != 1 && h.ncomp != 3) return fail(err, "jpeg: only 1- and 3-component images are supported");
            if (sl < 6 + 3 * h.ncomp) return fail(err, "jpeg: short SOF");
            for (int i = 0; i < h.ncomp; ++i) {
                Component &c = h.c[i];
                c.id = s[6 + 3 * i];
                c.h = s[7 + 3 * i] >> 4;
                c.v = s[7 + 3 * i] & 15;
                c.tq = s[8 + 3 * i];
                if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return fail(err, "jpeg: unsupported sampling factors");
            }
            have_sof = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return fail(err, "jpeg: lossless / arithmetic-coded / hierarchical streams are not supported");
        } else if (m == 0xE0) {  // APP0: "JFIF\0" fixes the colour space to YCbCr (libjpeg: saw_JFIF_marker)
            if (sl >= 5 && !std::memcmp(s, "JFIF", 5)) saw_jfif = true;
        } else if (m == 0xEE) {  // APP14: "Adobe" + version(2) flags0(2) flags1(2) transform(1)
            if (sl >= 12 && !std::memcmp(s, "Adobe", 5)) {
                saw_adobe = true;
                adobe_transform = s[11];
            }
        } else if (m == 0xDD) {  // DRI
            if (sl < 2) return fail(err, "jpeg: short DRI");
            h.restart_interval = be16(s);
        } else if (m == 0xDA) {  // SOS
            if (!have_sof) return fail(err, "jpeg: SOS before SOF");
            if (h.progressive) {  // the scans are walked by decode_progressive; tables may still change between them
                for (int i = 0; i < h.ncomp; ++i)
                    if (!h.qset[h.c[i].tq]) return fail(err, "jpeg: component refers to a missing quantisation table");
                h.first_sos = p;
                break;
            }
            if (sl < 1 || s[0] != h.ncomp || sl < 1 + 2 * h.ncomp + 3) return fail(err, "jpeg: multi-scan sequential streams are not supported");
            for (int i = 0; i < h.ncomp; ++i) {
                const int cid = s[1 + 2 * i];
                Component *c = nullptr;
                for (int j = 0; j < h.ncomp; ++j)
                    if (h.c[j].id == cid) c = &h.c[j];
                if (!c || c != &h.c[i]) return fail(err, "jpeg: scan component order differs from the frame header");
                c->td = s[2 + 2 * i] >> 4;
                c->ta = s[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !out.dc[c->td].set || !out.ac[c->ta].set) return fail(err, "jpeg: scan refers to a missing Huffman table");
                if (!h.qset[c->tq]) return fail(err, "jpeg: component refers to a missing quantisation table");
            }
            h.scan_begin = p + len;
            break;
        }
        p += len;
    }
    // Colour space of a 3-component stream, decided the way libjpeg's default_decompress_parms does (cv::imdecode inherits it): JFIF
    // -> YCbCr; else Adobe transform 0 -> RGB, 1 -> YCbCr; else component ids 'R','G','B' -> RGB, anything else -> YCbCr.  The
    // device half only implements the YCbCr -> BGR conversion, so RGB-coded streams are refused instead of decoded with wrong colours.
    if (h.ncomp == 3 && !saw_jfif) {
        const bool rgb_ids = h.c[0].id == 'R' && h.c[1].id == 'G' && h.c[2].id == 'B';
        if ((saw_adobe && adobe_transform == 0) || (!saw_adobe && rgb_ids))
            return fail(err, "jpeg: RGB-coded streams (Adobe transform 0 / component ids R,G,B) are not supported");
    }
    // geometry
    Header &g = out.h;
    if (g.ncomp == 1) {  // a single-component scan is never interleaved: its sampling factors do not pad the plane
        g.c[0].h = g.c[0].v = 1;
    }
    g.hmax = g.vmax = 1;
    for (int i = 0; i < g.ncomp; ++i) {
        g.hmax = g.c[i].h > g.hmax ? g.c[i].h : g.hmax;
        g.vmax = g.c[i].v > g.vmax ? g.c[i].v : g.vmax;
    }
    if (g.ncomp == 3 && (g.c[1].h != 1 || g.c[1].v != 1 || g.c[2].h != 1 || g.c[2].v != 1))
        return fail(err, "jpeg: only chroma sampling 1x1 relative to luma 1x1 / 2x1 / 1x2 / 2x2 is supported");
    g.mcux = (g.width + 8 * g.hmax - 1) / (8 * g.hmax);
    g.mcuy = (g.height + 8 * g.vmax - 1) / (8 * g.vmax);
    size_t b = 0;
    for (int i = 0; i < g.ncomp; ++i) {
        Component &c = g.c[i];
        c.bw = g.mcux * c.h;
        c.bh = g.mcuy * c.v;
        c.dw = (g.width * c.h + g.hmax - 1) / g.hmax;
        c.dh = (g.height * c.v + g.vmax - 1) / g.vmax;
        c.block0 = b;
        b += (size_t)c.bw * c.bh;
    }
    g.total_blocks = b;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- entropy decoder
namespace {

struct BitReader {
    const uint8_t *d;
    size_t n, p;
    uint64_t acc = 0;   // bits are consumed from the top
    int cnt = 0;        // valid bits in acc
    int marker = 0;     // pending marker (0: none); once hit, zero bits are supplied
    int fed_zeros = 0;  // zero bytes supplied past the end of the data (a marker or the end of the buffer)
    // the segment has nothing left: more than a full bit buffer of made-up zeros has been handed out.  A scan that goes on from here only
    // decodes zeros; the walkers stop it (what has been decoded stands) so that a short crafted stream cannot buy whole-image passes
    bool dry() const { return fed_zeros > 16; }
    void refill() {
        while (cnt <= 56) {
            int b = 0;
            if (marker || p >= n) ++fed_zeros;
            if (!marker && p < n) {
                b = d[p];
                if (b == 0xFF) {
                    const int b2 = p + 1 < n ? d[p + 1] : 0xD9;
                    if (b2 == 0) {
                        p += 2;
                    } else {  // a marker ends the entropy-coded segment
                        marker = b2;
                        b = 0;
                    }
                } else {
                    ++p;
                }
            }
            acc |= (uint64_t)b << (56 - cnt);
            cnt += 8;
        }
    }
    inline int peek(int k) { return (int)(acc >> (64 - k)); }
    inline void skip(int k) {
        acc <<= k;
        cnt -= k;
    }
    inline int get(int k) {  // k in 1..16
        if (cnt < k) refill();
        const int v = peek(k);
        skip(k);
        return v;
    }
    // byte-align and consume an expected RSTn
    bool restart(int expect) {
        acc = 0;
        cnt = 0;
        fed_zeros = 0;
        if (!marker) {  // the marker has not been reached by the bit buffer yet: scan for it (skipping padding bits already consumed)
            while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] != 0 && d[p + 1] != 0xFF)) ++p;
            if (p + 1 >= n) return false;
            marker = d[p + 1];
        }
        if (marker != 0xD0 + expect) return false;
        p += 2;
        marker = 0;
        return true;
    }
};

inline int decode_sym(BitReader &br, const HuffTable &t) {
    if (br.cnt < 16) br.refill();
    const int f = t.fast[br.peek(9)];
    if (f) {
        br.skip(f >> 8);
        return f & 255;
    }
    int l = 10;
    int code = br.peek(10);
    while (l <= 16 && code > t.maxcode[l]) {
        ++l;
        code = br.peek(l);
    }
    if (l > 16) return -1;
    br.skip(l);
    return t.vals[code + t.valoff[l]];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
// DC predictor update, kept in the 16-bit range the coefficient has anyway (unsigned arithmetic: crafted streams cannot overflow an int)
inline int dc_add(int pred, int diff) { return (int)(int16_t)(uint16_t)((unsigned)pred + (unsigned)diff); }

}  // namespace

int decode_scan(const uint8_t *d, size_t n, const Parsed &P, int16_t *coef, std::string &err) {
    const Header &h = P.h;
    BitReader br{d, n, h.scan_begin};
    int pred[3] = {0, 0, 0};
    const int total_mcu = h.mcux * h.mcuy;
    int until_restart = h.restart_interval ? h.restart_interval : total_mcu + 1, rst = 0;
    for (int mcu = 0; mcu < total_mcu; ++mcu) {
        if (until_restart == 0) {
            if (!br.restart(rst)) return fail(err, "jpeg: missing restart marker");
            rst = (rst + 1) & 7;
            pred[0] = pred[1] = pred[2] = 0;
            until_restart = h.restart_interval;
        }
        --until_restart;
        const int my = mcu / h.mcux, mx = mcu - my * h.mcux;
        for (int ci = 0; ci < h.ncomp; ++ci) {
            const Component &c = h.c[ci];
            const HuffTable &dct = P.dc[c.td], &act = P.ac[c.ta];
            for (int v = 0; v < c.v; ++v)
                for (int hh = 0; hh < c.h; ++hh) {
                    int16_t *blk = coef + (c.block0 + (size_t)(my * c.v + v) * c.bw + (mx * c.h + hh)) * 64;
                    int s = decode_sym(br, dct);
                    if (s < 0 || s > 11) return fail(err, "jpeg: corrupt DC code");
                    if (s) pred[ci] = dc_add(pred[ci], extend(br.get(s), s));
                    blk[0] = (int16_t)pred[ci];
                    for (int k = 1; k < 64;) {
                        const int rs = decode_sym(br, act);
                        if (rs < 0) return fail(err, "jpeg: corrupt AC code");
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r != 15) break;  // EOB
                            k += 16;
                            continue;
                        }
                        k += r;
                        if (k > 63) return fail(err, "jpeg: AC run past the end of the block");
                        blk[kNaturalOrder[k]] = (int16_t)extend(br.get(sz), sz);
                        ++k;
                    }
                }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- progressive
namespace {

// one DHT segment body -> tables (shared with parse(): same validation)
int read_dht(const uint8_t *s, int sl, Parsed &out, std::string &err) {
    int q = 0;
    while (q < sl) {
        if (q + 17 > sl) return fail(err, "jpeg: short DHT");
        const int tc = s[q] >> 4, th = s[q] & 15;
        if (tc > 1 || th > 3) return fail(err, "jpeg: bad DHT");
        HuffTable &t = tc ? out.ac[th] : out.dc[th];
        int total = 0;
        t.bits[0] = 0;
        for (int i = 1; i <= 16; ++i) {
            t.bits[i] = s[q + i];
            total += t.bits[i];
        }
        if (total > 256 || q + 17 + total > sl) return fail(err, "jpeg: bad DHT counts");
        for (int l = 1, code = 0; l <= 16; ++l) {
            code += t.bits[l];
            if (code > (1 << l)) return fail(err, "jpeg: bad Huffman table (over-subscribed code lengths)");
            code <<= 1;
        }
        std::memcpy(t.vals, s + q + 17, (size_t)total);
        t.build();
        t.set = true;
        q += 17 + total;
    }
    return 0;
}

struct ScanComp {
    int ci, td, ta;
};

}  // namespace

int decode_progressive(const uint8_t *d, size_t n, Parsed &P, int16_t *coef, std::string &err) {
    Header &h = P.h;
    size_t p = h.first_sos;  // at the length field of the first SOS
    int m = 0xDA;
    int n_scans = 0;
    for (;;) {
        // ---- marker segment `m` with its length field at p
        if (m == 0xD9) break;  // EOI
        if (p + 2 > n) break;  // truncated file: what has been decoded so far stands (libjpeg: "premature end of data")
        const int len = be16(d + p);
        if (len < 2 || p + len > n) return fail(err, "jpeg: bad segment length");
        const uint8_t *s = d + p + 2;
        const int sl = len - 2;
        if (m == 0xC4) {
            if (read_dht(s, sl, P, err)) return 3;
        } else if (m == 0xDD) {
            if (sl < 2) return fail(err, "jpeg: short DRI");
            h.restart_interval = be16(s);
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            return fail(err, "jpeg: two SOF markers");
        } else if (m == 0xDA) {
            if (++n_scans > 4 * 64 * h.ncomp) return fail(err, "jpeg: too many scans");  // (a full spectral / successive-approximation script is far below)
            if (sl < 1) return fail(err, "jpeg: short SOS");
            const int ns = s[0];
            if (ns < 1 || ns > h.ncomp || sl < 1 + 2 * ns + 3) return fail(err, "jpeg: bad SOS");
            ScanComp sc[3];
            for (int i = 0; i < ns; ++i) {
                const int cid = s[1 + 2 * i];
                sc[i].ci = -1;
                for (int j = 0; j < h.ncomp; ++j)
                    if (h.c[j].id == cid) sc[i].ci = j;
                if (sc[i].ci < 0 || (i > 0 && sc[i].ci <= sc[i - 1].ci)) return fail(err, "jpeg: bad scan component");
                sc[i].td = s[2 + 2 * i] >> 4;
                sc[i].ta = s[2 + 2 * i] & 15;
                if (sc[i].td > 3 || sc[i].ta > 3) return fail(err, "jpeg: bad scan table index");
            }
            const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
            if (Ss > Se || Se > 63 || Al > 13 || (Ah != 0 && Ah != Al + 1)) return fail(err, "jpeg: bad progressive scan parameters");
            if (Ss == 0 && Se != 0) return fail(err, "jpeg: a progressive scan mixes DC and AC coefficients");
            if (Ss > 0 && ns != 1) return fail(err, "jpeg: an AC scan must hold one component");
            if (ns > 1 && ns != h.ncomp) return fail(err, "jpeg: interleaved scans over a subset of the components are not supported");
            for (int i = 0; i < ns; ++i) {
                if (Ss == 0 && Ah == 0 && !P.dc[sc[i].td].set) return fail(err, "jpeg: scan refers to a missing Huffman table");
                if (Ss > 0 && !P.ac[sc[i].ta].set) return fail(err, "jpeg: scan refers to a missing Huffman table");
            }
            // ---- the entropy-coded segment
            BitReader br{d, n, p + len};
            int pred[3] = {0, 0, 0};
            int eobrun = 0, rst = 0;
            const int p1 = 1 << Al, m1 = -(1 << Al);
            if (ns > 1 || (h.ncomp == 1 && Ss == 0)) {
                // interleaved DC scan (or the only component's DC scan): MCU order, component planes padded to whole MCUs
                const int total_mcu = h.mcux * h.mcuy;
                int until = h.restart_interval ? h.restart_interval : total_mcu + 1;
                for (int mcu = 0; mcu < total_mcu && !br.dry(); ++mcu) {
                    if (until == 0) {
                        if (!br.restart(rst)) return fail(err, "jpeg: missing restart marker");
                        rst = (rst + 1) & 7;
                        pred[0] = pred[1] = pred[2] = 0;
                        until = h.restart_interval;
                    }
                    --until;
                    const int my = mcu / h.mcux, mx = mcu - my * h.mcux;
                    for (int i = 0; i < ns; ++i) {
                        const Component &c = h.c[sc[i].ci];
                        for (int v = 0; v < c.v; ++v)
                            for (int hh = 0; hh < c.h; ++hh) {
                                int16_t *blk = coef + (c.block0 + (size_t)(my * c.v + v) * c.bw + (mx * c.h + hh)) * 64;
                                if (Ah == 0) {
                                    const int sz = decode_sym(br, P.dc[sc[i].td]);
                                    if (sz < 0 || sz > 11) return fail(err, "jpeg: corrupt DC code");
                                    if (sz) pred[i] = dc_add(pred[i], extend(br.get(sz), sz));
                                    blk[0] = (int16_t)((unsigned)pred[i] << Al);
                                } else if (br.get(1)) {
                                    blk[0] = (int16_t)(blk[0] | p1);
                                }
                            }
                    }
                }
            } else {
                // non-interleaved scan of one component: its own block raster, ceil(samples / 8) blocks per row / column
                const Component &c = h.c[sc[0].ci];
                const int bwn = (c.dw + 7) / 8, bhn = (c.dh + 7) / 8;
                const int total = bwn * bhn;
                int until = h.restart_interval ? h.restart_interval : total + 1;
                const HuffTable &act = P.ac[sc[0].ta];
                for (int b = 0; b < total && !br.dry(); ++b) {
                    if (until == 0) {
                        if (!br.restart(rst)) return fail(err, "jpeg: missing restart marker");
                        rst = (rst + 1) & 7;
                        pred[0] = 0;
                        eobrun = 0;
                        until = h.restart_interval;
                    }
                    --until;
                    const int by = b / bwn, bx = b - by * bwn;
                    int16_t *blk = coef + (c.block0 + (size_t)by * c.bw + bx) * 64;
                    if (Ss == 0) {  // DC scan of one component of a multi-component image
                        if (Ah == 0) {
                            const int sz = decode_sym(br, P.dc[sc[0].td]);
                            if (sz < 0 || sz > 11) return fail(err, "jpeg: corrupt DC code");
                            if (sz) pred[0] = dc_add(pred[0], extend(br.get(sz), sz));
                            blk[0] = (int16_t)((unsigned)pred[0] << Al);
                        } else if (br.get(1)) {
                            blk[0] = (int16_t)(blk[0] | p1);
                        }
                    } else if (Ah == 0) {  // AC first pass (G.2: decode_mcu_AC_first)
                        if (eobrun > 0) {
                            --eobrun;
                            continue;
                        }
                        for (int k = Ss; k <= Se;) {
                            const int rs = decode_sym(br, act);
                            if (rs < 0) return fail(err, "jpeg: corrupt AC code");
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz) {
                                k += r;
                                if (k > Se) return fail(err, "jpeg: AC run past the end of the band");
                                blk[kNaturalOrder[k]] = (int16_t)(extend(br.get(sz), sz) * (1 << Al));
                                ++k;
                            } else if (r == 15) {
                                k += 16;
                            } else {  // EOBr: this block and (1 << r) + extra - 1 following ones have nothing more in the band
                                eobrun = 1 << r;
                                if (r) eobrun += br.get(r);
                                --eobrun;
                                break;
                            }
                        }
                    } else {  // AC refinement (G.1.2.3: decode_mcu_AC_refine)
                        int k = Ss;
                        if (eobrun == 0) {
                            for (; k <= Se; ++k) {
                                const int rs = decode_sym(br, act);
                                if (rs < 0) return fail(err, "jpeg: corrupt AC code");
                                int r = rs >> 4;
                                const int sz = rs & 15;
                                int val = 0;
                                if (sz) {
                                    if (sz != 1) return fail(err, "jpeg: corrupt AC refinement code");
                                    val = br.get(1) ? p1 : m1;
                                } else if (r != 15) {
                                    eobrun = 1 << r;
                                    if (r) eobrun += br.get(r);
                                    break;  // the rest of the block is handled as end-of-band below
                                }
                                // skip r still-zero coefficients, refining the already non-zero ones that are passed
                                for (; k <= Se; ++k) {
                                    int16_t &cf = blk[kNaturalOrder[k]];
                                    if (cf != 0) {
                                        if (br.get(1) && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                                    } else if (--r < 0) {
                                        break;
                                    }
                                }
                                if (val && k <= Se) blk[kNaturalOrder[k]] = (int16_t)val;
                            }
                        }
                        if (eobrun > 0) {  // refine the remaining non-zero coefficients of the band
                            for (; k <= Se; ++k) {
                                int16_t &cf = blk[kNaturalOrder[k]];
                                if (cf != 0 && br.get(1) && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                            }
                            --eobrun;
                        }
                    }
                }
            }
            // ---- the next marker: where the bit reader met it, or further on if the buffer had not got there yet
            p = br.p;
            while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] != 0 && d[p + 1] != 0xFF && !(d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7))) ++p;
            if (p + 1 >= n) break;  // no EOI: accept what was decoded
            m = d[p + 1];
            p += 2;
            continue;
        }
        // ---- next marker behind a non-scan segment
        p += len;
        if (p + 2 > n) break;
        if (d[p] != 0xFF) return fail(err, "jpeg: marker expected");
        while (p < n && d[p] == 0xFF) ++p;
        if (p >= n) break;
        m = d[p++];
        while (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) {  // parameterless
            if (p + 2 > n || d[p] != 0xFF) return 0;
            while (p < n && d[p] == 0xFF) ++p;
            if (p >= n) return 0;
            m = d[p++];
        }
    }
    if (n_scans == 0) return fail(err, "jpeg: no scan");
    return 0;
}

int decode_coefficients(const uint8_t *d, size_t n, Parsed &P, int16_t *coef, std::string &err) {
    return P.h.progressive ? decode_progressive(d, n, P, coef, err) : decode_scan(d, n, P, coef, err);
}

// ---------------------------------------------------------------------------------------------------------------- encoder
namespace {

// ITU T.81 Annex K.3 typical Huffman tables (the ones jpeg_set_defaults installs)
const uint8_t kDcLumBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChrBits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
const uint8_t kAcChrBits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
// Annex K.1 / K.2 quantisation tables (natural order)
const uint8_t kStdLumQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,  69,  56,
                              14, 17, 22, 29, 51,  87,  80,  62,  18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                              49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kStdChrQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

struct HSpec {
    const uint8_t *bits, *vals;
    int n;
};
const HSpec kSpec[4] = {{kDcLumBits, kDcVals, 12}, {kAcLumBits, kAcLumVals, 162}, {kDcChrBits, kDcVals, 12}, {kAcChrBits, kAcChrVals, 162}};

struct BitWriter {
    std::vector<uint8_t> &o;
    uint32_t acc = 0;
    int cnt = 0;
    void put(unsigned code, int len) {
        acc = (acc << len) | (code & ((1u << len) - 1));
        cnt += len;
        while (cnt >= 8) {
            const uint8_t b = (uint8_t)(acc >> (cnt - 8));
            o.push_back(b);
            if (b == 0xFF) o.push_back(0);
            cnt -= 8;
        }
    }
    void flush() {
        if (cnt) put(0x7F, 8 - cnt);  // pad with one bits
    }
};

void seg(std::vector<uint8_t> &o, int marker, const std::vector<uint8_t> &body) {
    o.push_back(0xFF);
    o.push_back((uint8_t)marker);
    const int len = (int)body.size() + 2;
    o.push_back((uint8_t)(len >> 8));
    o.push_back((uint8_t)len);
    o.insert(o.end(), body.begin(), body.end());
}

inline int nbits(int v) {
    int n = 0;
    while (v) {
        ++n;
        v >>= 1;
    }
    return n;
}

}  // namespace

void make_enc_tables(int quality, EncTables &t) {
    // jpeg_quality_scaling + jpeg_add_quant_table(force_baseline = TRUE)
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 64; ++i) {
            long v = ((long)(k ? kStdChrQ[i] : kStdLumQ[i]) * scale + 50L) / 100L;
            if (v <= 0) v = 1;
            if (v > 255) v = 255;
            t.q[k][i] = (uint16_t)v;
        }
    for (int k = 0; k < 4; ++k) {
        std::memset(t.len[k], 0, 256);
        int code = 0, idx = 0;
        for (int l = 1; l <= 16; ++l) {
            for (int i = 0; i < kSpec[k].bits[l]; ++i, ++idx, ++code) {
                t.code[k][kSpec[k].vals[idx]] = (uint16_t)code;
                t.len[k][kSpec[k].vals[idx]] = (uint8_t)l;
            }
            code <<= 1;
        }
    }
}

void write_jfif_420(const EncTables &t, int width, int height, const int16_t *coef, std::vector<uint8_t> &o) {
    o.clear();
    o.push_back(0xFF);
    o.push_back(0xD8);
    seg(o, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});  // JFIF 1.01, aspect ratio 1:1, no thumbnail
    for (int k = 0; k < 2; ++k) {
        std::vector<uint8_t> b(65);
        b[0] = (uint8_t)k;
        for (int i = 0; i < 64; ++i) b[1 + i] = (uint8_t)t.q[k][kNaturalOrder[i]];
        seg(o, 0xDB, b);
    }
    seg(o, 0xC0, {8, (uint8_t)(height >> 8), (uint8_t)height, (uint8_t)(width >> 8), (uint8_t)width, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1});
    for (int k = 0; k < 4; ++k) {  // DC0, AC0, DC1, AC1: one marker segment each
        std::vector<uint8_t> b;
        b.push_back((uint8_t)(((k & 1) << 4) | (k >> 1)));
        for (int l = 1; l <= 16; ++l) b.push_back(kSpec[k].bits[l]);
        b.insert(b.end(), kSpec[k].vals, kSpec[k].vals + kSpec[k].n);
        seg(o, 0xC4, b);
    }
    seg(o, 0xDA, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});
    const int mcux = (width + 15) / 16, mcuy = (height + 15) / 16;
    const int ybw = mcux * 2;
    const size_t yblocks = (size_t)ybw * mcuy * 2, cblocks = (size_t)mcux * mcuy;
    BitWriter bw{o};
    int pred[3] = {0, 0, 0};
    auto put_block = [&](const int16_t *z, int comp) {
        const int dk = comp ? 2 : 0, ak = comp ? 3 : 1;
        int diff = z[0] - pred[comp];
        pred[comp] = z[0];
        int mag = diff < 0 ? -diff : diff, bits = diff < 0 ? diff - 1 : diff;
        int nb = nbits(mag);
        bw.put(t.code[dk][nb], t.len[dk][nb]);
        if (nb) bw.put((unsigned)bits, nb);
        int run = 0;
        for (int k = 1; k < 64; ++k) {
            const int v = z[k];
            if (v == 0) {
                ++run;
                continue;
            }
            while (run > 15) {
                bw.put(t.code[ak][0xF0], t.len[ak][0xF0]);
                run -= 16;
            }
            mag = v < 0 ? -v : v;
            bits = v < 0 ? v - 1 : v;
            nb = nbits(mag);
            const int sym = (run << 4) | nb;
            bw.put(t.code[ak][sym], t.len[ak][sym]);
            bw.put((unsigned)bits, nb);
            run = 0;
        }
        if (run) bw.put(t.code[ak][0], t.len[ak][0]);  // EOB
    };
    // Blocks that lie completely outside the image (they only exist to fill the last MCU column / row) are not transformed pixels in
    // the IJG encoder: their AC terms are zero and their DC repeats the previous block of the MCU, so that they cost (almost) no bits.
    const int yw_real = (width + 7) / 8, yh_real = (height + 7) / 8;                    // real luma blocks
    const int cw_real = ((width + 1) / 2 + 7) / 8, ch_real = ((height + 1) / 2 + 7) / 8;  // real chroma blocks
    int16_t dummy[64];
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            int prev_dc = 0;  // DC of the previous block in this MCU's luma sequence
            for (int v = 0; v < 2; ++v)
                for (int h = 0; h < 2; ++h) {
                    const int16_t *b = coef + ((size_t)(my * 2 + v) * ybw + mx * 2 + h) * 64;
                    if (my * 2 + v >= yh_real || mx * 2 + h >= yw_real) {
                        std::memset(dummy, 0, sizeof(dummy));
                        dummy[0] = (int16_t)prev_dc;
                        b = dummy;
                    }
                    prev_dc = b[0];
                    put_block(b, 0);
                }
            for (int c = 1; c <= 2; ++c) {  // one chroma block per MCU: it is real iff the MCU reaches into the image (always true)
                (void)cw_real;
                (void)ch_real;
                put_block(coef + (yblocks + (c - 1) * cblocks + (size_t)my * mcux + mx) * 64, c);
            }
        }
    bw.flush();
    o.push_back(0xFF);
    o.push_back(0xD9);
}

std::string base64(const uint8_t *d, size_t n) {
    static const char *A = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string o;
    o.reserve((n + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 2 < n; i += 3) {
        const unsigned v = (d[i] << 16) | (d[i + 1] << 8) | d[i + 2];
        o += A[v >> 18];
        o += A[(v >> 12) & 63];
        o += A[(v >> 6) & 63];
        o += A[v & 63];
    }
    if (i + 1 == n) {
        const unsigned v = d[i] << 16;
        o += A[v >> 18];
        o += A[(v >> 12) & 63];
        o += "==";
    } else if (i + 2 == n) {
        const unsigned v = (d[i] << 16) | (d[i + 1] << 8);
        o += A[v >> 18];
        o += A[(v >> 12) & 63];
        o += A[(v >> 6) & 63];
        o += '=';
    }
    return o;
}

}  // namespace frtjpeg
