// ASan + UBSan job over libfrt's pure-host parsers (round-1 VERDICT, hygiene item 9): the JPEG marker parser / Huffman decoder / JFIF
// writer (csrc/frt_jpeg.cpp) and the FRTW weight-blob reader (csrc/frt_weights.hpp) are fed valid, truncated and randomly corrupted
// inputs.  Built by tests/test_sanitizers.py with clang++ -fsanitize=address,undefined -fno-sanitize-recover=all: any out-of-bounds
// access, overflow or misaligned load aborts the process.
//   sanitize_host <file.jpg>... -- <file.frtw>...
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>

#include "../../face-recognition-cpp-tensorrt_amd/csrc/frt_jpeg.hpp"
#include "../../face-recognition-cpp-tensorrt_amd/csrc/frt_weights.hpp"

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static int try_jpeg(const std::vector<uint8_t> &b) {
    frtjpeg::Parsed p;
    std::string err;
    if (frtjpeg::parse(b.data(), b.size(), p, err)) return 1;
    if (p.h.total_blocks > (1u << 20)) return 2;
    std::vector<int16_t> coef(p.h.total_blocks * 64, 0);
    return frtjpeg::decode_coefficients(b.data(), b.size(), p, coef.data(), err) ? 3 : 0;  // sequential or progressive
}

int main(int argc, char **argv) {
    std::mt19937 rng(1234);
    int i = 1, ok = 0, rejected = 0;
    for (; i < argc && std::strcmp(argv[i], "--"); ++i) {
        const std::vector<uint8_t> good = slurp(argv[i]);
        if (try_jpeg(good) == 0) ++ok;
        for (size_t cut = 0; cut < good.size(); cut += 1 + good.size() / 97) {  // every truncation point class
            std::vector<uint8_t> t(good.begin(), good.begin() + cut);
            rejected += try_jpeg(t) != 0;
        }
        // targeted DHT damage (round-2 advisor finding): the code-length COUNTS are rewritten so that the segment length and the
        // 256-symbol total stay consistent - random byte damage almost never does that - with over-subscribed, all-ones and
        // count-swapped tables.  parse() must reject what violates the Kraft limit before HuffTable::build() runs.
        for (size_t q = 2; q + 4 < good.size(); ++q) {
            if (good[q] != 0xFF || good[q + 1] != 0xC4) continue;
            const size_t len = ((size_t)good[q + 2] << 8) | good[q + 3], tab = q + 4;  // first table of the segment
            if (len < 19 || tab + 17 > good.size()) continue;
            int total = 0;
            for (int l = 1; l <= 16; ++l) total += good[tab + l];
            for (int variant = 0; variant < 40; ++variant) {
                std::vector<uint8_t> t = good;
                uint8_t *bits = t.data() + tab;  // bits[1..16]
                if (variant == 0) {              // everything on length 1: 2^1 codes cannot hold `total` symbols
                    for (int l = 2; l <= 16; ++l) bits[l] = 0;
                    bits[1] = (uint8_t)(total > 255 ? 255 : total);
                } else if (variant == 1) {       // the advisor's reproducer shape: bits[1] = 255
                    bits[1] = 255;
                } else if (variant == 2) {       // all ones (16 symbols, valid Kraft sum) - must not crash either way
                    for (int l = 1; l <= 16; ++l) bits[l] = 1;
                } else if (variant < 20) {       // swap two counts (total unchanged)
                    const int a = 1 + rng() % 16, b = 1 + rng() % 16;
                    std::swap(bits[a], bits[b]);
                } else {                         // move symbols towards the short lengths (total unchanged)
                    const int from = 5 + rng() % 12, to = 1 + rng() % 4;
                    const int n = bits[from];
                    if (bits[to] + n <= 255) {
                        bits[to] = (uint8_t)(bits[to] + n);
                        bits[from] = 0;
                    }
                }
                rejected += try_jpeg(t) != 0;
            }
        }
        for (int k = 0; k < 400; ++k) {  // random byte / bit damage anywhere, headers included
            std::vector<uint8_t> t = good;
            const int n = 1 + rng() % 6;
            for (int j = 0; j < n; ++j) t[rng() % t.size()] = (uint8_t)rng();
            rejected += try_jpeg(t) != 0;
        }
    }
    // JFIF writer on extreme coefficient values and sizes
    frtjpeg::EncTables t;
    for (int q : {1, 50, 95, 100}) {
        frtjpeg::make_enc_tables(q, t);
        for (int w : {1, 16, 17, 112, 250})
            for (int h : {1, 15, 112}) {
                const int mx = (w + 15) / 16, my = (h + 15) / 16;
                std::vector<int16_t> c((size_t)6 * mx * my * 64);
                for (int16_t &v : c) v = (int16_t)((int)(rng() % 2047) - 1023);
                for (size_t b = 0; b < c.size(); b += 64) c[b] = (int16_t)((int)(rng() % 2047) - 1023);  // DC differences stay within category 11
                std::vector<uint8_t> out;
                frtjpeg::write_jfif_420(t, w, h, c.data(), out);
                if (try_jpeg(out) != 0) {  // what we write must parse and decode again
                    std::printf("writer output does not round-trip (q=%d %dx%d)\n", q, w, h);
                    return 1;
                }
            }
    }
    for (size_t n : {0u, 1u, 2u, 3u, 4u, 1000u}) {
        std::vector<uint8_t> d(n);
        for (uint8_t &v : d) v = (uint8_t)rng();
        if (frtjpeg::base64(d.data(), n).size() != (n + 2) / 3 * 4) return 1;
    }
    // weight blobs: valid file, truncations, header damage (incl. the uint64 offset / count fields)
    int blobs_ok = 0;
    for (++i; i < argc; ++i) {
        const std::vector<uint8_t> good = slurp(argv[i]);
        const std::string tmp = std::string(argv[i]) + ".fuzz";
        auto load = [&](const std::vector<uint8_t> &b) {
            std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<const char *>(b.data()), (std::streamsize)b.size());
            frt::Blob blob;
            std::string err;
            const int rc = blob.load(tmp.c_str(), err);
            if (rc == 0) {  // touch every tensor end to end: offsets that passed validation must be readable
                double s = 0;
                for (auto &kv : blob.t)
                    if (kv.second.numel) s += kv.second.data[0] + kv.second.data[kv.second.numel - 1];
                (void)s;
            }
            return rc;
        };
        blobs_ok += load(good) == 0;
        const size_t head = good.size() < 4096 ? good.size() : 4096;
        for (int k = 0; k < 300; ++k) {
            std::vector<uint8_t> t = good;
            if (k % 3 == 0) t.resize(rng() % good.size());
            const int n = 1 + rng() % 4;
            for (int j = 0; j < n && !t.empty(); ++j) t[rng() % (head < t.size() ? head : t.size())] = (uint8_t)rng();
            load(t);
        }
        std::remove(tmp.c_str());
    }
    std::printf("sanitize ok: %d jpeg(s) decoded, %d damaged inputs rejected, %d blob(s) loaded\n", ok, rejected, blobs_ok);
    return 0;
}
