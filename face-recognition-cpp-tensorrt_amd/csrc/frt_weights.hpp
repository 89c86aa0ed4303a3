// FRTW weight-blob reader + host-side folding helpers (see weights_io.py for the format).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace frt {

struct Tensor {
    const float *data = nullptr;
    std::vector<uint32_t> dims;
    size_t numel = 0;
};

class Blob {
  public:
    uint32_t kind = 0;
    std::map<std::string, Tensor> t;
    std::vector<char> buf;

    // returns 0 ok, 2 not found, 3 malformed
    int load(const char *path, std::string &err) {
        FILE *f = std::fopen(path, "rb");
        if (!f) {
            err = "Cant find engine file";  // the reference's message (src/retinaface.cpp:53, src/arcface.cpp:67)
            return 2;
        }
        std::fseek(f, 0, SEEK_END);
        long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (sz < 0) {  // not seekable (directory, pipe): ftell failed
            std::fclose(f);
            err = "weight blob: bad magic / truncated file";
            return 3;
        }
        buf.resize((size_t)sz);
        size_t rd = std::fread(buf.data(), 1, (size_t)sz, f);
        std::fclose(f);
        if ((long)rd != sz || sz < 16 || std::memcmp(buf.data(), "FRTW0001", 8) != 0) {
            err = "weight blob: bad magic / truncated file";
            return 3;
        }
        uint32_t n;
        std::memcpy(&kind, buf.data() + 8, 4);
        std::memcpy(&n, buf.data() + 12, 4);
        size_t p = 16;
        for (uint32_t i = 0; i < n; ++i) {
            if (p + 2 > buf.size()) return bad(err);
            uint16_t ln;
            std::memcpy(&ln, buf.data() + p, 2);
            p += 2;
            if (p + ln + 1 > buf.size()) return bad(err);
            std::string name(buf.data() + p, ln);
            p += ln;
            uint8_t nd = (uint8_t)buf[p++];
            Tensor x;
            if (p + 4u * nd + 16 > buf.size()) return bad(err);
            for (int d = 0; d < nd; ++d) {
                uint32_t v;
                std::memcpy(&v, buf.data() + p, 4);
                p += 4;
                x.dims.push_back(v);
            }
            uint64_t off, ne;
            std::memcpy(&off, buf.data() + p, 8);
            std::memcpy(&ne, buf.data() + p + 8, 8);
            p += 16;
            if ((off & 3) || off > buf.size() || ne > (buf.size() - off) / 4) return bad(err);  // (no uint64 wrap: ne is checked against the room left)
            x.data = reinterpret_cast<const float *>(buf.data() + off);
            x.numel = (size_t)ne;
            t[name] = x;
        }
        return 0;
    }
    bool has(const std::string &n) const { return t.count(n) != 0; }
    const Tensor &get(const std::string &n, size_t expect_numel) const {
        auto it = t.find(n);
        if (it == t.end()) throw std::runtime_error("weight blob: missing tensor " + n);
        if (expect_numel && it->second.numel != expect_numel) throw std::runtime_error("weight blob: wrong size for " + n);
        return it->second;
    }

  private:
    static int bad(std::string &err) {
        err = "weight blob: malformed header";
        return 3;
    }
};

// BatchNorm (eval) as y = x*scale + bias; eps = 1e-5 (PyTorch default, never overridden by the reference).
inline void bn_fold(const Blob &b, const std::string &p, int c, std::vector<float> &scale, std::vector<float> &bias) {
    const float *g = b.get(p + ".weight", c).data, *be = b.get(p + ".bias", c).data;
    const float *mu = b.get(p + ".running_mean", c).data, *var = b.get(p + ".running_var", c).data;
    scale.resize(c);
    bias.resize(c);
    for (int i = 0; i < c; ++i) {
        const double s = (double)g[i] / std::sqrt((double)var[i] + 1e-5);
        scale[i] = (float)s;
        bias[i] = (float)((double)be[i] - (double)mu[i] * s);
    }
}

// IEEE fp32 -> fp16, round-to-nearest-even (clang's native _Float16 conversion; this file is compiled by hipcc only).
inline uint16_t f32_to_f16(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

}  // namespace frt
