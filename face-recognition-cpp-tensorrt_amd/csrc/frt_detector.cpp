// libfrt.so: the detector object (weights, network, post-processing) and its C ABI (frt_detector_*, frame ingest).
// All device work is hand-written HIP (kernels_*.hip); there is no CPU fallback anywhere in this file: without a HIP
// device every entry point that needs one fails with FRT_ERR_DEVICE.
#include "frt_detector.hpp"

namespace {

// conv weight [Cout][Cin][3][3] (+BN) -> transposed [Cin][9][Cout] fp32 with the BN scale folded, bias [Cout]
void fold_conv3(const frt::Blob &b, const std::string &conv, const std::string &bn, int cout, int cin, std::vector<float> &w, std::vector<float> &bias) {
    const float *src = b.get(conv + ".weight", (size_t)cout * cin * 9).data;
    std::vector<float> sc, bi;
    frt::bn_fold(b, bn, cout, sc, bi);
    w.assign((size_t)cin * 9 * cout, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 9; ++t) w[((size_t)ci * 9 + t) * cout + co] = src[((size_t)co * cin + ci) * 9 + t] * sc[co];
    bias = bi;
}
void fold_dw(const frt::Blob &b, const std::string &conv, const std::string &bn, int c, std::vector<float> &w, std::vector<float> &bias) {
    const float *src = b.get(conv + ".weight", (size_t)c * 9).data;
    std::vector<float> sc, bi;
    frt::bn_fold(b, bn, c, sc, bi);
    w.resize((size_t)c * 9);
    for (int i = 0; i < c; ++i)
        for (int t = 0; t < 9; ++t) w[(size_t)i * 9 + t] = src[(size_t)i * 9 + t] * sc[i];
    bias = bi;
}
void fold_pw(const frt::Blob &b, const std::string &conv, const std::string &bn, int cout, int cin, std::vector<float> &w, std::vector<float> &bias) {
    const float *src = b.get(conv + ".weight", (size_t)cout * cin).data;
    std::vector<float> sc, bi;
    frt::bn_fold(b, bn, cout, sc, bi);
    w.resize((size_t)cin * cout);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) w[(size_t)ci * cout + co] = src[(size_t)co * cin + ci] * sc[co];
    bias = bi;
}
inline int conv_out(int x, int stride) { return (x + 2 - 3) / stride + 1; }
// [Cin][9][Cout] fp32 -> fp16 hi/lo split [Cin/16][9][64][hi16 | lo16] (kernels_det_conv3h.hip); empty unless Cin is 64 or 16 and 16 <= Cout <= 64
std::vector<uint16_t> pack_conv3_split(const std::vector<float> &w, int cin, int cout) {
    if ((cin != 64 && cin != 16) || cout > 64 || cout < 16) return {};
    const int nch = cin / 16;
    std::vector<uint16_t> o((size_t)nch * 9 * 64 * 32, 0);
    auto h2f = [](uint16_t h) {  // fp16 -> fp32 (normal / subnormal / zero; no inf/nan expected in weights)
        const uint32_t sgn = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 31, m = h & 1023;
        float f;
        if (e == 0) f = std::ldexp((float)m, -24);
        else f = std::ldexp((float)(m | 1024), (int)e - 25);
        return sgn ? -f : f;
    };
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < 9; ++t)
            for (int co = 0; co < cout; ++co)
                for (int k = 0; k < 16; ++k) {
                    const float x = w[((size_t)(c * 16 + k) * 9 + t) * cout + co];
                    const uint16_t hi = frt::f32_to_f16(x);
                    const uint16_t lo = frt::f32_to_f16(x - h2f(hi));
                    const size_t row = (((size_t)c * 9 + t) * 64 + co) * 32;
                    o[row + k] = hi;
                    o[row + 16 + k] = lo;
                }
    return o;
}
// pointwise weights [Cin][Cout] fp32 -> fp16 hi/lo split [Cout][Cin/16][hi16 | lo16] (dwpw_mfma_kernel / pw_mfma_kernel); empty unless Cin % 16 == 0
std::vector<uint16_t> pack_pw_split(const std::vector<float> &w, int cin, int cout) {
    if (cin % 16) return {};
    auto h2f = [](uint16_t h) {
        const uint32_t sgn = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 31, m = h & 1023;
        const float f = e == 0 ? std::ldexp((float)m, -24) : std::ldexp((float)(m | 1024), (int)e - 25);
        return sgn ? -f : f;
    };
    std::vector<uint16_t> o((size_t)cout * cin * 2, 0);
    for (int co = 0; co < cout; ++co)
        for (int k = 0; k < cin; ++k) {
            const float x = w[(size_t)k * cout + co];
            const uint16_t hi = frt::f32_to_f16(x);
            const size_t row = ((size_t)co * (cin / 16) + k / 16) * 32;
            o[row + k % 16] = hi;
            o[row + 16 + k % 16] = frt::f32_to_f16(x - h2f(hi));
        }
    return o;
}
// [Cin][9][Cout] -> matrix-core layout [9][Cin/kc][cpad][kc] (kernels_det_conv3.hip); empty when the shape is not covered
std::vector<float> pack_conv3_mfma(const std::vector<float> &w, int cin, int cout, int &kc, int &cpad) {
    kc = cin == 16 ? 16 : 32;
    cpad = cout > 32 ? 64 : 32;
    if (cin % kc || cout > 64 || cout < 16) return {};
    const int ncc = cin / kc;
    std::vector<float> o((size_t)9 * ncc * cpad * kc, 0.f);
    for (int t = 0; t < 9; ++t)
        for (int cc = 0; cc < ncc; ++cc)
            for (int co = 0; co < cout; ++co)
                for (int k = 0; k < kc; ++k) o[(((size_t)t * ncc + cc) * cpad + co) * kc + k] = w[((size_t)(cc * kc + k) * 9 + t) * cout + co];
    return o;
}

}  // namespace

void frt_detector::build(const frt::Blob &b) {
    const int B = max_batch, H = g.in_h, W = g.in_w;
    std::vector<float> w, bias, w2, bias2;
    auto act = [&](int c, int h, int w_) { return arena.alloc<float>((size_t)B * c * h * w_); };
    auto add_c3 = [&](const float *in, float *out, const std::string &conv, const std::string &bn, int cin, int cout, int h, int w_, int stride,
                      int ctotal, int coff) {
        fold_conv3(b, conv, bn, cout, cin, w, bias);
        Op o{};
        o.type = 1;
        o.n = 1;
        o.c3[0] = Conv3Args{in, out, arena.upload(w), arena.upload(bias), B, cin, h, w_, cout, conv_out(h, stride), conv_out(w_, stride), stride, 1, ctotal, coff};
        if (stride == 1) {
            const std::vector<float> pk = pack_conv3_mfma(w, cin, cout, o.c3[0].wm_kc, o.c3[0].wm_cpad);
            if (!pk.empty()) o.c3[0].wm = arena.upload(pk);
            const std::vector<uint16_t> ph = pack_conv3_split(w, cin, cout);
            if (!ph.empty()) o.c3[0].wh = reinterpret_cast<const half_t *>(arena.upload(ph));
        }
        ops.push_back(o);
        flops_per_frame += 2.0 * cin * 9 * cout * o.c3[0].Ho * o.c3[0].Wo;
    };
    // the same conv on every pyramid level -> ONE launch (blockIdx.z = level)
    auto add_c3_levels = [&](const float *const in[3], float *const out[3], const std::string &name, int cin, int cout, const int *hs, const int *ws,
                             int ctotal, int coff) {
        Op o{};
        o.type = 1;
        o.n = 3;
        for (int k = 0; k < 3; ++k) {
            const std::string pfx = "ssh" + std::to_string(k + 1) + "." + name;
            fold_conv3(b, pfx + ".0", pfx + ".1", cout, cin, w, bias);
            o.c3[k] = Conv3Args{in[k], out[k], arena.upload(w), arena.upload(bias), B, cin, hs[k], ws[k], cout, hs[k], ws[k], 1, 1, ctotal, coff};
            const std::vector<float> pk = pack_conv3_mfma(w, cin, cout, o.c3[k].wm_kc, o.c3[k].wm_cpad);
            if (!pk.empty()) o.c3[k].wm = arena.upload(pk);
            const std::vector<uint16_t> ph = pack_conv3_split(w, cin, cout);
            if (!ph.empty()) o.c3[k].wh = reinterpret_cast<const half_t *>(arena.upload(ph));
            flops_per_frame += 2.0 * cin * 9 * cout * hs[k] * ws[k];
        }
        ops.push_back(o);
    };
    // two convs reading the same input on every level (SSH conv3X3 64->32 and conv5X5_1 64->16): ONE matrix-core launch with the
    // output channels concatenated and a split epilogue; the two separate ops stay behind it as the scalar fallback
    auto add_c3_pair_levels = [&](const float *const in[3], float *const outa[3], const std::string &na, int couta, int ctotala, int coffa,
                                  float *const outb[3], const std::string &nb, int coutb, int ctotalb, int coffb, int cin, const int *hs,
                                  const int *ws) {
        Op o{};
        o.type = 3;
        o.n = 3;
        const int cout = couta + coutb;
        for (int k = 0; k < 3; ++k) {
            const std::string pa = "ssh" + std::to_string(k + 1) + "." + na, pb = "ssh" + std::to_string(k + 1) + "." + nb;
            fold_conv3(b, pa + ".0", pa + ".1", couta, cin, w, bias);
            fold_conv3(b, pb + ".0", pb + ".1", coutb, cin, w2, bias2);
            std::vector<float> wc((size_t)cin * 9 * cout), bc(bias);
            bc.insert(bc.end(), bias2.begin(), bias2.end());
            for (size_t row = 0; row < (size_t)cin * 9; ++row) {
                std::copy(w.begin() + row * couta, w.begin() + (row + 1) * couta, wc.begin() + row * cout);
                std::copy(w2.begin() + row * coutb, w2.begin() + (row + 1) * coutb, wc.begin() + row * cout + couta);
            }
            o.c3[k] = Conv3Args{in[k], outa[k], nullptr, arena.upload(bc), B, cin, hs[k], ws[k], cout, hs[k], ws[k], 1, 1, ctotala, coffa};
            const std::vector<float> pk = pack_conv3_mfma(wc, cin, cout, o.c3[k].wm_kc, o.c3[k].wm_cpad);
            if (pk.empty()) raise(FRT_ERR_INVALID, "detector: fused SSH conv shape not covered");
            o.c3[k].wm = arena.upload(pk);
            const std::vector<uint16_t> ph = pack_conv3_split(wc, cin, cout);
            if (!ph.empty()) o.c3[k].wh = reinterpret_cast<const half_t *>(arena.upload(ph));
            o.c3[k].out2 = outb[k];
            o.c3[k].split = couta;
            o.c3[k].out2_ctotal = ctotalb;
            o.c3[k].out2_coff = coffb;
        }
        ops.push_back(o);
    };
    // ---- body (net.py:102-124); return layers stage1/2/3 (config.py:17)
    struct L {
        int cin, cout, stride;
    };
    const std::vector<std::pair<std::string, std::vector<L>>> stages = {
        {"stage1", {{3, 8, 2}, {8, 16, 1}, {16, 32, 2}, {32, 32, 1}, {32, 64, 2}, {64, 64, 1}}},
        {"stage2", {{64, 128, 2}, {128, 128, 1}, {128, 128, 1}, {128, 128, 1}, {128, 128, 1}, {128, 128, 1}}},
        {"stage3", {{128, 256, 2}, {256, 256, 1}}}};
    const float *cur = d_input;
    int ch = H, cw = W;
    const float *feat[3];
    int fh[3], fw[3];
    int si = 0;
    for (auto &st : stages) {
        for (size_t i = 0; i < st.second.size(); ++i) {
            const L l = st.second[i];
            const std::string p = "body." + st.first + "." + std::to_string(i);
            const int oh = conv_out(ch, l.stride), ow = conv_out(cw, l.stride);
            float *out = act(l.cout, oh, ow);
            if (l.cin == 3) {
                add_c3(cur, out, p + ".0", p + ".1", 3, l.cout, ch, cw, l.stride, l.cout, 0);
            } else {
                fold_dw(b, p + ".0", p + ".1", l.cin, w, bias);
                fold_pw(b, p + ".3", p + ".4", l.cout, l.cin, w2, bias2);
                Op o{};
                o.type = 0;
                std::vector<float> w12((size_t)l.cin * 12, 0.f);
                for (int ci = 0; ci < l.cin; ++ci) {
                    for (int t = 0; t < 9; ++t) w12[(size_t)ci * 12 + t] = w[(size_t)ci * 9 + t];
                    w12[(size_t)ci * 12 + 9] = bias[ci];
                }
                o.dw = DwPwArgs{cur, out, arena.upload(w), arena.upload(bias), arena.upload(w2), arena.upload(bias2), nullptr, 0, 0,
                                B, l.cin, ch, cw, l.cout, oh, ow, l.stride, 1, d_tmp, arena.upload(w12), nullptr};
                {
                    const std::vector<uint16_t> ph = pack_pw_split(w2, l.cin, l.cout);
                    if (!ph.empty()) o.dw.wph = reinterpret_cast<const half_t *>(arena.upload(ph));
                    if (l.cin % 2 == 0) {  // depthwise weights of channel pairs (kernels_det_wave.hip, kernels_det_stem.hip)
                        std::vector<float> wp2((size_t)l.cin * 10, 0.f);  // [Cin/2][10][2]: taps 0-8, bias; the channel pair interleaved
                        for (int ci = 0; ci < l.cin; ++ci) {
                            for (int t = 0; t < 9; ++t) wp2[(size_t)(ci / 2) * 20 + 2 * t + (ci & 1)] = w[(size_t)ci * 9 + t];
                            wp2[(size_t)(ci / 2) * 20 + 18 + (ci & 1)] = bias[ci];
                        }
                        o.dw.wdp = arena.upload(wp2);
                        if (l.cin <= 16) {
                            std::vector<float> wt((size_t)l.cin * 10, 0.f);
                            for (int ci = 0; ci < l.cin; ++ci) {
                                for (int t = 0; t < 9; ++t) wt[((size_t)t * (l.cin / 2) + ci / 2) * 2 + (ci & 1)] = w[(size_t)ci * 9 + t];
                                wt[((size_t)9 * (l.cin / 2) + ci / 2) * 2 + (ci & 1)] = bias[ci];
                            }
                            o.dw.wdt = arena.upload(wt);
                        }
                    }
                    if (!ph.empty() && l.cout % 32 == 0) {  // the other operands of dwpw_wave_kernel
                        std::vector<uint16_t> pf(ph.size());
                        const int ng = l.cin / 16, ncb = l.cout / 32;
                        for (int gq = 0; gq < ng; ++gq)
                            for (int cb = 0; cb < ncb; ++cb)
                                for (int part = 0; part < 2; ++part)
                                    for (int ln = 0; ln < 64; ++ln)
                                        for (int j = 0; j < 8; ++j)
                                            pf[((((size_t)gq * ncb + cb) * 2 + part) * 64 + ln) * 8 + j] =
                                                ph[((size_t)(cb * 32 + (ln & 31)) * ng + gq) * 32 + part * 16 + 8 * (ln >> 5) + j];
                        if (!d_wave_zeros) {
                            d_wave_zeros = arena.alloc<float>(dwpw_wave_zero_bytes() / 4);
                            HIPCHK(hipMemset(d_wave_zeros, 0, dwpw_wave_zero_bytes()));
                        }
                        o.dw.zeros = d_wave_zeros;
                        o.dw.wpf = reinterpret_cast<const half_t *>(arena.upload(pf));
                    }
                }
                ops.push_back(o);
                flops_per_frame += 2.0 * oh * ow * (9.0 * l.cin + (double)l.cin * l.cout);
            }
            cur = out;
            ch = oh;
            cw = ow;
        }
        feat[si] = cur;
        fh[si] = ch;
        fw[si] = cw;
        ++si;
    }
    // the first three layers as one kernel (kernels_det_stem.hip): their weights gathered into one buffer
    if (ops.size() >= 3 && ops[0].type == 1 && ops[0].n == 1 && ops[1].type == 0 && ops[2].type == 0 && ops[1].dw.wdt && ops[2].dw.wdt &&
        ops[1].dw.Cin == 8 && ops[1].dw.Cout == 16 && ops[2].dw.Cin == 16 && ops[2].dw.Cout == 32) {
        float *stem = arena.alloc<float>(det_stem_weight_floats());
        det_stem_pack(ops[0].c3[0], ops[1].dw, ops[2].dw, stem, nullptr);
        HIPCHK(hipStreamSynchronize(nullptr));
        ops[1].dw.stem = stem;
    }
    for (int k = 0; k < 3; ++k)
        if (fh[k] != g.fh[k] || fw[k] != g.fw[k]) raise(FRT_ERR_INVALID, "detector: feature-map size mismatch");
    // ---- FPN (net.py:81-98): laterals 1x1+BN+ReLU, nearest-upsample-add top-down (fused), 3x3 merges
    const int cins[3] = {64, 128, 256};
    float *lat[3];
    auto add_lat = [&](int k, const float *addsrc, int ah, int aw) {
        const std::string p = "fpn.output" + std::to_string(k + 1);
        fold_pw(b, p + ".0", p + ".1", 64, cins[k], w2, bias2);
        lat[k] = act(64, fh[k], fw[k]);
        Op o{};
        o.type = 0;
        o.dw = DwPwArgs{feat[k], lat[k], nullptr, nullptr, arena.upload(w2), arena.upload(bias2), addsrc, ah, aw,
                        B, cins[k], fh[k], fw[k], 64, fh[k], fw[k], 1, 1, nullptr, nullptr, nullptr};
        {
            const std::vector<uint16_t> ph = pack_pw_split(w2, cins[k], 64);
            if (!ph.empty()) o.dw.wph = reinterpret_cast<const half_t *>(arena.upload(ph));
        }
        ops.push_back(o);
        flops_per_frame += 2.0 * fh[k] * fw[k] * cins[k] * 64;
    };
    add_lat(2, nullptr, 0, 0);
    add_lat(1, lat[2], fh[2], fw[2]);
    float *p4 = act(64, fh[1], fw[1]);
    add_c3(lat[1], p4, "fpn.merge2.0", "fpn.merge2.1", 64, 64, fh[1], fw[1], 1, 64, 0);
    add_lat(0, p4, fh[1], fw[1]);
    float *p3 = act(64, fh[0], fw[0]);
    add_c3(lat[0], p3, "fpn.merge1.0", "fpn.merge1.1", 64, 64, fh[0], fw[0], 1, 64, 0);
    float *const pyr_m[3] = {p3, p4, lat[2]};
    const float *const pyr[3] = {p3, p4, lat[2]};
    (void)pyr_m;
    // ---- SSH (net.py:55-66) + heads (retinaface_trim.py:14-35).  Every SSH conv ends in a ReLU: either its own or the
    //      ReLU applied to the concat it feeds exclusively.
    float *cat[3], *t1[3], *t2[3];
    for (int k = 0; k < 3; ++k) {
        cat[k] = act(64, fh[k], fw[k]);
        t1[k] = act(16, fh[k], fw[k]);
        t2[k] = act(16, fh[k], fw[k]);
    }
    add_c3_pair_levels(pyr, cat, "conv3X3", 32, 64, 0, t1, "conv5X5_1", 16, 16, 0, 64, fh, fw);  // type 3: skips the next two ops when it ran
    add_c3_levels(pyr, cat, "conv3X3", 64, 32, fh, fw, 64, 0);
    add_c3_levels(pyr, t1, "conv5X5_1", 64, 16, fh, fw, 16, 0);
    {
        // conv5X5_2 (-> cat[32:48]) and conv7X7_2 (-> t2) read the same 16-channel tensor: one launch with the output channels
        // concatenated (two channel tiles of the scalar kernel, the second writing to t2); same weights, same summation order
        Op o{};
        o.type = 1;
        o.n = 3;
        for (int k = 0; k < 3; ++k) {
            const std::string pa = "ssh" + std::to_string(k + 1) + ".conv5X5_2", pb = "ssh" + std::to_string(k + 1) + ".conv7X7_2";
            fold_conv3(b, pa + ".0", pa + ".1", 16, 16, w, bias);
            fold_conv3(b, pb + ".0", pb + ".1", 16, 16, w2, bias2);
            std::vector<float> wc((size_t)16 * 9 * 32), bc(bias);
            bc.insert(bc.end(), bias2.begin(), bias2.end());
            for (size_t row = 0; row < (size_t)16 * 9; ++row) {
                std::copy(w.begin() + row * 16, w.begin() + (row + 1) * 16, wc.begin() + row * 32);
                std::copy(w2.begin() + row * 16, w2.begin() + (row + 1) * 16, wc.begin() + row * 32 + 16);
            }
            o.c3[k] = Conv3Args{t1[k], cat[k], arena.upload(wc), arena.upload(bc), B, 16, fh[k], fw[k], 32, fh[k], fw[k], 1, 1, 64, 32};
            {
                const std::vector<uint16_t> ph = pack_conv3_split(wc, 16, 32);  // round 5: the 16-channel SSH convs on the split-fp16 matrix-core kernel
                if (!ph.empty()) o.c3[k].wh = reinterpret_cast<const half_t *>(arena.upload(ph));
            }
            o.c3[k].out2 = t2[k];
            o.c3[k].split = 16;
            o.c3[k].out2_ctotal = 16;
            o.c3[k].out2_coff = 0;
            flops_per_frame += 2.0 * 16 * 9 * 32 * fh[k] * fw[k];
        }
        ops.push_back(o);
    }
    add_c3_levels(t2, cat, "conv7x7_3", 16, 16, fh, fw, 64, 48);
    Op ho{};
    ho.type = 2;
    ho.n = 3;
    for (int k = 0; k < 3; ++k) {
        const std::string hb = "BboxHead." + std::to_string(k) + ".conv1x1", hc = "ClassHead." + std::to_string(k) + ".conv1x1";
        const float *wb = b.get(hb + ".weight", 8 * 64).data, *wc = b.get(hc + ".weight", 4 * 64).data;
        std::vector<float> tb(64 * 8), tc(64 * 4);
        for (int co = 0; co < 8; ++co)
            for (int ci = 0; ci < 64; ++ci) tb[ci * 8 + co] = wb[co * 64 + ci];
        for (int co = 0; co < 4; ++co)
            for (int ci = 0; ci < 64; ++ci) tc[ci * 4 + co] = wc[co * 64 + ci];
        std::vector<float> bb(b.get(hb + ".bias", 8).data, b.get(hb + ".bias", 8).data + 8);
        std::vector<float> bc(b.get(hc + ".bias", 4).data, b.get(hc + ".bias", 4).data + 4);
        ho.hd[k] = HeadArgs{cat[k], arena.upload(tb), arena.upload(bb), arena.upload(tc), arena.upload(bc), d_loc, d_conf, B, 64, fh[k], fw[k], g.A, g.base[k],
                            nullptr, nullptr, nullptr};
        flops_per_frame += 2.0 * fh[k] * fw[k] * 64 * 12;
        if (has_landmarks) {
            const std::string hl = "LandmarkHead." + std::to_string(k) + ".conv1x1";
            const float *wl = b.get(hl + ".weight", 20 * 64).data, *bl = b.get(hl + ".bias", 20).data;
            std::vector<float> tl(64 * 20), blv(bl, bl + 20);
            for (int co = 0; co < 20; ++co)
                for (int ci = 0; ci < 64; ++ci) tl[ci * 20 + co] = wl[co * 64 + ci];
            ho.hd[k].wl = arena.upload(tl);
            ho.hd[k].bl = arena.upload(blv);
            ho.hd[k].ldm = d_ldm;
            flops_per_frame += 2.0 * fh[k] * fw[k] * 64 * 20;
        }
    }
    ops.push_back(ho);
}

void frt_detector::preprocess(const uint8_t *frames_dev, int n, size_t row_stride, size_t frame_stride, hipStream_t s) {
    ProfScope ps(2, "det_preprocess", (double)n * g.frame_h * g.frame_w * 3, s);
    launch_det_preprocess(frames_dev, n, g.frame_h, g.frame_w, row_stride, frame_stride, g.in_h, g.in_w, d_input, s);
}

void frt_detector::forward_frames(const uint8_t *frames_dev, int n, size_t row_stride, size_t frame_stride, hipStream_t s) {
    if (g.frame_h == g.in_h && g.frame_w == g.in_w && !ops.empty() && ops[0].type == 1 && ops[0].n == 1) {
        Conv3Args c = ops[0].c3[0];
        c.B = n;
        if (ops.size() >= 3 && ops[1].type == 0 && ops[2].type == 0) {  // first conv + the first two conv_dw blocks in one kernel
#ifdef FRT_TUNING
            if (frt_tuning_env("FRT_DET_STEM_CHECK")) {  // debugging aid: the three layers one by one against the fused kernel, element for element
                const size_t cnt = (size_t)n * ops[2].dw.Cout * ops[2].dw.Ho * ops[2].dw.Wo;
                std::vector<float> ref(cnt), got(cnt);
                (void)launch_det_conv1_u8(frames_dev, row_stride, frame_stride, c, s);
                ops[1].dw.B = n; ops[2].dw.B = n;
                launch_dwpw(ops[1].dw, s);
                launch_dwpw(ops[2].dw, s);
                HIPCHK(hipStreamSynchronize(s));
                HIPCHK(hipMemcpy(ref.data(), ops[2].dw.out, cnt * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemset(ops[2].dw.out, 0xff, cnt * 4));
                const bool ran = launch_det_stem(frames_dev, row_stride, frame_stride, c, ops[1].dw, ops[2].dw, s);
                HIPCHK(hipStreamSynchronize(s));
                HIPCHK(hipMemcpy(got.data(), ops[2].dw.out, cnt * 4, hipMemcpyDeviceToHost));
                size_t bad = 0, first = cnt;
                double maxd = 0;
                for (size_t i = 0; i < cnt; ++i) {
                    const double d = std::fabs((double)ref[i] - (double)got[i]);
                    if (!(d == 0)) { if (first == cnt) first = i; ++bad; }
                    if (d > maxd || d != d) maxd = d;
                }
                const int hw = ops[2].dw.Ho * ops[2].dw.Wo;
                fprintf(stderr, "[stem check] ran %d, %zu of %zu differ, max |d| %g", (int)ran, bad, cnt, maxd);
                if (first < cnt) fprintf(stderr, "; first at b=%zu c=%zu y=%zu x=%zu: got %g want %g", first / ((size_t)32 * hw), (first / hw) % 32, (first % hw) / ops[2].dw.Wo, first % ops[2].dw.Wo, got[first], ref[first]);
                fprintf(stderr, "\n");
                size_t by_c[32] = {0};
                for (size_t i = 0; i < cnt; ++i) if (ref[i] != got[i]) ++by_c[(i / hw) % 32];
                fprintf(stderr, "[stem check] differing by channel:");
                for (int k = 0; k < 32; ++k) fprintf(stderr, " %zu", by_c[k]);
                fprintf(stderr, "\n");
            }
#endif
            bool stem;
            {
                ProfScope ps(2, "det_stem", (double)n * g.frame_h * g.frame_w * 3, s);
                stem = launch_det_stem(frames_dev, row_stride, frame_stride, c, ops[1].dw, ops[2].dw, s);
            }
            if (stem) return forward(n, s, 3);
        }
        bool fused;
        {
            ProfScope ps(2, "det_preprocess", (double)n * g.frame_h * g.frame_w * 3, s);  // fused into the first conv
            fused = launch_det_conv1_u8(frames_dev, row_stride, frame_stride, c, s);
        }
        if (fused) return forward(n, s, 1);
    }
    preprocess(frames_dev, n, row_stride, frame_stride, s);
    forward(n, s);
}

void frt_detector::forward(int n, hipStream_t s, int first_op) {
    ProfScope ps(2, "det_network", flops_per_frame * n, s);
    int skip = first_op;
    for (Op &o : ops) {
        if (skip > 0) {
            --skip;
            continue;
        }
        if (o.type == 3) {
            for (int k = 0; k < o.n; ++k) o.c3[k].B = n;
            if (det_mfma_enabled() && (launch_conv3x3_split(o.c3, o.n, s) || launch_conv3x3_mfma(o.c3, o.n, s))) skip = 2;  // else: the two separate convs
            continue;
        }
        if (o.type == 0) {
            o.dw.B = n;
            launch_dwpw(o.dw, s);
        } else if (o.type == 1) {
            for (int k = 0; k < o.n; ++k) o.c3[k].B = n;
            launch_conv3x3_multi(o.c3, o.n, s);
        } else {
            for (int k = 0; k < o.n; ++k) o.hd[k].B = n;
            launch_heads_multi(o.hd, o.n, s);
        }
    }
    HIPCHK(hipGetLastError());  // a failed launch (e.g. the dynamic-LDS opt-in missing on this device) must not pass silently
}

void frt_detector::postprocess(int n, hipStream_t s, frt_bbox *boxes_out, int *nout_out, float *landmarks_out) {
    ProfScope ps(2, "det_postprocess", (double)n * g.A, s);
    frt_bbox *bo = boxes_out ? boxes_out : d_boxes;  // the pipeline passes its slot buffers: no device-to-device copies afterwards
    int *no = nout_out ? nout_out : d_nout;
    launch_decode(d_loc, d_conf, n, g, d_cand, d_cand_count, s);
    launch_nms(d_cand, d_loc, d_cand_count, n, g, d_dead, bo, no, d_kept_anchor, s);
    if (has_landmarks) launch_landmark_decode(d_ldm, d_kept_anchor, no, n, g, landmarks_out ? landmarks_out : d_landmarks, s);
    HIPCHK(hipGetLastError());
}


extern "C" {

int frt_detector_create(const char *weights_path, int frame_w, int frame_h, int in_c, int in_h, int in_w, int max_batch, int max_faces,
                        float nms_threshold, float bbox_threshold, int device, frt_detector **out) {
    return guarded([&] {
        if (!out || !weights_path) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (in_c != 3 || in_h < 32 || in_w < 32 || frame_w < 1 || frame_h < 1 || max_batch < 1 || max_faces < 1)
            raise(FRT_ERR_INVALID, "detector: invalid shape arguments");
        frt::Blob blob;
        std::string err;
        const int rc = blob.load(weights_path, err);
        if (rc) raise(rc, err);
        if (blob.kind != 1) raise(FRT_ERR_FORMAT, "detector: weight blob is not a RetinaFace-mobilenet0.25 blob");
        use_device(device);
        std::unique_ptr<frt_detector> d(new frt_detector);
        d->device = device;
        d->max_batch = max_batch;
        DetGeom &g = d->g;
        g.in_w = in_w; g.in_h = in_h; g.frame_w = frame_w; g.frame_h = frame_h;
        const float steps[3] = {8.f, 16.f, 32.f};
        int base = 0;
        for (int k = 0; k < 3; ++k) {
            g.fh[k] = (int)std::ceil(in_h / steps[k]);
            g.fw[k] = (int)std::ceil(in_w / steps[k]);
            g.base[k] = base;
            base += g.fh[k] * g.fw[k] * 2;
        }
        g.A = base;
        g.scale_h = (float)in_h / frame_h;  // retinaface.cpp:21-22
        g.scale_w = (float)in_w / frame_w;
        g.nms_thr = nms_threshold;
        g.bbox_thr = bbox_threshold;
        g.max_faces = max_faces;
        HIPCHK(hipStreamCreate(&d->stream));
        HIPCHK(hipEventCreateWithFlags(&d->ev_busy, hipEventDisableTiming));
        const size_t B = (size_t)max_batch;
        d->d_frames = d->arena.alloc<uint8_t>(B * frame_h * frame_w * 3);
        d->d_input = d->arena.alloc<float>(B * 3 * in_h * in_w);
        d->d_loc = d->arena.alloc<float>(B * g.A * 4);
        d->d_conf = d->arena.alloc<float>(B * g.A * 2);
        d->d_cand = d->arena.alloc<Candidate>(B * g.A);
        d->d_cand_count = d->arena.alloc<int>((size_t)B * 32);  // one 128-byte line per frame (kernels_post.hip: CC_STRIDE)
        HIPCHK(hipMemset(d->d_cand_count, 0, sizeof(int) * B * 32));  // kept at zero between calls by nms_kernel
        d->d_nout = d->arena.alloc<int>(B);
        d->d_dead = d->arena.alloc<uint8_t>(B * g.A);
        d->d_boxes = d->arena.alloc<frt_bbox>(B * max_faces);
        d->d_tmp = d->arena.alloc<float>(B * 64 * (size_t)g.fh[0] * g.fw[0]);  // largest depthwise intermediate of a split conv_dw block
        d->has_landmarks = blob.has("LandmarkHead.0.conv1x1.weight");
        if (d->has_landmarks) {
            d->d_ldm = d->arena.alloc<float>(B * g.A * 10);
            d->d_kept_anchor = d->arena.alloc<int>(B * max_faces);
            d->d_landmarks = d->arena.alloc<float>(B * max_faces * 10);
        }
        d->build(blob);
        HIPCHK(hipDeviceSynchronize());
        *out = d.release();
    });
}

void frt_detector_destroy(frt_detector *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream) {
        (void)hipStreamSynchronize(d->stream);
        (void)hipStreamDestroy(d->stream);
    }
    if (d->ev_busy) (void)hipEventDestroy(d->ev_busy);
    d->arena.release();
    delete d;
}

int frt_detector_num_anchors(const frt_detector *d) { return d ? d->g.A : 0; }
int frt_detector_geometry(const frt_detector *d, int *frame_w, int *frame_h, int *max_batch, int *max_faces, int *device) {
    if (!d) return FRT_ERR_INVALID;
    if (frame_w) *frame_w = d->g.frame_w;
    if (frame_h) *frame_h = d->g.frame_h;
    if (max_batch) *max_batch = d->max_batch;
    if (max_faces) *max_faces = d->g.max_faces;
    if (device) *device = d->device;
    return FRT_OK;
}

int frt_detector_find_faces_batch(frt_detector *d, const uint8_t *bgr, int n_frames, int rows, int cols, size_t row_stride,
                                  size_t frame_stride, frt_bbox *out, int *n_out) {
    return guarded([&] {
        if (!d || !bgr || !out || !n_out) raise(FRT_ERR_INVALID, "null argument");
        if (rows != d->g.frame_h || cols != d->g.frame_w) raise(FRT_ERR_INVALID, "findFace: frame must be frameWidth x frameHeight");
        if (n_frames < 1 || n_frames > d->max_batch) raise(FRT_ERR_CAPACITY, "findFace: more frames than det_maxBatchSize");
        std::lock_guard<std::mutex> lk(d->mu);
        use_device(d->device);
        hipStream_t s = d->stream;
        d->wait_idle(s);
        const size_t tight = (size_t)cols * 3;
        for (int f = 0; f < n_frames; ++f)
            HIPCHK(hipMemcpy2DAsync(d->d_frames + (size_t)f * rows * tight, tight, bgr + (size_t)f * frame_stride, row_stride, tight, rows,
                                    hipMemcpyHostToDevice, s));
        d->forward_frames(d->d_frames, n_frames, tight, (size_t)rows * tight, s);
        d->postprocess(n_frames, s);
        HIPCHK(hipMemcpyAsync(out, d->d_boxes, sizeof(frt_bbox) * n_frames * d->g.max_faces, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(n_out, d->d_nout, sizeof(int) * n_frames, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_detector_find_faces(frt_detector *d, const uint8_t *bgr, int rows, int cols, size_t row_stride, frt_bbox *out, int *n_out) {
    return frt_detector_find_faces_batch(d, bgr, 1, rows, cols, row_stride, row_stride * (size_t)rows, out, n_out);
}

int frt_detector_has_landmarks(const frt_detector *d) { return d && d->has_landmarks ? 1 : 0; }

int frt_detector_find_faces_landmarks(frt_detector *d, const uint8_t *bgr, int rows, int cols, size_t row_stride, frt_bbox *out,
                                      float *landmarks_out, int *n_out) {
    return guarded([&] {
        if (!d || !bgr || !out || !n_out || !landmarks_out) raise(FRT_ERR_INVALID, "null argument");
        if (!d->has_landmarks) raise(FRT_ERR_FORMAT, "findFaceLandmarks: the detector blob has no LandmarkHead (trimmed export)");
        if (rows != d->g.frame_h || cols != d->g.frame_w) raise(FRT_ERR_INVALID, "findFace: frame must be frameWidth x frameHeight");
        std::lock_guard<std::mutex> lk(d->mu);
        use_device(d->device);
        hipStream_t s = d->stream;
        d->wait_idle(s);
        const size_t tight = (size_t)cols * 3;
        HIPCHK(hipMemcpy2DAsync(d->d_frames, tight, bgr, row_stride, tight, rows, hipMemcpyHostToDevice, s));
        d->forward_frames(d->d_frames, 1, tight, (size_t)rows * tight, s);
        d->postprocess(1, s);
        HIPCHK(hipMemcpyAsync(out, d->d_boxes, sizeof(frt_bbox) * d->g.max_faces, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(landmarks_out, d->d_landmarks, sizeof(float) * 10 * d->g.max_faces, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(n_out, d->d_nout, sizeof(int), hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_detector_preprocess(frt_detector *d, const uint8_t *bgr, int rows, int cols, size_t row_stride, float *chw_out) {
    return guarded([&] {
        if (!d || !bgr || !chw_out) raise(FRT_ERR_INVALID, "null argument");
        if (rows != d->g.frame_h || cols != d->g.frame_w) raise(FRT_ERR_INVALID, "preprocess: frame must be frameWidth x frameHeight");
        std::lock_guard<std::mutex> lk(d->mu);
        use_device(d->device);
        hipStream_t s = d->stream;
        d->wait_idle(s);
        const size_t tight = (size_t)cols * 3;
        HIPCHK(hipMemcpy2DAsync(d->d_frames, tight, bgr, row_stride, tight, rows, hipMemcpyHostToDevice, s));
        d->preprocess(d->d_frames, 1, tight, (size_t)rows * tight, s);
        HIPCHK(hipMemcpyAsync(chw_out, d->d_input, sizeof(float) * 3 * d->g.in_h * d->g.in_w, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_detector_infer(frt_detector *d, const float *chw, int batch, float *loc_out, float *conf_out) {
    return guarded([&] {
        if (!d || !chw || !loc_out || !conf_out) raise(FRT_ERR_INVALID, "null argument");
        if (batch < 1 || batch > d->max_batch) raise(FRT_ERR_CAPACITY, "doInference: batch exceeds det_maxBatchSize");
        std::lock_guard<std::mutex> lk(d->mu);
        use_device(d->device);
        hipStream_t s = d->stream;
        d->wait_idle(s);
        const size_t in_elems = (size_t)3 * d->g.in_h * d->g.in_w;
        HIPCHK(hipMemcpyAsync(d->d_input, chw, sizeof(float) * in_elems * batch, hipMemcpyHostToDevice, s));
        d->forward(batch, s);
        HIPCHK(hipMemcpyAsync(loc_out, d->d_loc, sizeof(float) * (size_t)batch * d->g.A * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(conf_out, d->d_conf, sizeof(float) * (size_t)batch * d->g.A * 2, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_detector_infer_landmarks(frt_detector *d, const float *chw, int batch, float *loc_out, float *conf_out, float *ldm_out) {
    return guarded([&] {
        if (!d || !chw || !loc_out || !conf_out || !ldm_out) raise(FRT_ERR_INVALID, "null argument");
        if (!d->has_landmarks) raise(FRT_ERR_FORMAT, "doInference: the detector blob has no LandmarkHead (trimmed export)");
        if (batch < 1 || batch > d->max_batch) raise(FRT_ERR_CAPACITY, "doInference: batch exceeds det_maxBatchSize");
        std::lock_guard<std::mutex> lk(d->mu);
        use_device(d->device);
        hipStream_t s = d->stream;
        d->wait_idle(s);
        const size_t in_elems = (size_t)3 * d->g.in_h * d->g.in_w;
        HIPCHK(hipMemcpyAsync(d->d_input, chw, sizeof(float) * in_elems * batch, hipMemcpyHostToDevice, s));
        d->forward(batch, s);
        HIPCHK(hipMemcpyAsync(loc_out, d->d_loc, sizeof(float) * (size_t)batch * d->g.A * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(conf_out, d->d_conf, sizeof(float) * (size_t)batch * d->g.A * 2, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(ldm_out, d->d_ldm, sizeof(float) * (size_t)batch * d->g.A * 10, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_detector_postprocess(frt_detector *d, const float *loc, const float *conf, frt_bbox *out, int *n_out) {
    return guarded([&] {
        if (!d || !loc || !conf || !out || !n_out) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(d->mu);
        use_device(d->device);
        hipStream_t s = d->stream;
        d->wait_idle(s);
        HIPCHK(hipMemcpyAsync(d->d_loc, loc, sizeof(float) * (size_t)d->g.A * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d->d_conf, conf, sizeof(float) * (size_t)d->g.A * 2, hipMemcpyHostToDevice, s));
        d->postprocess(1, s);
        HIPCHK(hipMemcpyAsync(out, d->d_boxes, sizeof(frt_bbox) * d->g.max_faces, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(n_out, d->d_nout, sizeof(int), hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

// -------------------------------------------------------------------------------------------------------- frame ingest
int frt_resize_frame(const uint8_t *bgr, int rows, int cols, size_t row_stride, uint8_t *out, int out_rows, int out_cols, int device) {
    return guarded([&] {
        if (!bgr || !out || rows < 1 || cols < 1 || out_rows < 1 || out_cols < 1) raise(FRT_ERR_INVALID, "resize: bad argument");
        if (device >= 0) use_device(device);
        Arena a;
        struct Guard {
            Arena &a;
            ~Guard() { a.release(); }
        } guard{a};
        const size_t tight = (size_t)cols * 3, otight = (size_t)out_cols * 3;
        uint8_t *d_src = a.alloc<uint8_t>((size_t)rows * tight);
        uint8_t *d_dst = a.alloc<uint8_t>((size_t)out_rows * otight);
        HIPCHK(hipMemcpy2D(d_src, tight, bgr, row_stride, tight, rows, hipMemcpyHostToDevice));
        launch_resize_linear(d_src, 1, rows, cols, tight, 0, d_dst, out_rows, out_cols, otight, 0, nullptr);
        HIPCHK(hipMemcpy(out, d_dst, (size_t)out_rows * otight, hipMemcpyDeviceToHost));
    });
}

int frt_resize_frames_dev(const void *src_dev, int n, int rows, int cols, size_t row_stride, size_t frame_stride, void *dst_dev, int out_rows,
                          int out_cols, void *hip_stream) {
    return guarded([&] {
        if (!src_dev || !dst_dev || n < 0 || rows < 1 || cols < 1 || out_rows < 1 || out_cols < 1) raise(FRT_ERR_INVALID, "resize: bad argument");
        launch_resize_linear(reinterpret_cast<const uint8_t *>(src_dev), n, rows, cols, row_stride, frame_stride, reinterpret_cast<uint8_t *>(dst_dev),
                             out_rows, out_cols, (size_t)out_cols * 3, (size_t)out_rows * out_cols * 3, reinterpret_cast<hipStream_t>(hip_stream));
        HIPCHK(hipGetLastError());
    });
}


}  // extern "C"
