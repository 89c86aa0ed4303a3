// libfrt.so: the recogniser object (weights, fp16 and fp32 networks) and its C ABI (frt_embedder_*, frt_crop_faces, frt_align_faces).
// All device work is hand-written HIP (kernels_*.hip); there is no CPU fallback anywhere in this file: without a HIP
// device every entry point that needs one fails with FRT_ERR_DEVICE.
#include "frt_embedder.hpp"

namespace {

std::vector<uint16_t> conv_w_f16(const float *src, int cout, int cin, int ks) {
    // [Cout][Cin][kh][kw] fp32 -> [Cout][kh][kw][Cin] fp16 (K index = tap*Cin + ci)
    std::vector<uint16_t> w((size_t)cout * cin * ks * ks);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < ks * ks; ++t) w[((size_t)co * ks * ks + t) * cin + ci] = frt::f32_to_f16(src[((size_t)co * cin + ci) * ks * ks + t]);
    return w;
}
std::vector<uint16_t> conv_w_f16(const frt::Blob &b, const std::string &name, int cout, int cin, int ks) {
    return conv_w_f16(b.get(name, (size_t)cout * cin * ks * ks).data, cout, cin, ks);
}
// 3x3 weights in the order the strip kernel's MFMA A fragments consume them: [Cout/32][Cin/64][tap][kk][lane = (k half, cout row)][8]
// (kernels_arc.hip: conv_patch_kernel); a wave's load of one fragment is then one contiguous kilobyte.  Empty unless Cin % 64 == 0.
// stride2: taps in the step order of the stride-2 strip kernel (kernels_arc_s2.hip: phase planes (odd,odd) (even,even) (odd,even) (even,odd)).
std::vector<uint16_t> conv_w_f16_frag(const float *src, int cout, int cin, bool stride2 = false) {
    if (cin % 64 || cout % 32) return {};
    static const int s2_step_of_tap[9] = {0, 5, 1, 7, 4, 8, 2, 6, 3};  // inverse of the step -> tap table 0,2,6,8,4,1,7,3,5
    std::vector<uint16_t> w((size_t)cout * cin * 9);
    const int nch = cin / 64;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 9; ++t) {
                const int blk = co >> 5, r = co & 31, ch = ci >> 6, kk = (ci & 63) >> 4, hi = (ci & 15) >> 3, e = ci & 7;
                const int st = stride2 ? s2_step_of_tap[t] : t;
                const size_t off = (((((size_t)blk * nch + ch) * 9 + st) * 4 + kk) * 64 + hi * 32 + r) * 8 + e;
                w[off] = frt::f32_to_f16(src[((size_t)co * cin + ci) * 9 + t]);
            }
    return w;
}
// 1x1 shortcut weights [Cout][Cin] in the stride-2 strip kernel's fragment order [Cout/32][Cin/64][kk][lane = (k half, cout row)][8]
std::vector<uint16_t> conv1x1_w_f16_frag(const frt::Blob &b, const std::string &name, int cout, int cin) {
    if (cin % 64 || cout % 32) return {};
    const float *src = b.get(name, (size_t)cout * cin).data;
    std::vector<uint16_t> w((size_t)cout * cin);
    const int nch = cin / 64;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const int blk = co >> 5, r = co & 31, ch = ci >> 6, kk = (ci & 63) >> 4, hi = (ci & 15) >> 3, e = ci & 7;
            w[(((((size_t)blk * nch + ch) * 4 + kk) * 64) + hi * 32 + r) * 8 + e] = frt::f32_to_f16(src[(size_t)co * cin + ci]);
        }
    return w;
}
std::vector<float> vec_of(const frt::Blob &b, const std::string &name, size_t n) {
    const float *p = b.get(name, n).data;
    return std::vector<float>(p, p + n);
}

}  // namespace

void frt_embedder::build(const frt::Blob &b) {
    std::vector<float> sc, bi;
    // input layer (model_irse.py:139-141)
    {
        const float *src = b.get("input_layer.0.weight", 64 * 27).data;
        std::vector<float> w(27 * 64);
        for (int co = 0; co < 64; ++co)
            for (int k = 0; k < 27; ++k) w[k * 64 + co] = src[co * 27 + k];
        in_w = arena.upload(w);
        frt::bn_fold(b, "input_layer.1", 64, sc, bi);
        in_s0 = arena.upload(sc);
        in_b0 = arena.upload(bi);
        std::vector<uint16_t> wh(64 * 32, 0);  // matrix-core layout (kernels_arc_input.hip): BN folded, bias in tap slot 27
        for (int co = 0; co < 64; ++co) {
            for (int k = 0; k < 27; ++k) wh[co * 32 + k] = frt::f32_to_f16(src[co * 27 + k] * sc[co]);
            wh[co * 32 + 27] = frt::f32_to_f16(bi[co]);
        }
        in_wh = reinterpret_cast<half_t *>(arena.upload(wh));
        in_slope = arena.upload(vec_of(b, "input_layer.2.weight", 64));
        frt::bn_fold(b, "body.0.res_layer.0", 64, sc, bi);
        in_s1 = arena.upload(sc);
        in_b1 = arena.upload(bi);
        flops_per_face += 2.0 * 27 * 64 * 112 * 112;
    }
    // units (model_irse.py:97-109 for IR-50)
    const int cfg[4][3] = {{64, 64, 3}, {64, 128, 4}, {128, 256, 14}, {256, 512, 3}};
    const bool condition = !(frt_tuning_env("FRT_ARC_CONDITION") && frt_tuning_env("FRT_ARC_CONDITION")[0] == '0');  // (tuning build: the sweep's "off" leg)
    int h = 112, idx = 0;
    for (int st = 0; st < 4; ++st)
        for (int u = 0; u < cfg[st][2]; ++u) {
            ArcUnit a;
            a.cin = u == 0 ? cfg[st][0] : cfg[st][1];
            a.depth = cfg[st][1];
            a.stride = u == 0 ? 2 : 1;
            a.h_in = h;
            const std::string p = "body." + std::to_string(idx);
            // Conditioning of the branch conv1 -> PReLU -> conv2 -> BN (round 5; model_irse.py:57-66).  conv1's accumulators leave as the fp16
            // tensor T and both convs multiply fp16 weights: a trained backbone (conversion/arcface/torch2trt.py:21-22 loads one nobody here
            // has seen) may keep that branch orders of magnitude away from 1 - tools/dynamic_range_sweep.py: a branch 1e-4 times smaller pushes
            // T and conv1's weights into fp16's subnormals and conv2's towards its overflow, SILENTLY (1 - cos 4.8e-4).  PReLU is positively
            // homogeneous, so for powers of two c_j, d_k > 0 the unit computes exactly the same function with
            //     conv1 row j * c_j      conv2 column j / c_j, row k * d_k      BN scale k / d_k        (every scaling exact in binary fp)
            // c_j brings conv1's row norm to ~ 1 (T = O(1) behind a normalised input), d_k conv2's largest row entry into [0.5, 1).  A branch
            // that is in range already is left bit for bit as it was (the scalings commute with every rounding).
            std::vector<float> w1v = vec_of(b, p + ".res_layer.1.weight", (size_t)a.depth * a.cin * 9);
            std::vector<float> w2v = vec_of(b, p + ".res_layer.3.weight", (size_t)a.depth * a.depth * 9);
            std::vector<float> dinv(a.depth, 1.f);
            if (condition) {
                auto pow2_inv = [](double v) {  // 2^-round(log2 v), clamped; 1 for zero / non-finite rows
                    if (!(v > 0.0) || !std::isfinite(v)) return 1.0;
                    const double e = std::max(-60.0, std::min(60.0, -std::nearbyint(std::log2(v))));
                    return std::exp2(e);
                };
                for (int j = 0; j < a.depth; ++j) {
                    double n2 = 0.0;
                    float *row = &w1v[(size_t)j * a.cin * 9];
                    for (int i = 0; i < a.cin * 9; ++i) n2 += (double)row[i] * row[i];
                    const double c = pow2_inv(std::sqrt(n2));
                    if (c == 1.0) continue;
                    for (int i = 0; i < a.cin * 9; ++i) row[i] = (float)(row[i] * c);
                    for (int k = 0; k < a.depth; ++k)
                        for (int t = 0; t < 9; ++t) {
                            float &v = w2v[((size_t)k * a.depth + j) * 9 + t];
                            v = (float)(v / c);
                        }
                }
                for (int k = 0; k < a.depth; ++k) {
                    double mx = 0.0;
                    float *row = &w2v[(size_t)k * a.depth * 9];
                    for (int i = 0; i < a.depth * 9; ++i) mx = std::max(mx, (double)std::fabs(row[i]));
                    double d = 1.0;
                    if (mx > 0.0 && std::isfinite(mx) && (mx >= 2.0 || mx < 0.03125)) d = std::exp2(std::max(-60.0, std::min(60.0, -std::ceil(std::log2(mx)))));
                    if (d == 1.0) continue;   // (entries already inside [2^-5, 2): nothing to gain, keep the trained numbers as they are)
                    for (int i = 0; i < a.depth * 9; ++i) row[i] = (float)(row[i] * d);
                    dinv[k] = (float)(1.0 / d);
                }
            }
            a.w1 = reinterpret_cast<half_t *>(arena.upload(conv_w_f16(w1v.data(), a.depth, a.cin, 3)));
            {  // conv1 is always stride 1; conv2 only in the units that keep the resolution
                const std::vector<uint16_t> f1 = conv_w_f16_frag(w1v.data(), a.depth, a.cin);
                if (!f1.empty()) a.w1f = reinterpret_cast<half_t *>(arena.upload(f1));
                // the 64 -> 64 stride-2 layer has its own kernel that stages rows in natural order and walks the taps in tap order
                const std::vector<uint16_t> f2 = conv_w_f16_frag(w2v.data(), a.depth, a.depth, a.stride == 2 && a.depth != 64);
                if (!f2.empty()) (a.stride == 1 ? a.w2f : a.w2f2) = reinterpret_cast<half_t *>(arena.upload(f2));
            }
            a.prelu = arena.upload(vec_of(b, p + ".res_layer.2.weight", a.depth));
            a.w2 = reinterpret_cast<half_t *>(arena.upload(conv_w_f16(w2v.data(), a.depth, a.depth, 3)));
            frt::bn_fold(b, p + ".res_layer.4", a.depth, sc, bi);
            a.s2f32 = arena.upload(sc);   // (the fp32 path multiplies the blob's own weights: the unconditioned scale)
            for (int k = 0; k < a.depth; ++k) sc[k] *= dinv[k];
            a.s2 = arena.upload(sc);
            a.b2 = arena.upload(bi);
            if (a.cin != a.depth) {
                a.wsc = reinterpret_cast<half_t *>(arena.upload(conv_w_f16(b, p + ".shortcut_layer.0.weight", a.depth, a.cin, 1)));
                const std::vector<uint16_t> fs = conv1x1_w_f16_frag(b, p + ".shortcut_layer.0.weight", a.depth, a.cin);
                if (!fs.empty()) a.wscf = reinterpret_cast<half_t *>(arena.upload(fs));
                frt::bn_fold(b, p + ".shortcut_layer.1", a.depth, sc, bi);
                a.ssc = arena.upload(sc);
                a.bsc = arena.upload(bi);
            }
            if (se) {
                a.se_w1 = arena.upload(vec_of(b, p + ".res_layer.5.fc1.weight", (size_t)a.depth / 16 * a.depth));
                a.se_w2 = arena.upload(vec_of(b, p + ".res_layer.5.fc2.weight", (size_t)a.depth * (a.depth / 16)));
            }
            const bool last = st == 3 && u == cfg[st][2] - 1;
            frt::bn_fold(b, last ? std::string("output_layer.0") : "body." + std::to_string(idx + 1) + ".res_layer.0", a.depth, sc, bi);
            a.sn = arena.upload(sc);
            a.bn = arena.upload(bi);
            const int ho = h / a.stride;
            flops_per_face += 2.0 * 9 * a.cin * a.depth * h * h + 2.0 * 9 * a.depth * a.depth * ho * ho;
            if (a.wsc) flops_per_face += 2.0 * a.cin * a.depth * ho * ho;
            units.push_back(a);
            h = ho;
            ++idx;
        }
    // output layer (model_irse.py:143-147): Linear over the NCHW flatten (index c*49 + hw) re-ordered to NHWC (hw*512 + c)
    {
        const float *src = b.get("output_layer.3.weight", (size_t)512 * 25088).data;
        // ... and packed in MFMA-fragment order for kernels_arc_fc.hip: [output block o / 32][k step k / 16][lane = (k half, o % 32)][8]
        std::vector<uint16_t> w((size_t)512 * 25088);
        for (int o = 0; o < 512; ++o)
            for (int c = 0; c < 512; ++c)
                for (int hw = 0; hw < 49; ++hw) {
                    const size_t k = (size_t)hw * 512 + c;
                    const size_t off = ((((size_t)(o >> 5) * (25088 / 16) + (k >> 4)) * 64) + ((k >> 3) & 1) * 32 + (o & 31)) * 8 + (k & 7);
                    w[off] = frt::f32_to_f16(src[(size_t)o * 25088 + (size_t)c * 49 + hw]);
                }
        wfc = reinterpret_cast<half_t *>(arena.upload(w));
        fc_bias = arena.upload(vec_of(b, "output_layer.3.bias", 512));
        frt::bn_fold(b, "output_layer.4", 512, sc, bi);
        bn_s = arena.upload(sc);
        bn_b = arena.upload(bi);
        flops_per_face += 2.0 * 25088 * 512;
    }
    {
        const char *sf = frt_tuning_env("FRT_SC_FUSED");
        sc_fusion = !(sf && sf[0] == '0');
    }
    const size_t F = (size_t)max_batch;
    const size_t big = F * 112 * 112 * 64;
    d_in = arena.alloc<float>(F * 3 * 112 * 112);
    for (int i = 0; i < 2; ++i) {
        Y[i] = arena.alloc<half_t>(big);
        Z[i] = arena.alloc<half_t>(big);
    }
    T = arena.alloc<half_t>(big);
    SC = arena.alloc<half_t>(F * 28 * 28 * 128);  // largest conv-shortcut output (56->28, 128 ch)
    if (se) {
        RES = arena.alloc<half_t>(F * 56 * 56 * 64);
        se_pool = arena.alloc<float>(F * 512 * 4 + 2 * F);  // SE_SPLIT partial sums per (face, channel) + per-face arrival counters + gate-ready flags
        HIPCHK(hipMemset(se_pool + F * 512 * 4, 0, 2 * F * sizeof(int)));  // (kept at zero between launches by the kernel)
        se_gate = arena.alloc<float>(F * 512);
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h_se_error), sizeof(int), hipHostMallocMapped));
        *h_se_error = 0;
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_se_error), h_se_error, 0));
        const char *sf = frt_tuning_env("FRT_SE_FUSED");  // (tuning build; the product's switch is frt_embedder_set_se_fused)
        se_fused = !(sf && sf[0] == '0');
    }
    fc_partial = arena.alloc<float>((size_t)FC_SPLITS * F * 512);
    d_out = arena.alloc<float>(F * 512);
    d_crops = arena.alloc<uint8_t>(F * 112 * 112 * 3);
    d_valid = arena.alloc<int>(F);
    d_boxes = arena.alloc<frt_bbox>(F);
    d_lm = arena.alloc<float>((size_t)F * 10);
    zeros = arena.alloc<half_t>(256);
    HIPCHK(hipMemset(zeros, 0, 256 * sizeof(half_t)));
}

void frt_embedder::ensure_alt() {
    if (has_alt) return;
    const size_t F = (size_t)max_batch;
    const size_t big = F * 112 * 112 * 64;
    for (int i = 0; i < 2; ++i) {
        alt.Y[i] = arena.alloc<half_t>(big);
        alt.Z[i] = arena.alloc<half_t>(big);
    }
    alt.T = arena.alloc<half_t>(big);
    alt.SC = arena.alloc<half_t>(F * 28 * 28 * 128);
    alt.RES = nullptr;
    alt.se_pool = alt.se_gate = nullptr;
    if (se) {
        alt.RES = arena.alloc<half_t>(F * 56 * 56 * 64);
        alt.se_pool = arena.alloc<float>(F * 512 * 4 + 2 * F);
        HIPCHK(hipMemset(alt.se_pool + F * 512 * 4, 0, 2 * F * sizeof(int)));
        alt.se_gate = arena.alloc<float>(F * 512);
    }
    alt.fc_partial = arena.alloc<float>((size_t)FC_SPLITS * F * 512);
    has_alt = true;
}

// fp32 weights: 3x3 [Cout][Cin][3][3] -> [Cout][tap][Cin]; 1x1 and Linear as described at the kernels
void frt_embedder::build_f32() {
    if (!f32.units.empty()) return;
    frt::Blob b;
    std::string err;
    const int rc = b.load(blob_path.c_str(), err);
    if (rc) raise(rc, "fp32 mode: cannot re-read the weight blob: " + err);
    auto up = [&](const std::vector<float> &v) {
        float *d = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), v.size() * sizeof(float)));
        f32.owned.push_back(d);
        HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        return d;
    };
    auto dev = [&](size_t n) {
        float *d = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(float)));
        f32.owned.push_back(d);
        return d;
    };
    auto w3 = [&](const std::string &name, int cout, int cin) {
        const float *src = b.get(name, (size_t)cout * cin * 9).data;
        std::vector<float> w((size_t)cout * cin * 9);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t) w[((size_t)co * 9 + t) * cin + ci] = src[((size_t)co * cin + ci) * 9 + t];
        std::vector<float> wf(w.size());
        pack_conv32_weights(w.data(), cout, 9, cin, wf.data());
        return up(wf);
    };
    auto w1x1 = [&](const std::string &name, int cout, int cin) {
        const std::vector<float> w = vec_of(b, name, (size_t)cout * cin);
        std::vector<float> wf(w.size());
        pack_conv32_weights(w.data(), cout, 1, cin, wf.data());
        return up(wf);
    };
    int idx = 0;
    for (const ArcUnit &u : units) {
        const std::string p = "body." + std::to_string(idx++);
        F32Unit fu;
        fu.w1 = w3(p + ".res_layer.1.weight", u.depth, u.cin);
        fu.w2 = w3(p + ".res_layer.3.weight", u.depth, u.depth);
        if (u.wsc) fu.wsc = w1x1(p + ".shortcut_layer.0.weight", u.depth, u.cin);
        f32.units.push_back(fu);
    }
    {
        const float *src = b.get("output_layer.3.weight", (size_t)512 * 25088).data;
        std::vector<float> w((size_t)512 * 25088);
        for (int o = 0; o < 512; ++o)
            for (int c = 0; c < 512; ++c)
                for (int hw = 0; hw < 49; ++hw) w[(size_t)o * 25088 + (size_t)hw * 512 + c] = src[(size_t)o * 25088 + (size_t)c * 49 + hw];
        f32.wfc = up(w);
    }
    f32.chunk = std::min(max_batch, 8);
    const size_t C = (size_t)f32.chunk, big = C * 112 * 112 * 64;
    f32.A[0] = dev(big);
    f32.A[1] = dev(big);
    f32.T = dev(big);
    f32.SCb = dev(C * 56 * 56 * 128);
    if (se) {
        f32.RES = dev(C * 56 * 56 * 64);
        f32.gate = dev(C * 512);
    }
    f32.fc_out = dev(C * 512);
    HIPCHK(hipEventCreateWithFlags(&f32.done, hipEventDisableTiming));
}

// Backbone.forward in fp32 (model_irse.py:166-173), CHUNK faces at a time.  The per-channel parameters (folded BatchNorms, PReLU slopes, SE
// weights, Linear bias) are the fp32 arrays the default path's epilogues use.
void frt_embedder::forward_f32(const float *chw_dev, int F, const int *valid_dev, float *out_dev, hipStream_t s) {
    ProfScope ps(2, "embed_network", flops_per_face * F, s);
    if (f32.busy) HIPCHK(hipStreamWaitEvent(s, f32.done, 0));
    for (int f0 = 0; f0 < F; f0 += f32.chunk) {
        const int n = std::min(f32.chunk, F - f0);
        launch_arc32_input(chw_dev + (size_t)f0 * 3 * 112 * 112, in_w, in_s0, in_b0, in_slope, f32.A[0], n, s);
        int cur = 0;
        const float *lead_s = in_s1, *lead_b = in_b1;  // the leading BatchNorm of the unit about to run
        for (size_t i = 0; i < units.size(); ++i) {
            const ArcUnit &u = units[i];
            const F32Unit &fu = f32.units[i];
            const int h = u.h_in, ho = h / u.stride;
            const float *x = f32.A[cur];
            // conv1: BN(x) (on load) -> conv3x3 -> PReLU
            Conv32Args c1{x, fu.w1, lead_s, lead_b, f32.T, n, h, h, u.cin, h, h, u.depth, 3, 1, 1, 0, u.prelu, nullptr, nullptr, 0, 0, 0};
            launch_conv32(c1, s);
            // shortcut: MaxPool2d(1, stride) of x, or conv1x1 stride s + BN
            const float *sc = x;
            int sc_h = h, sc_stride = u.stride;
            if (fu.wsc) {
                Conv32Args cs{x, fu.wsc, nullptr, nullptr, f32.SCb, n, h, h, u.cin, ho, ho, u.depth, 1, u.stride, 0, 1, u.ssc, u.bsc, nullptr, 0, 0, 0};
                launch_conv32(cs, s);
                sc = f32.SCb;
                sc_h = ho;
                sc_stride = 1;
            }
            // conv2: conv3x3 stride s -> BN (-> SE) -> + shortcut
            float *y = f32.A[cur ^ 1];
            if (se) {
                Conv32Args c2{f32.T, fu.w2, nullptr, nullptr, f32.RES, n, h, h, u.depth, ho, ho, u.depth, 3, u.stride, 1, 1, u.s2f32, u.b2, nullptr, 0, 0, 0};
                launch_conv32(c2, s);
                launch_se32(f32.RES, u.se_w1, u.se_w2, f32.gate, sc, y, n, ho, ho, u.depth, sc_h, sc_h, sc_stride, s);
            } else {
                Conv32Args c2{f32.T, fu.w2, nullptr, nullptr, y, n, h, h, u.depth, ho, ho, u.depth, 3, u.stride, 1, 2, u.s2f32, u.b2, sc, sc_h, sc_h, sc_stride};
                launch_conv32(c2, s);
            }
            lead_s = u.sn;
            lead_b = u.bn;
            cur ^= 1;
        }
        // output_layer: BN2d (on load) -> Flatten -> Linear -> BN1d -> L2 normalise
        launch_fc32(f32.A[cur], lead_s, lead_b, f32.wfc, f32.fc_out, n, s);
        launch_fc_finalize(f32.fc_out, 1, n, fc_bias, bn_s, bn_b, valid_dev ? valid_dev + f0 : nullptr, out_dev + (size_t)f0 * 512, s);
    }
    HIPCHK(hipEventRecord(f32.done, s));
    f32.busy = true;
    HIPCHK(hipGetLastError());
}

void frt_embedder::forward(const float *chw_dev, int F, const int *valid_dev, float *out_dev, hipStream_t s) {
    if (fp32_mode) return forward_f32(chw_dev, F, valid_dev, out_dev, s);
    ProfScope ps(2, "embed_network", flops_per_face * F, s);
    ArcInputArgs ia{chw_dev, in_w, in_s0, in_b0, in_slope, in_s1, in_b1, Y[0], Z[0], F, 112, 112, in_wh};
    launch_arc_input(ia, s);
    int cur = 0;
    for (const ArcUnit &u : units) {
        const int h = u.h_in, ho = h / u.stride;
        {  // conv1: BN(x) [already applied -> Z] -> conv3x3 s1 -> PReLU
            ConvMfmaArgs a{};
            a.x = Z[cur];
            a.w = u.w1;
            a.wf = u.w1f;
            a.B = F; a.H = h; a.W = h; a.Cin = u.cin; a.Ho = h; a.Wo = h; a.Cout = u.depth; a.ks = 3; a.stride = 1; a.pad = 1;
            a.mode = EPI_PRELU;
            a.p0 = u.prelu;
            a.out0 = T;
            a.splits = 1;
            a.zeros = zeros;
            ProfScope pk(1, conv_kernel_label(a), 2.0 * 9 * u.cin * u.depth * (double)F * h * h, s);
            launch_conv_mfma(a, s);
        }
        const half_t *sc_t = Y[cur];
        int sc_h = h, sc_stride = u.stride;
        if (&u == &units[0]) {  // the input layer already wrote its raw output at the even positions only
            sc_h = ho;
            sc_stride = 1;
        }
        // IR-50: the stride-2 strip kernel computes the 1x1 stride-2 shortcut conv itself (its input pixels are the (even, even) phase
        // plane) - no launch, no shortcut tensor.  IR-SE keeps the tensor: the gate multiplies the residual branch only.
        bool sc_fused = false;
        if (u.wsc && u.wscf && !se && u.stride == 2 && sc_fusion) {
            ConvMfmaArgs t{};
            t.x = T; t.w = u.w2; t.wf2 = u.w2f2;
            t.B = F; t.H = h; t.W = h; t.Cin = u.depth; t.Ho = ho; t.Wo = ho; t.Cout = u.depth; t.ks = 3; t.stride = 2; t.pad = 1;
            t.mode = EPI_BN_ADD_BN; t.splits = 1;
            t.scx = Y[cur]; t.wscf = u.wscf; t.psc0 = u.ssc; t.psc1 = u.bsc; t.Csc = u.cin;
            sc_fused = conv_small_applies(t) || conv_s2_applies(t);
        }
        if (u.wsc && !sc_fused) {  // conv1x1 stride s + BN on the raw input
            ConvMfmaArgs a{};
            a.x = Y[cur];
            a.w = u.wsc;
            a.B = F; a.H = h; a.W = h; a.Cin = u.cin; a.Ho = ho; a.Wo = ho; a.Cout = u.depth; a.ks = 1; a.stride = u.stride; a.pad = 0;
            a.mode = EPI_BN;
            a.p0 = u.ssc;
            a.p1 = u.bsc;
            a.out0 = SC;
            a.splits = 1;
            a.zeros = zeros;
            launch_conv_mfma(a, s);
            sc_t = SC;
            sc_h = ho;
            sc_stride = 1;
        }
        {  // conv2: conv3x3 stride s -> BN -> (+SE) -> + shortcut ; also emit BN_next(y)
            ConvMfmaArgs a{};
            a.x = T;
            a.w = u.w2;
            a.wf = u.w2f;
            a.wf2 = u.w2f2;
            a.B = F; a.H = h; a.W = h; a.Cin = u.depth; a.Ho = ho; a.Wo = ho; a.Cout = u.depth; a.ks = 3; a.stride = u.stride; a.pad = 1;
            a.p0 = u.s2;
            a.p1 = u.b2;
            a.splits = 1;
            a.zeros = zeros;
            a.mode = EPI_BN_ADD_BN;
            a.p2 = u.sn;
            a.p3 = u.bn;
            a.sc = sc_t;
            a.sc_h = sc_h; a.sc_w = sc_h; a.sc_stride = sc_stride;
            if (sc_fused) {
                a.sc = nullptr;
                a.scx = Y[cur]; a.wscf = u.wscf; a.psc0 = u.ssc; a.psc1 = u.bsc; a.Csc = u.cin;
            }
            a.out0 = Y[cur ^ 1];
            a.out1 = Z[cur ^ 1];
            bool se_tail = false;  // IR-SE: the SE tail as separate launches behind conv2
            if (se) {
                int *cnt = reinterpret_cast<int *>(se_pool + (size_t)max_batch * 512 * 4);
                a.se_pool = se_pool;
                a.se_w1 = u.se_w1;
                a.se_w2 = u.se_w2;
                a.se_counter = cnt;
                a.se_flag_off = max_batch;
                a.se_error = d_se_error;
                if (se_fused && conv_se_fused(a)) {  // the strip kernel runs the whole tail in its epilogue
                    if (se_epoch >= (1 << 30)) {  // the flags carry launch numbers: start over with clean flags (both scratch sets)
                        HIPCHK(hipMemsetAsync(cnt + max_batch, 0, (size_t)max_batch * sizeof(int), s));
                        if (has_alt) HIPCHK(hipMemsetAsync(reinterpret_cast<int *>(alt.se_pool + (size_t)max_batch * 512 * 4) + max_batch, 0, (size_t)max_batch * sizeof(int), s));
                        se_epoch = 0;
                    }
                    a.se_epoch = ++se_epoch;
                    a.mode = EPI_BN_SE;
                } else {
                    a.mode = EPI_BN;
                    a.out0 = RES;
                    a.out1 = nullptr;
                    a.sc = nullptr;
                    se_tail = true;
                }
            }
            {
                ProfScope pk(1, conv_kernel_label(a), (2.0 * 9 * u.depth * u.depth + (sc_fused ? 2.0 * u.cin * u.depth : 0.0)) * (double)F * ho * ho, s);
                launch_conv_mfma(a, s);
            }
            if (se_tail) {
                SeArgs sa{RES, u.se_w1, u.se_w2, sc_t, sc_h, sc_h, sc_stride, u.sn, u.bn, Y[cur ^ 1], Z[cur ^ 1], se_pool, se_gate, F, ho, ho, u.depth,
                          reinterpret_cast<int *>(se_pool + (size_t)max_batch * 512 * 4)};
                launch_se(sa, s);
            }
        }
        cur ^= 1;
    }
    {  // Linear 25088 -> 512 as 49 K-slices over the NHWC-flattened BN2d output (Z), then slice sum + bias + BN1d + L2 norm
        launch_fc_slices(Z[cur], wfc, F, fc_partial, s);
        launch_fc_finalize(fc_partial, FC_SPLITS, F, fc_bias, bn_s, bn_b, valid_dev, out_dev, s);
    }
    HIPCHK(hipGetLastError());
}


extern "C" {

// ---------------------------------------------------------------------------------------------------------------- crop
int frt_crop_faces(const uint8_t *bgr, int rows, int cols, size_t row_stride, const frt_bbox *boxes, int n, int out_w, int out_h,
                   uint8_t *crops_out, int device) {
    return guarded([&] {
        if (!bgr || !boxes || !crops_out || n < 0 || out_w < 1 || out_h < 1) raise(FRT_ERR_INVALID, "getCroppedFaces: bad argument");
        if (n == 0) return;
        if (device >= 0) use_device(device);
        Arena a;
        struct Guard {
            Arena &a;
            ~Guard() { a.release(); }
        } guard{a};
        const size_t tight = (size_t)cols * 3;
        uint8_t *d_frame = a.alloc<uint8_t>((size_t)rows * tight);
        frt_bbox *d_boxes = a.alloc<frt_bbox>(n);
        uint8_t *d_crops = a.alloc<uint8_t>((size_t)n * out_h * out_w * 3);
        float *d_chw = a.alloc<float>((size_t)n * out_h * out_w * 3);
        int *d_valid = a.alloc<int>(n);
        HIPCHK(hipMemcpy2D(d_frame, tight, bgr, row_stride, tight, rows, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_boxes, boxes, sizeof(frt_bbox) * n, hipMemcpyHostToDevice));
        launch_crop_faces(d_frame, rows, cols, tight, 0, d_boxes, nullptr, 1, n, 1, out_h, out_w, d_crops, d_chw, d_valid, nullptr);
        std::vector<int> valid(n);
        HIPCHK(hipMemcpy(valid.data(), d_valid, sizeof(int) * n, hipMemcpyDeviceToHost));
        std::vector<uint8_t> tmp((size_t)n * out_h * out_w * 3);
        HIPCHK(hipMemcpy(tmp.data(), d_crops, tmp.size(), hipMemcpyDeviceToHost));
        bool bad = false;
        for (int i = 0; i < n; ++i) {
            if (valid[i])
                std::memcpy(crops_out + (size_t)i * out_h * out_w * 3, tmp.data() + (size_t)i * out_h * out_w * 3, (size_t)out_h * out_w * 3);
            else
                bad = true;
        }
        if (bad) raise(FRT_ERR_EMPTY_ROI, "getCroppedFaces: empty or out-of-frame ROI");
    });
}

// ------------------------------------------------------------------------------------------------------------ embedder
int frt_embedder_create(const char *weights_path, int in_c, int in_h, int in_w, int out_dim, int max_batch, int device, frt_embedder **out) {
    return guarded([&] {
        if (!out || !weights_path) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (in_c != 3 || in_h != 112 || in_w != 112 || out_dim != 512 || max_batch < 1)
            raise(FRT_ERR_INVALID, "embedder: only rec_inputShape [3,112,112] and rec_outputDim 512 are supported");
        frt::Blob blob;
        std::string err;
        const int rc = blob.load(weights_path, err);
        if (rc) raise(rc, err);
        if (blob.kind != 2 && blob.kind != 3) raise(FRT_ERR_FORMAT, "embedder: weight blob is not an ArcFace IR-50 / IR-SE-50 blob");
        use_device(device);
        std::unique_ptr<frt_embedder> e(new frt_embedder);
        e->device = device;
        e->max_batch = max_batch;
        e->se = blob.kind == 3;
        e->blob_path = weights_path;
        HIPCHK(hipStreamCreate(&e->stream));
        HIPCHK(hipEventCreateWithFlags(&e->ev_busy[0], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e->ev_busy[1], hipEventDisableTiming));
        e->build(blob);
        HIPCHK(hipDeviceSynchronize());
        *out = e.release();
    });
}

void frt_embedder_destroy(frt_embedder *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamDestroy(e->stream);
    }
    if (e->d_frame) (void)hipFree(e->d_frame);
    for (void *p : e->f32.owned) (void)hipFree(p);
    if (e->f32.done) (void)hipEventDestroy(e->f32.done);
    if (e->h_se_error) (void)hipHostFree(e->h_se_error);
    for (hipEvent_t ev : e->ev_busy)
        if (ev) (void)hipEventDestroy(ev);
    e->arena.release();
    delete e;
}

int frt_embedder_set_se_fused(frt_embedder *e, int enable) {
    return guarded([&] {
        if (!e) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(e->mu);
        e->se_fused = enable != 0;
    });
}

int frt_embedder_set_precision(frt_embedder *e, int fp32) {
    return guarded([&] {
        if (!e) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(e->mu);
        use_device(e->device);
        if (fp32) {
            HIPCHK(hipStreamSynchronize(e->stream));
            e->build_f32();
            HIPCHK(hipDeviceSynchronize());
        }
        e->fp32_mode = fp32 != 0;
    });
}

int frt_embedder_preprocess_face(frt_embedder *e, const uint8_t *bgr_crop, float *chw_out) {
    return guarded([&] {
        if (!e || !bgr_crop || !chw_out) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(e->mu);
        use_device(e->device);
        hipStream_t s = e->stream;
        e->wait_idle(s);
        HIPCHK(hipMemcpyAsync(e->d_crops, bgr_crop, 112 * 112 * 3, hipMemcpyHostToDevice, s));
        launch_face_normalize(e->d_crops, 1, 112, 112, e->d_in, s);
        HIPCHK(hipMemcpyAsync(chw_out, e->d_in, sizeof(float) * 3 * 112 * 112, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_embedder_infer(frt_embedder *e, const float *chw, int batch, float *embeds_out) {
    return guarded([&] {
        if (!e || !chw || !embeds_out || batch < 1) raise(FRT_ERR_INVALID, "doInference: bad argument");
        std::lock_guard<std::mutex> lk(e->mu);
        use_device(e->device);
        hipStream_t s = e->stream;
        e->wait_idle(s);
        const size_t in_elems = (size_t)3 * 112 * 112;
        for (int f0 = 0; f0 < batch; f0 += e->max_batch) {
            const int nf = std::min(e->max_batch, batch - f0);
            HIPCHK(hipMemcpyAsync(e->d_in, chw + (size_t)f0 * in_elems, sizeof(float) * in_elems * nf, hipMemcpyHostToDevice, s));
            e->forward(e->d_in, nf, nullptr, e->d_out, s);
            HIPCHK(hipMemcpyAsync(embeds_out + (size_t)f0 * 512, e->d_out, sizeof(float) * 512 * nf, hipMemcpyDeviceToHost, s));
            sync_stream_spinning(s);
            e->check_se_error();
        }
    });
}

int frt_embedder_forward(frt_embedder *e, const uint8_t *bgr, int rows, int cols, size_t row_stride, const frt_bbox *boxes, int n,
                         float *embeds_out, uint8_t *crops_out) {
    return guarded([&] {
        if (!e || !bgr || !boxes || !embeds_out || n < 0 || rows < 1 || cols < 1) raise(FRT_ERR_INVALID, "forward: bad argument");
        if (n == 0) return;
        std::lock_guard<std::mutex> lk(e->mu);
        use_device(e->device);
        hipStream_t s = e->stream;
        e->wait_idle(s);
        const size_t tight = (size_t)cols * 3, need = (size_t)rows * tight;
        if (need > e->frame_cap) {
            if (e->d_frame) (void)hipFree(e->d_frame);
            e->d_frame = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&e->d_frame), need));
            e->frame_cap = need;
        }
        HIPCHK(hipMemcpy2DAsync(e->d_frame, tight, bgr, row_stride, tight, rows, hipMemcpyHostToDevice, s));
        bool bad = false;
        for (int f0 = 0; f0 < n; f0 += e->max_batch) {
            const int nf = std::min(e->max_batch, n - f0);
            HIPCHK(hipMemcpyAsync(e->d_boxes, boxes + f0, sizeof(frt_bbox) * nf, hipMemcpyHostToDevice, s));
            launch_crop_faces(e->d_frame, rows, cols, tight, 0, e->d_boxes, nullptr, 1, nf, 1, 112, 112, e->d_crops, e->d_in, e->d_valid, s);
            e->forward(e->d_in, nf, e->d_valid, e->d_out, s);
            HIPCHK(hipMemcpyAsync(embeds_out + (size_t)f0 * 512, e->d_out, sizeof(float) * 512 * nf, hipMemcpyDeviceToHost, s));
            if (crops_out) HIPCHK(hipMemcpyAsync(crops_out + (size_t)f0 * 112 * 112 * 3, e->d_crops, (size_t)nf * 112 * 112 * 3, hipMemcpyDeviceToHost, s));
            std::vector<int> valid(nf);
            HIPCHK(hipMemcpyAsync(valid.data(), e->d_valid, sizeof(int) * nf, hipMemcpyDeviceToHost, s));
            sync_stream_spinning(s);
            e->check_se_error();
            for (int v : valid) bad = bad || !v;
        }
        if (bad) raise(FRT_ERR_EMPTY_ROI, "forward: empty or out-of-frame ROI (embedding set to zeros)");
    });
}

int frt_align_faces(const uint8_t *bgr, int rows, int cols, size_t row_stride, const float *landmarks, int n, uint8_t *crops_out, int device) {
    return guarded([&] {
        if (!bgr || !landmarks || !crops_out || n < 0 || rows < 1 || cols < 1) raise(FRT_ERR_INVALID, "alignFaces: bad argument");
        if (n == 0) return;
        if (device >= 0) use_device(device);
        Arena a;
        struct Guard {
            Arena &a;
            ~Guard() { a.release(); }
        } guard{a};
        const size_t tight = (size_t)cols * 3;
        uint8_t *d_frame = a.alloc<uint8_t>((size_t)rows * tight);
        float *d_lm = a.alloc<float>((size_t)n * 10);
        uint8_t *d_crops = a.alloc<uint8_t>((size_t)n * 112 * 112 * 3);
        float *d_chw = a.alloc<float>((size_t)n * 112 * 112 * 3);
        int *d_valid = a.alloc<int>(n);
        HIPCHK(hipMemcpy2D(d_frame, tight, bgr, row_stride, tight, rows, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_lm, landmarks, sizeof(float) * 10 * n, hipMemcpyHostToDevice));
        launch_align_faces(d_frame, rows, cols, tight, 0, d_lm, nullptr, 1, n, 1, d_crops, d_chw, d_valid, nullptr);
        std::vector<int> valid(n);
        HIPCHK(hipMemcpy(valid.data(), d_valid, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(crops_out, d_crops, (size_t)n * 112 * 112 * 3, hipMemcpyDeviceToHost));
        for (int v : valid)
            if (!v) raise(FRT_ERR_EMPTY_ROI, "alignFaces: degenerate landmarks (crop set to zeros)");
    });
}

int frt_embedder_forward_aligned(frt_embedder *e, const uint8_t *bgr, int rows, int cols, size_t row_stride, const float *landmarks, int n,
                                 float *embeds_out, uint8_t *crops_out) {
    return guarded([&] {
        if (!e || !bgr || !landmarks || !embeds_out || n < 0 || rows < 1 || cols < 1) raise(FRT_ERR_INVALID, "forwardAligned: bad argument");
        if (n == 0) return;
        std::lock_guard<std::mutex> lk(e->mu);
        use_device(e->device);
        hipStream_t s = e->stream;
        e->wait_idle(s);
        const size_t tight = (size_t)cols * 3, need = (size_t)rows * tight;
        if (need > e->frame_cap) {
            if (e->d_frame) (void)hipFree(e->d_frame);
            e->d_frame = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&e->d_frame), need));
            e->frame_cap = need;
        }
        HIPCHK(hipMemcpy2DAsync(e->d_frame, tight, bgr, row_stride, tight, rows, hipMemcpyHostToDevice, s));
        bool bad = false;
        for (int f0 = 0; f0 < n; f0 += e->max_batch) {
            const int nf = std::min(e->max_batch, n - f0);
            HIPCHK(hipMemcpyAsync(e->d_lm, landmarks + (size_t)f0 * 10, sizeof(float) * 10 * nf, hipMemcpyHostToDevice, s));
            launch_align_faces(e->d_frame, rows, cols, tight, 0, e->d_lm, nullptr, 1, nf, 1, e->d_crops, e->d_in, e->d_valid, s);
            e->forward(e->d_in, nf, e->d_valid, e->d_out, s);
            HIPCHK(hipMemcpyAsync(embeds_out + (size_t)f0 * 512, e->d_out, sizeof(float) * 512 * nf, hipMemcpyDeviceToHost, s));
            if (crops_out) HIPCHK(hipMemcpyAsync(crops_out + (size_t)f0 * 112 * 112 * 3, e->d_crops, (size_t)nf * 112 * 112 * 3, hipMemcpyDeviceToHost, s));
            std::vector<int> valid(nf);
            HIPCHK(hipMemcpyAsync(valid.data(), e->d_valid, sizeof(int) * nf, hipMemcpyDeviceToHost, s));
            sync_stream_spinning(s);
            e->check_se_error();
            for (int v : valid) bad = bad || !v;
        }
        if (bad) raise(FRT_ERR_EMPTY_ROI, "forwardAligned: degenerate landmarks (embedding set to zeros)");
    });
}


}  // extern "C"
