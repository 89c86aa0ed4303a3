// libfrt.so host side: weight loading/folding, the three objects behind the C ABI (include/frt.h) and the batched pipeline.
// All device work is hand-written HIP (kernels_*.hip); there is no CPU fallback anywhere in this file: without a HIP
// device every entry point that needs one fails with FRT_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "frt_host.hpp"
#include "frt_kernels.h"
#include "frt_weights.hpp"

void launch_pack_results(const frt_bbox *boxes, const int *n_boxes, const int *valid, const int32_t *idx, const float *sim, int max_faces,
                         int F, frt_face_result *out, hipStream_t s);

namespace frthost {
std::string &last_error() {
    static thread_local std::string err;
    return err;
}
}  // namespace frthost
using frthost::guarded;
using frthost::raise;
using frthost::use_device;

namespace {

// ------------------------------------------------------------------------------------------------ device memory helpers
struct Arena {
    std::vector<void *> ptrs;
    template <typename T>
    T *alloc(size_t n) {
        void *p = nullptr;
        HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        ptrs.push_back(p);
        return reinterpret_cast<T *>(p);
    }
    template <typename T>
    T *upload(const std::vector<T> &v) {
        T *d = alloc<T>(v.size());
        HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return d;
    }
    void release() {
        for (void *p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

// Host waits: spin briefly, then poll at a low duty cycle, then block.
//   1. busy-poll hipEventQuery / hipStreamQuery for FRT_WAIT_SPIN_US (default 200 us; frt_set_wait_spin_us): a reply that is about to arrive
//      is picked up without a sleep / wake-up round trip (tens of microseconds on every synchronous call);
//   2. then query once per ~50 us sleep (nanosleep: the thread is off the core in between, ~1 % of a core) for up to 2 s.  The reference's
//      server is .multithreaded() (src/app.cpp:367): every request thread waiting in findFace / forward must not burn a core for the whole
//      GPU latency, which the 50 ms busy-poll of round 3 did;
//   3. then hipEventSynchronize / hipStreamSynchronize (interrupt wait) - an idle pipeline costs nothing.
// Why not (3) at once: in the 20-step benchmark region (one wait every 3 ms) the interrupt wake-up was observed 20 - 30 ms late about once in
// four processes - the GPU finished all three batches in flight while the host slept (profiles/r03/r03v_step_times.txt; polling: 12 of 12
// processes within 1 %, r03w_step_times.txt).  A throughput driver that owns its core may raise the spin (bench.py sets 50 000 and says so).
static std::atomic<long> g_wait_spin_us{-1};
static long wait_spin_us() {
    long v = g_wait_spin_us.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("FRT_WAIT_SPIN_US");
        v = e ? std::max(0L, atol(e)) : 200L;
        g_wait_spin_us.store(v, std::memory_order_relaxed);
    }
    return v;
}
static long wait_poll_us() {
    static const long v = [] {
        const char *e = getenv("FRT_WAIT_POLL_US");
        return e ? std::max(0L, atol(e)) : 2000000L;
    }();
    return v;
}
template <class Query>
static bool spin_until_done(Query &&query) {
    const long spin_us = wait_spin_us(), poll_us = wait_poll_us();
    if (spin_us <= 0 && poll_us <= 0) return false;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed_us = [&] { return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); };
    bool queried = false;
    auto done = [&]() -> bool {
        const hipError_t q = query();
        if (q == hipSuccess) {
            if (queried) (void)hipGetLastError();  // "not ready" is an answer, not an error: do not leave it behind as the thread's last error
            return true;
        }
        if (q != hipErrorNotReady) HIPCHK(q);
        queried = true;
        return false;
    };
    if (spin_us > 0)
        for (int it = 0;; ++it) {
            if (done()) return true;
            if ((it & 15) == 15 && elapsed_us() > spin_us) break;
            __builtin_ia32_pause();
        }
    while (poll_us > 0 && elapsed_us() < spin_us + poll_us) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (done()) return true;
    }
    (void)hipGetLastError();
    return false;
}
static void wait_event_spinning(hipEvent_t ev) {
    if (!spin_until_done([&] { return hipEventQuery(ev); })) HIPCHK(hipEventSynchronize(ev));
}
static void sync_stream_spinning(hipStream_t st) {
    if (!spin_until_done([&] { return hipStreamQuery(st); })) HIPCHK(hipStreamSynchronize(st));
}

// ------------------------------------------------------------------------------------------------ profiling (HIP events)
struct ProfRec {
    std::string name;
    hipEvent_t a, b;
    double work;
};
std::mutex g_prof_mu;
int g_prof_kind = 0;
std::vector<ProfRec> g_prof;
// Events are created when profiling is switched on, not between the two records of a bracket: the first hipEventCreate calls of a process
// take ~ 100 us each (pool set-up), and a host stall between "record a" and the launch it brackets is GPU idle time INSIDE the bracket
// (it showed up as conv_s2c64_kernel - the second bracket of a pass - at 224 us "live" against 97 us in rocprofv3's trace).
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    bool on = false;
    ProfRec rec;
    hipStream_t s;
    ProfScope(int level, const char *name, double work, hipStream_t st) : s(st) {
        if (g_prof_kind != level) return;
        on = true;
        rec.name = name;
        rec.work = work;
        rec.a = prof_event();
        rec.b = prof_event();
        (void)hipEventRecord(rec.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(rec.b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(rec);
    }
};

}  // namespace

// =====================================================================================================================
// Detector
// =====================================================================================================================
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
            if (getenv("FRT_DET_STEM_CHECK")) {  // debugging aid: the three layers one by one against the fused kernel, element for element
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

// =====================================================================================================================
// Embedder
// =====================================================================================================================
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

// =====================================================================================================================
// Matcher
// =====================================================================================================================
struct frt_matcher {
    unsigned generation = 0;  // bumped whenever gallery pointers / sizes / offsets change (invalidates captured graphs)
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    hipEvent_t ev_busy = nullptr;  // end of the last pipeline match stage that used this object's scratch (see wait_idle)
    bool busy = false;
    float *d_gallery = nullptr;  // fp32 rows [N][D]; null when the gallery is STORED as fp16 (store16)
    int N = 0, D = 0;
    int row_offset = 0;  // global index of local row 0 (sharded galleries)
    // scratch (grown on demand)
    float *d_q = nullptr, *d_sim = nullptr, *d_full = nullptr, *d_kth = nullptr;
    int32_t *d_idx = nullptr;
    static constexpr int KCAP = 16;  // d_sim / d_idx hold [q_cap][KCAP] (top-k lists of the host entry point)
    MatchPartial *d_partial = nullptr;
    int q_cap = 0;
    size_t full_cap = 0;
    int blocks = 0;
    // screened top-1 (fp16 shadow gallery; see kernels_match.hip).  Off for small galleries, for widths the coarse kernel is not
    // instantiated for (anything but 64 / 128 / 256 / 512) and with FRT_MATCH_SCREEN=0.
    half_t *d_g16 = nullptr;   // fp16 shadow of d_gallery, or the fp16-STORED gallery itself
    uint8_t *d_g8 = nullptr;   // int8 shadow of d_gallery (round 4: fp32-stored galleries with 512 columns take this instead of the fp16 shadow)
    float *d_g8_scale = nullptr;
    float gerr = 0.f;          // largest quantisation error norm of the int8 rows (part of the screening bound)
    bool store16 = false;      // current gallery is fp16-stored
    bool want16 = false;       // storage mode of the NEXT init / gallery_begin (frt_matcher_set_storage)
    float gmax_norm = 0.f;
    bool screen = false;
    bool screen_on = true;     // frt_matcher_set_screening: false = every top-1 call takes the exact fp32 scan (the shadow gallery stays resident)
    ScreenScratch scr{};
    void free_screen_scratch() {
        for (void *p : {(void *)scr.q16, (void *)scr.tilemax, (void *)scr.tile_flags, (void *)scr.tile_list, (void *)scr.segmax, (void *)scr.wgmax, scr.pairs,
                        (void *)scr.ctl, (void *)scr.qkey})  // scr.count lives behind tile_flags
            if (p) (void)hipFree(p);
        scr = ScreenScratch{};
    }
    // Object-level entry points share the scratch buffers with the pipeline's match stage, which keeps running on the pipeline's
    // stream after frt_pipeline_run_dev / submit returned: order this object's stream behind it (one event wait, no host sync).
    void wait_idle(hipStream_t s) {
        if (busy) HIPCHK(hipStreamWaitEvent(s, ev_busy, 0));
    }

    // ---- streaming gallery load (frt_matcher_gallery_begin / append / commit == initKnownEmbeds / addEmbedding / initMatMul,
    //      src/db.cpp:316-346): rows are copied into pinned staging chunks as they arrive (the caller's pointer may be SQLite's
    //      blob buffer) and every full chunk goes to the device with an asynchronous copy while the next one fills.  The previous
    //      gallery stays live (and searchable) until commit swaps the pointers.
    struct Load {
        static constexpr int NCH = 3;
        static constexpr int CH_ROWS = 4096;   // x 512 floats = 8 MB per chunk
        bool active = false;
        bool f16 = false;
        int cap = 0, D = 0, rows = 0, fill = 0, cur = 0;
        float *d_new32 = nullptr;
        half_t *d_new16 = nullptr;
        float *h_stage[NCH] = {};
        float *d_stage[NCH] = {};   // fp16 storage only: fp32 landing buffers in front of the conversion kernel
        hipEvent_t ev[NCH] = {};
        bool pending[NCH] = {};
        size_t stage_floats = 0;
        hipStream_t s = nullptr;
    } ld;
    void load_release_staging() {
        for (int i = 0; i < Load::NCH; ++i) {
            if (ld.h_stage[i]) (void)hipHostFree(ld.h_stage[i]);
            if (ld.d_stage[i]) (void)hipFree(ld.d_stage[i]);
            if (ld.ev[i]) (void)hipEventDestroy(ld.ev[i]);
            ld.h_stage[i] = ld.d_stage[i] = nullptr;
            ld.ev[i] = nullptr;
            ld.pending[i] = false;
        }
        ld.stage_floats = 0;
    }
    void load_abort() {
        if (ld.s) (void)hipStreamSynchronize(ld.s);
        if (ld.d_new32) (void)hipFree(ld.d_new32);
        if (ld.d_new16) (void)hipFree(ld.d_new16);
        ld.d_new32 = nullptr;
        ld.d_new16 = nullptr;
        ld.active = false;
    }
    void load_begin(int cap, int cols) {
        if (ld.active) load_abort();
        // The load runs on the matcher's own stream.  NOT on a stream of its own: one more hipStreamCreateWithFlags(hipStreamNonBlocking)
        // stream in the process before the pipeline's stage streams exist changes how ROCm maps those onto hardware queues, and the
        // stages of consecutive calls stop overlapping (measured: batch-1 step 0.56 -> 1.39 ms, batch-32 step +10 %).
        ld.s = stream;
        const size_t need = (size_t)Load::CH_ROWS * cols;
        if (ld.stage_floats != need) {
            load_release_staging();
            for (int i = 0; i < Load::NCH; ++i) {
                HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&ld.h_stage[i]), need * sizeof(float), hipHostMallocDefault));
                HIPCHK(hipEventCreateWithFlags(&ld.ev[i], hipEventDisableTiming));
            }
            ld.stage_floats = need;
        }
        ld.f16 = want16;
        ld.cap = cap;
        ld.D = cols;
        ld.rows = ld.fill = ld.cur = 0;
        if (cap > 0) {
            if (ld.f16) {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&ld.d_new16), gallery16_elems(cap, cols) * sizeof(half_t)));
                HIPCHK(hipMemsetAsync(ld.d_new16, 0, gallery16_elems(cap, cols) * sizeof(half_t), ld.s));  // fragment order, zero pad rows
                for (int i = 0; i < Load::NCH; ++i)
                    if (!ld.d_stage[i]) HIPCHK(hipMalloc(reinterpret_cast<void **>(&ld.d_stage[i]), need * sizeof(float)));
            } else {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&ld.d_new32), (size_t)cap * cols * sizeof(float)));
            }
        }
        ld.active = true;
    }
    void load_flush() {  // current chunk -> device
        if (!ld.fill) return;
        const int c = ld.cur;
        const size_t off = (size_t)(ld.rows - ld.fill) * ld.D, n = (size_t)ld.fill * ld.D;
        if (ld.f16) {
            HIPCHK(hipMemcpyAsync(ld.d_stage[c], ld.h_stage[c], n * sizeof(float), hipMemcpyHostToDevice, ld.s));
            launch_rows_to_half(ld.d_stage[c], (long)(ld.rows - ld.fill), (long)ld.fill, ld.D, ld.d_new16, ld.s);  // (chunks start on 128-row tiles)
        } else {
            HIPCHK(hipMemcpyAsync(ld.d_new32 + off, ld.h_stage[c], n * sizeof(float), hipMemcpyHostToDevice, ld.s));
        }
        HIPCHK(hipEventRecord(ld.ev[c], ld.s));
        ld.pending[c] = true;
        ld.cur = (c + 1) % Load::NCH;
        ld.fill = 0;
        if (ld.pending[ld.cur]) {  // the chunk about to be refilled must have left the host (and its landing buffer)
            HIPCHK(hipEventSynchronize(ld.ev[ld.cur]));
            ld.pending[ld.cur] = false;
        }
    }
    void load_append(const float *rows, int n) {
        if (!ld.active) raise(FRT_ERR_INVALID, "gallery_append: no load in progress (call frt_matcher_gallery_begin first)");
        if (n < 0 || (n > 0 && !rows)) raise(FRT_ERR_INVALID, "gallery_append: bad argument");
        if ((long)ld.rows + n > ld.cap) raise(FRT_ERR_CAPACITY, "gallery_append: more rows than gallery_begin reserved (initKnownEmbeds)");
        while (n > 0) {
            const int take = std::min(n, Load::CH_ROWS - ld.fill);
            std::memcpy(ld.h_stage[ld.cur] + (size_t)ld.fill * ld.D, rows, (size_t)take * ld.D * sizeof(float));
            ld.fill += take;
            ld.rows += take;
            rows += (size_t)take * ld.D;
            n -= take;
            if (ld.fill == Load::CH_ROWS) load_flush();
        }
    }
    // make the loaded rows THE gallery: swap pointers, rebuild the screening data, free the previous gallery
    void load_commit() {
        if (!ld.active) raise(FRT_ERR_INVALID, "gallery_commit: no load in progress");
        load_flush();
        HIPCHK(hipStreamSynchronize(ld.s));
        for (bool &p : ld.pending) p = false;
        HIPCHK(hipStreamSynchronize(stream));
        if (busy) HIPCHK(hipEventSynchronize(ev_busy));
        float *old32 = d_gallery;
        half_t *old16 = d_g16;
        if (d_g8) (void)hipFree(d_g8);  // (the streams were synchronised above: no scan is reading it)
        if (d_g8_scale) (void)hipFree(d_g8_scale);
        d_g8 = nullptr;
        d_g8_scale = nullptr;
        gerr = 0.f;
        ++generation;
        N = ld.rows;
        D = ld.D;
        store16 = ld.f16;
        d_gallery = ld.rows > 0 ? ld.d_new32 : nullptr;
        d_g16 = ld.rows > 0 ? ld.d_new16 : nullptr;
        if (ld.rows == 0) {  // empty gallery: nothing to keep
            if (ld.d_new32) (void)hipFree(ld.d_new32);
            if (ld.d_new16) (void)hipFree(ld.d_new16);
        }
        ld.d_new32 = nullptr;
        ld.d_new16 = nullptr;
        ld.active = false;
        if (old32) (void)hipFree(old32);  // (hipFree waits for the device: stages of earlier pipeline calls have finished with it)
        if (old16) (void)hipFree(old16);
        blocks = match_top1_blocks(N, 0);
        const char *scr_env = frt_tuning_env("FRT_MATCH_SCREEN");  // (tuning build only; the product's switch is frt_matcher_set_screening)
        screen = N >= 32768 && match_screen_supported(D) && !(scr_env && scr_env[0] == '0');
        gmax_norm = 0.f;
        if (N > 0 && (screen || store16)) {  // fp16 shadow copy (fp32 storage) + the largest row norm (rounding bound of the screening pass)
            int *d_bits = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_bits), sizeof(int)));
            // fp32-stored galleries of 512 columns are screened through an INT8 shadow (half the bytes of the per-call scan; kernels_match.hip);
            // FRT_MATCH_I8=0 / FRT_MATCH_FAST=0 keep the fp16 shadow (A/B measurements, the round-2 tile-list path)
            const char *i8_env = frt_tuning_env("FRT_MATCH_I8"), *fast_env = frt_tuning_env("FRT_MATCH_FAST");
            const bool use_i8 = screen && !store16 && D == 512 && !(i8_env && i8_env[0] == '0') && !(fast_env && fast_env[0] == '0');
            int *d_ebits = nullptr;
            if (store16) {
                launch_gallery_norm16(d_g16, N, D, d_bits, stream);
            } else if (use_i8) {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_ebits), sizeof(int)));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_g8), gallery8_bytes(N, D)));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_g8_scale), ((size_t)(N + 127) / 128) * 128 * sizeof(float)));
                launch_gallery_shadow8(d_gallery, N, D, d_g8, d_g8_scale, d_ebits, d_bits, stream);
            } else {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_g16), gallery16_elems(N, D) * sizeof(half_t)));
                launch_gallery_shadow(d_gallery, N, D, d_g16, d_bits, stream);
            }
            int bits = 0, ebits = 0;
            HIPCHK(hipMemcpyAsync(&bits, d_bits, sizeof(int), hipMemcpyDeviceToHost, stream));
            if (d_ebits) HIPCHK(hipMemcpyAsync(&ebits, d_ebits, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            (void)hipFree(d_bits);
            if (d_ebits) (void)hipFree(d_ebits);
            float n2, e2;
            std::memcpy(&n2, &bits, 4);
            std::memcpy(&e2, &ebits, 4);
            gmax_norm = std::sqrt(n2);
            gerr = std::sqrt(e2);
        }
        q_cap = 0;  // partial scratch depends on `blocks`
        if (d_partial) {
            (void)hipFree(d_partial);
            d_partial = nullptr;
        }
    }

    void ensure_queries(int F) {
        if (F <= q_cap && d_partial) return;
        const int cap = std::max(F, 128);
        ++generation;  // scratch buffers move
        if (d_q) (void)hipFree(d_q);
        if (d_sim) (void)hipFree(d_sim);
        if (d_idx) (void)hipFree(d_idx);
        if (d_kth) (void)hipFree(d_kth);
        if (d_partial) (void)hipFree(d_partial);
        d_q = d_sim = d_kth = nullptr;
        d_idx = nullptr;
        d_partial = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_q), (size_t)cap * D * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_sim), (size_t)cap * KCAP * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_idx), (size_t)cap * KCAP * sizeof(int32_t)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_kth), (size_t)cap * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_partial), (size_t)blocks * cap * sizeof(MatchPartial)));
        if (screen) {
            const size_t tiles = ((size_t)N + 127) / 128;
            free_screen_scratch();
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.q16), (size_t)cap * D * sizeof(half_t)));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.tilemax), (size_t)cap * tiles * 4 * sizeof(float)));  // 4 coarse entries per tile (one per wave)
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.tile_flags), (tiles + 1) * sizeof(int)));  // [tiles] flags + the candidate count:
            scr.count = scr.tile_flags + tiles;                                                           // one contiguous range to clear per call
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.tile_list), tiles * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.segmax), (size_t)cap * 16 * sizeof(float)));
            const char *fe = frt_tuning_env("FRT_MATCH_FAST");  // "0": the round-2 tile-list re-rank (diagnostics, tuning build)
            if (!(fe && fe[0] == '0')) {
                scr.pair_cap = std::max(cap * 64, 8192);  // (a multiple of the 16 sub-lists)
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.wgmax), (size_t)256 * cap * sizeof(float)));
                HIPCHK(hipMalloc(&scr.pairs, (size_t)scr.pair_cap * 8));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.ctl), 32 * sizeof(int)));  // CTL_WORDS (kernels_match.hip)
                HIPCHK(hipMemset(scr.ctl, 0, 32 * sizeof(int)));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.qkey), (size_t)cap * sizeof(unsigned long long)));
            }
        }
        q_cap = cap;
    }
    // queries_dev [F][D] -> idx_dev, sim_dev (device pointers)
    void top1_dev(const float *queries_dev, int F, int32_t *idx_dev, float *sim_dev, hipStream_t s) {
        ProfScope ps(2, "match_top1", 2.0 * D * (double)N * F, s);
        // the partial scratch is [blocks][F]
        if (screen && screen_on) {  // (d_gallery == nullptr with fp16 storage: the exact re-rank then reads the stored fp16 rows)
            ScreenScratch w = scr;
            w.g8 = d_g8;
            w.g8_scale = d_g8_scale;
            w.gerr = gerr;
            launch_match_top1_screened(d_gallery, d_g16, N, D, queries_dev, F, gmax_norm, w, d_partial, blocks, idx_dev, sim_dev, row_offset, s);
        }
        else if (store16)
            launch_match_top1_h(d_g16, N, D, queries_dev, F, d_partial, blocks, idx_dev, sim_dev, row_offset, s);
        else
            launch_match_top1(d_gallery, N, D, queries_dev, F, d_partial, blocks, idx_dev, sim_dev, row_offset, s);
        HIPCHK(hipGetLastError());
    }
    // exact top-k lists [F][k] (idx_dev / sim_dev device pointers); queries fp32 on the device
    void topk_dev(const float *queries_dev, int F, int k, int32_t *idx_dev, float *sim_dev, hipStream_t s) {
        ProfScope ps(2, "match_topk", 2.0 * D * (double)N * F, s);
        ScreenScratch w = scr;
        w.g8 = d_g8;
        w.g8_scale = d_g8_scale;
        w.gerr = gerr;
        launch_match_topk(d_gallery, d_g16, N, D, queries_dev, F, k, screen, gmax_norm, w, d_kth, d_partial, blocks, idx_dev, sim_dev, row_offset, s);
        HIPCHK(hipGetLastError());
    }
};

// =====================================================================================================================
// Pipeline
// =====================================================================================================================
struct frt_pipeline {
    frt_detector *det;
    frt_embedder *emb;
    frt_matcher *mat;
    int max_frames, max_faces, F_cap;
    hipStream_t stream = nullptr, own_stream = nullptr;
    // Three-stage software pipeline over consecutive calls: detector of call b+1 (det_stream), crop + recogniser of call b
    // (emb_stream / emb_stream2 alternately), match + pack of call b-1 (behind its recogniser pass on the same stream, i.e. beside
    // the other set's pass) - three stages with different bottlenecks (latency / MFMA+LDS / HBM) that overlap on the same CUs.  `stream` (the caller's) only joins.  Fork/join with events; boxes, embeddings and validity
    // flags of a call live in one of two slots so that a later stage of the previous call can still read them.
    hipStream_t det_stream = nullptr, emb_stream = nullptr, emb_stream2 = nullptr;
    bool dual_embed = true;   // recogniser passes of consecutive calls on two streams with two activation sets (FRT_PIPELINE_DUAL_EMBED=0: one)
    float *d_chw2 = nullptr;
    hipEvent_t ev_serial = nullptr;  // end of the last serial (profiled) call while overlap is on
    bool serial_pending = false;
    // calls in flight between the start of D and the end of M (2: 32.5k, 3: 33.6k, 4: 33.6k faces/s).  Six since pairing exists: paired calls finish
    // two at a time and one call late, so two pairs in the later stages + the detector a call or two ahead need six slots (with three the detector
    // of call b + 5 waited for the pair (b + 2, b + 3) and the recogniser passes ran one after the other: 0.84 instead of 0.74 ms per 4-frame call)
    static constexpr int NSLOT = 10;  // (groups of four: 2 * 4 + 2)
    hipEvent_t ev_det[NSLOT] = {}, ev_emb[NSLOT] = {}, ev_done[NSLOT] = {};
    float *slot_embeds[NSLOT] = {};
    int *slot_valid[NSLOT] = {};
    frt_bbox *slot_boxes[NSLOT] = {};
    int *slot_nout[NSLOT] = {};
    float *slot_landmarks[NSLOT] = {};
    bool align = false;  // optional: 5-point similarity warp instead of the reference's bbox crop + bicubic resize
    unsigned seq = 0;
    bool overlap = true;
    bool serial_call = false;  // this call only: every stage on the caller's stream (a synchronous call with nothing else in flight, see frt_pipeline_run)
    Arena arena;
    float *d_chw, *d_sim;
    int32_t *d_idx;
    std::mutex run_mu;               // serialises run(): stream selection, slot counters and the stage enqueue order are per-call state
    bool input_sync = false;         // frt_pipeline_set_input_sync: order every run_dev call behind the work queued on `stream` so far
    hipEvent_t ev_input = nullptr;   // ... recorded on `stream` at the call
    hipEvent_t ev_ready = nullptr;   // caller's "frames are ready" event of frt_pipeline_run_dev_after (borrowed, one call)

    // ---- asynchronous host boundary (frt_pipeline_submit / frt_pipeline_wait): NBUF staging sets so that the H2D copy of batch
    //      b+1 (copy_stream, the SDMA engine) and the D2H of batch b-1 run under the stages of batch b
    static constexpr int NBUF = 12;  // (4 until pairing: up to eleven batches between submit and wait)
    struct AsyncBuf {
        uint8_t *d_frames = nullptr;
        frt_face_result *d_results = nullptr;
        float *d_embeds = nullptr;
        uint8_t *d_crops = nullptr;  // u8 BGR 112x112 crops of the batch's faces (frt_pipeline_submit_crops)
        hipEvent_t ev_h2d = nullptr, ev_out = nullptr;
        long ticket = -1;  // ticket whose results ev_out guards; -1: never used
        std::atomic<bool> failed{false};  // the held stages of this ticket could not be queued (flush_pending / start_held): frt_pipeline_wait reports it
    };
    AsyncBuf abuf[NBUF];
    hipStream_t copy_stream = nullptr;
    int copy_prio = 0;
    hipEvent_t ev_frames = nullptr;  // set by submit for the next run(): the detector stream waits for it
    uint8_t *crops_req = nullptr;    // set by submit for the next run(): the crop kernel also writes the u8 crops there
    long next_ticket = 0;
    std::mutex async_mu;

    // ---- pairing (frt_pipeline_set_pairing; off by default).  A recogniser pass over 16 faces costs 0.59 ms, one over 32 faces 0.92 ms
    //      (profiles/r05p_small_batch_layers.txt: below ~ 64 faces a pass is a chain of launch latencies, not work), and one match call scans
    //      the gallery once whatever the number of queries.  With pairing on, the crop + recogniser + match stages of TWO consecutive calls
    //      run as one pass: a call's detector stage is queued at the call as always, its later stages wait for the next call (or for a
    //      flush: frt_pipeline_wait on its ticket, frt_pipeline_sync, any mode switch).  Nothing about a result changes except when it is
    //      ready - one call later - and which batch-size class of recogniser kernels produced it (the class of the two calls' faces together).
    //      A call is only ever deferred when it could be paired: both calls' face slots together must fit this pipeline's max_frames *
    //      max_faces and the recogniser's max_batch, i.e. create the pipeline for twice the frames a call carries.
    struct Sub {  // one frt_pipeline_submit ticket inside a call: `n` frames, where its results go, the staging set that carries its events
        AsyncBuf *ab = nullptr;
        frt_face_result *h_results = nullptr;
        float *h_embeds = nullptr;
        uint8_t *h_crops = nullptr;
        int n = 0;
        long ticket = -1;
    };
    static constexpr int MAXSUB = 4;
    struct CallRec {
        bool on = false;
        unsigned call = 0;
        int slot = 0, n = 0;
        const uint8_t *frames = nullptr;
        frt_face_result *results = nullptr;
        float *embeds = nullptr;
        uint8_t *crops = nullptr;
        // host side of frt_pipeline_submit: the downloads of this call's results follow its match stage, wherever that is queued.  One entry per
        // ticket: a call is the frames of up to MAXSUB consecutive submits when they were merged at the host boundary (Held, below)
        int nsub = 0;
        Sub sub[MAXSUB];
    };
    static constexpr int MAXG = 4;  // calls per recogniser pass at most
    CallRec pend[MAXG];  // the calls whose later stages are still to be queued (fewer than `group` of them)
    int npend = 0;
    CallRec host_req;  // set by submit for the next run(): staging set + host destinations
    // group: 0 off; 2 .. MAXG: ALWAYS wait for that many calls per recogniser pass (results up to group - 1 calls late);
    //        -1 (default, round 6) ADAPTIVE: a call's later stages are held back only while the recogniser is still busy with earlier calls -
    //        the pass could not start now anyway, so waiting for the next call costs a lone caller nothing - and go out together with the
    //        next call's (up to MAXG calls per pass while the backlog lasts).  A call that finds the recogniser idle is queued at once,
    //        exactly like group == 0.  frt_pipeline_wait on ANY ticket releases held calls once the recogniser has gone idle.
    //        Only calls that came through frt_pipeline_submit are held (their contract is the ticket); frt_pipeline_run_dev promises that
    //        the pipeline stream joins the results AT the call, so device-resident calls are held only on request (adaptive_dev).
    int group = -1;
    bool adaptive_dev = false;
    bool merge_submits = true;   // adaptive mode: merge held submits at the host boundary (Held, below)
    // ---- merging of consecutive submits at the host boundary (adaptive mode only, round 6).  A small call's DETECTOR stage is as much a chain of
    //      launch latencies as its recogniser pass (4 frames: 258 us of kernels, 32 frames: 809 - profiles/r06g_det_tables.txt).  A submit that
    //      finds the pipeline backed up (detector or recogniser still busy with earlier calls) is not queued at all: its frames are uploaded into its staging set and the call is
    //      HELD; the next submit's frames go into the same staging set behind them, and the held frames then run as ONE call (one detector pass,
    //      one recogniser pass, one match call; per-ticket result downloads).  Released by: the submit that fills it (MAXSUB tickets / the
    //      pipeline's capacity) or finds the detector idle, any submit that cannot join, frt_pipeline_wait on one of its tickets - or on any
    //      ticket once the detector has gone idle -, run_dev, sync, set_*, destroy.  A call that finds the detector idle is never held.
    struct Held {
        bool on = false;
        AsyncBuf *base = nullptr;  // the staging set that holds the frames / results / embeddings / crops of every ticket of the call
        int n = 0, nsub = 0;
        bool want_embeds = false, want_crops = false;
        Sub sub[MAXSUB];
    } held;
    long merged_calls = 0, merged_tickets = 0;
    std::string held_error;  // why the last held call could not be queued
    bool detector_busy() const { return det->busy && hipEventQuery(det->ev_busy) == hipErrorNotReady; }
    // "backed up": a stage of an earlier call is still running or queued.  (The detector alone is the wrong signal: under load the recogniser is
    // the bottleneck and the detector is often idle at the moment of a submit - single 4-frame calls then slip in between the merged ones:
    // measured 0.559 ms per 4-frame step against 0.524 without any merging.)
    bool backed_up() const { return detector_busy() || recogniser_busy(); }
    // tickets whose stages are queued and whose results have not left yet (held / pending ones are not counted: nothing of theirs is queued)
    int tickets_running() const {
        int n = 0;
        for (const AsyncBuf &b : abuf) {
            if (b.ticket < 0 || is_pending(b.ticket)) continue;
            bool h = false;
            for (int j = 0; held.on && j < held.nsub; ++j) h = h || held.sub[j].ticket == b.ticket;
            if (!h && hipEventQuery(b.ev_out) == hipErrorNotReady) ++n;
        }
        return n;
    }
    // Holding is only free while the GPU has enough queued work to stay busy until the held frames are released: a submit is held only when at
    // least HOLD_MIN tickets are running, and the held ones go out as soon as fewer are.  Measured with 4-frame calls (tools/proxy_only.py, ms
    // per call; pairing off 0.71 - 0.72 at every depth): without the threshold 2 / 3 / 4 calls in flight cost 0.98 / 0.86 / 0.74 - a caller that
    // keeps few calls in flight is latency-coupled to each of them; with HOLD_MIN = 5 and the release rule: <= 5 in flight as without pairing,
    // 6: 0.58, 7: 0.55, 8: 0.52, 11: 0.49 (profiles/r06_adaptive_hold_sweep.txt).
    static constexpr int HOLD_MIN = 5;
    unsigned epass = 0;  // recogniser passes queued so far (activation set / stream of the next one)
    long paired_passes = 0, single_passes = 0;
    // is a recogniser pass queued earlier still running (or waiting to run)?  Two event queries, ~ 1 us each
    bool recogniser_busy() const {
        for (int k = 0; k < 2; ++k)
            if (emb->busy[k] && hipEventQuery(emb->ev_busy[k]) == hipErrorNotReady) return true;
        return false;
    }
    void ensure_async() {
        if (copy_stream) return;
        // The upload stream sits in the stage streams' priority class (its own hardware-queue pool): as a normal-priority stream it is
        // dealt round-robin onto the four queues the CALLER's streams live on, and whenever it lands on the queue of the caller's joining
        // stream the next batch's upload sits behind the pending joins of the batches in flight (measured with RCCL's streams in the
        // process: 4-frame step 0.96 -> 1.69 ms, 32-frame step 3.28 -> 3.45 ms).  copy_prio: see frt_pipeline_create.
        HIPCHK(hipStreamCreateWithPriority(&copy_stream, hipStreamNonBlocking, copy_prio));
        const size_t F = (size_t)F_cap;
        for (AsyncBuf &b : abuf) {
            b.d_frames = arena.alloc<uint8_t>((size_t)max_frames * det->g.frame_h * det->g.frame_w * 3);
            b.d_results = arena.alloc<frt_face_result>(F);
            b.d_embeds = arena.alloc<float>(F * 512);
            b.d_crops = arena.alloc<uint8_t>(F * 112 * 112 * 3);
            HIPCHK(hipEventCreateWithFlags(&b.ev_h2d, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&b.ev_out, hipEventDisableTiming));
        }
    }

    // ---- hipGraph replay.  A step is ~150 dependent launches; eager dispatch costs 3.1 us per dependent kernel on this part,
    //      a graph replay 1.8 us (tools/ubench/launch_gap.hip).  Each call is two graphs - the detector part on det_stream, the
    //      rest on `stream` - so the cross-call overlap of the two streams survives; the fork/join events stay ordinary stream
    //      operations between the graph launches.  A part is keyed by everything baked into its nodes (buffers, batch, slot,
    //      mode, gallery generation); first sighting of a key runs eagerly (lazy one-time setup inside the launchers), the second
    //      is captured, later ones replay.  Off while the profiling hooks record events (frt_profile_enable).
    //      OPT-IN (FRT_PIPELINE_GRAPH=1 / frt_pipeline_set_graph(p, 1)): on the benchmark step the replay measured 4.41 ms against
    //      4.39 ms eager - the launches are queued far enough ahead that the per-dispatch cost hides behind the previous kernel.
    struct GraphKey {
        int part;
        const void *frames, *results, *embeds;
        int n, slot, align;
        unsigned gallery_gen;
        bool operator==(const GraphKey &o) const {
            return part == o.part && frames == o.frames && results == o.results && embeds == o.embeds && n == o.n && slot == o.slot && align == o.align &&
                   gallery_gen == o.gallery_gen;
        }
    };
    struct GraphEntry {
        GraphKey key;
        int seen = 0;
        hipGraphExec_t exec = nullptr;
        unsigned long used = 0;  // tick of the last sighting (least-recently-used eviction)
    };
    std::vector<GraphEntry> graphs;
    bool use_graphs = false;
    // a steady pipelined workload cycles through NSLOT keys of stage 0, up to 2 * NSLOT (slot, activation set) pairs of stage 1 and NSLOT of
    // stage 2: the cache holds them all (a smaller one evicted every key before it recurred - nothing was ever replayed)
    static constexpr size_t GRAPH_CAP = 4 * NSLOT + 8;
    unsigned long graph_tick = 0;
    long graphs_captured = 0, graphs_replayed = 0;
    void drop_graphs() {
        for (GraphEntry &e : graphs)
            if (e.exec) (void)hipGraphExecDestroy(e.exec);
        graphs.clear();
    }
    template <typename Body>
    void run_part(const GraphKey &key, hipStream_t st, Body body) {
        if (!use_graphs || g_prof_kind != 0 || serial_call) return body(st);
        GraphEntry *e = nullptr;
        for (GraphEntry &g : graphs)
            if (g.key == key) e = &g;
        if (!e) {
            if (graphs.size() >= GRAPH_CAP) {  // callers that never repeat their buffers: bounded memory - the least recently seen key goes
                size_t lru = 0;
                for (size_t i = 1; i < graphs.size(); ++i)
                    if (graphs[i].used < graphs[lru].used) lru = i;
                if (graphs[lru].exec) (void)hipGraphExecDestroy(graphs[lru].exec);
                graphs.erase(graphs.begin() + (long)lru);
            }
            graphs.push_back(GraphEntry{key, 0, nullptr, 0});
            e = &graphs.back();
        }
        e->used = ++graph_tick;
        if (e->exec) {
            HIPCHK(hipGraphLaunch(e->exec, st));
            ++graphs_replayed;
            return;
        }
        if (e->seen++ == 0) return body(st);
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        try {
            body(st);
        } catch (...) {
            (void)hipStreamEndCapture(st, &g);
            if (g) (void)hipGraphDestroy(g);
            throw;
        }
        HIPCHK(hipStreamEndCapture(st, &g));
        const hipError_t ie = hipGraphInstantiate(&e->exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) {
            e->exec = nullptr;
            HIPCHK(ie);
        }
        ++graphs_captured;
        HIPCHK(hipGraphLaunch(e->exec, st));
    }

    // ---- stream-overlap self-check.  The three-stage pipeline only overlaps when its stage streams (and the caller's joining stream)
    //      sit on different hardware queues: ROCm maps streams round-robin onto GPU_MAX_HW_QUEUES (4) queues per priority level, a
    //      queue is in-order, and one extra stream created before the pipeline has been seen to cost 7 % - 2.5x.
    //      Measured, not assumed: one 150 us single-wave spin kernel per stream, started together; `ratio` = elapsed / 150 us is ~1 when
    //      they run side by side and ~n when n streams share a queue.
    std::string warning;      // last self-check verdict ("" = fine); frt_pipeline_check_overlap returns it through frt_last_error
    float overlap_ratio = 0.f;
    float check_streams(const std::vector<hipStream_t> &sts, double us = 150.0) {
        std::vector<hipEvent_t> a(sts.size()), b(sts.size());
        for (size_t i = 0; i < sts.size(); ++i) {
            HIPCHK(hipEventCreate(&a[i]));
            HIPCHK(hipEventCreate(&b[i]));
            HIPCHK(hipStreamSynchronize(sts[i]));
        }
        for (size_t i = 0; i < sts.size(); ++i) launch_spin(5.0, sts[i]);  // first use of the kernel: code load off the clock
        for (size_t i = 0; i < sts.size(); ++i) HIPCHK(hipStreamSynchronize(sts[i]));
        float worst = 0.f;
        for (int rep = 0; rep < 3; ++rep) {  // best of three: a late host thread inflates a run, nothing deflates it
            for (size_t i = 0; i < sts.size(); ++i) {
                HIPCHK(hipEventRecord(a[i], sts[i]));
                launch_spin(us, sts[i]);
                HIPCHK(hipEventRecord(b[i], sts[i]));
            }
            for (size_t i = 0; i < sts.size(); ++i) HIPCHK(hipEventSynchronize(b[i]));
            float span = 0.f;  // first start -> last end
            for (size_t i = 0; i < sts.size(); ++i) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, a[0], b[i]));
                span = std::max(span, ms);
            }
            const float r = span * 1e3f / (float)us;
            worst = rep == 0 ? r : std::min(worst, r);
        }
        for (size_t i = 0; i < sts.size(); ++i) {
            (void)hipEventDestroy(a[i]);
            (void)hipEventDestroy(b[i]);
        }
        return worst;
    }
    // Does a wait that is PENDING on stream `j` hold up work on stream `x`?  That is what sharing a hardware queue means for this
    // pipeline: kernels of two streams multiplexed onto one queue may still run side by side, but a queue is in-order, so the caller's
    // stream - on which every call leaves "wait for the end of my match stage" - blocks whatever stream shares its queue until that
    // call has finished, and consecutive calls serialise.  Test: a 400 us probe kernel on `g`, an event behind it that `j` waits for,
    // then a 20 us probe on `x`: finished long before the gate opens (ratio << 1) or only behind it (>= 1).
    float blocked_by_wait(hipStream_t j, hipStream_t x, hipStream_t g) {
        hipEvent_t e0, gate, xb;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&gate));
        HIPCHK(hipEventCreate(&xb));
        for (hipStream_t st : {j, x, g}) HIPCHK(hipStreamSynchronize(st));
        float worst = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            HIPCHK(hipEventRecord(e0, g));
            launch_spin(400.0, g);
            HIPCHK(hipEventRecord(gate, g));
            HIPCHK(hipStreamWaitEvent(j, gate, 0));
            launch_spin(20.0, x);
            HIPCHK(hipEventRecord(xb, x));
            HIPCHK(hipEventSynchronize(xb));
            HIPCHK(hipStreamSynchronize(j));
            HIPCHK(hipStreamSynchronize(g));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, xb));
            const float r = ms / 0.4f;
            worst = rep == 0 ? r : std::min(worst, r);
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(gate);
        (void)hipEventDestroy(xb);
        return worst;
    }

    void self_check(bool with_caller) {
        std::vector<hipStream_t> sts = {det_stream, emb_stream, emb_stream2};
        std::vector<const char *> names = {"detector", "recogniser", "recogniser-2"};
        if (with_caller) {
            if (stream) {
                sts.push_back(stream);
                names.push_back("caller");
            }
            if (copy_stream) {
                sts.push_back(copy_stream);
                names.push_back("upload");
            }
        }
        overlap_ratio = check_streams({det_stream, emb_stream, emb_stream2});
        warning.clear();
        if (with_caller && stream) {  // the hazard proper: a pending join on the caller's stream must not hold up a pipeline stream
            std::string held;
            struct X {
                hipStream_t st;
                const char *name;
                hipStream_t gate_on;
            } xs[] = {{copy_stream, "upload", emb_stream2}, {det_stream, "detector", emb_stream2}, {emb_stream, "recogniser", emb_stream2},
                      {emb_stream2, "recogniser-2", emb_stream}};
            for (const X &x : xs) {
                if (!x.st) continue;
                const float r = blocked_by_wait(stream, x.st, x.gate_on);
                if (r > 0.8f) held += std::string(held.empty() ? "" : ", ") + x.name;
            }
            if (!held.empty()) {
                char buf[768];
                snprintf(buf, sizeof(buf),
                         "frt_pipeline: a wait pending on the caller's stream holds up the pipeline's %s stream(s) - they share a hardware queue, so "
                         "every call's final join blocks the next call and consecutive batches serialise (measured: 4-frame step 0.78 -> 1.65 ms).  "
                         "Hand the pipeline another stream (a newly created one lands on another queue) and check again; see INTEGRATION.md "
                         "'Streams and hardware queues'.",
                         held.c_str());
                warning = buf;
                overlap_ratio = std::max(overlap_ratio, 2.0f);
                if (!getenv("FRT_QUIET")) fprintf(stderr, "[libfrt] warning: %s\n", buf);
                return;
            }
        }
        if (overlap_ratio > 1.5f) {
            // which two?  pairwise probes (only on the failing path: 3 x 150 us per pair)
            std::string pairs;
            for (size_t i = 0; i < sts.size(); ++i)
                for (size_t j = i + 1; j < sts.size(); ++j)
                    if (check_streams({sts[i], sts[j]}) > 1.5f) pairs += std::string(pairs.empty() ? "" : ", ") + names[i] + " + " + names[j];
            char buf[768];
            snprintf(buf, sizeof(buf),
                     "frt_pipeline: %zu streams of the stage pipeline do not run side by side (150 us probe kernels took %.2fx as long together as "
                     "alone; sharing a hardware queue: %s): consecutive batches will not overlap.  Create the pipeline - and hand it the "
                     "caller's stream - before the process's other HIP streams (RCCL, codec, copy streams) are created or first used, keep "
                     "GPU_MAX_HW_QUEUES at its default 4, see INTEGRATION.md 'Streams and hardware queues'.",
                     sts.size(), overlap_ratio, pairs.empty() ? "?" : pairs.c_str());
            warning = buf;
            if (!getenv("FRT_QUIET")) fprintf(stderr, "[libfrt] warning: %s\n", buf);
        }
    }

    void ensure_stream() {
        if (!stream) {
            if (!own_stream) HIPCHK(hipStreamCreate(&own_stream));
            stream = own_stream;
        }
    }
    void run(const uint8_t *frames_dev, int n, frt_face_result *results_dev, float *embeds_dev) {
        ensure_stream();
        hipStream_t s = stream;
        const DetGeom &g = det->g;
        const int F = n * max_faces;
        // Three-stage software pipeline over consecutive calls (stage-profiling mode and overlap off: everything serially on `s`):
        //   D  detector of call b+1          (fp32 / split-fp16 MFMA + latency-bound stencils)
        //   E  crop + recogniser of call b   (fp16 MFMA / LDS bound)
        //   M  match + pack of call b-1      (HBM bound: streams the 1 GB fp16 shadow gallery)
        // The caller's stream only JOINS: it waits for M of this call, so everything the caller enqueues after the call sees the
        // results, exactly as if the call had run on that stream.  Boxes, embeddings and validity flags live in NSLOT slots.
        // Profiled calls (frt_profile_enable 1 or 2) run serially on `s`: HIP events around a launch only measure the kernel when
        // no other stream competes for the dispatch (with four streams in flight the bracketed time was 2.7x the kernel time).
        const bool pipe3 = overlap && g_prof_kind == 0 && !serial_call;
        // pairing: this call's later stages wait for the next call - or run together with the waiting call's
        const int gcap = std::min({group < 0 ? (int)MAXG : group, F_cap / F, emb->max_batch / F});  // calls of this size one pass can take
        const bool pairable = pipe3 && gcap >= 2 && (group > 0 || (group < 0 && (host_req.nsub || adaptive_dev)));
        if (npend && !(pairable && pend[0].n == n)) flush_pending();
        const unsigned call = seq++;
        const int slot = (int)(call % NSLOT);
        if (pipe3 && serial_pending) {  // a serial call used the shared detector / recogniser buffers on `s`: order the stages behind it
            HIPCHK(hipStreamWaitEvent(det_stream, ev_serial, 0));
            HIPCHK(hipStreamWaitEvent(emb_stream, ev_serial, 0));
            HIPCHK(hipStreamWaitEvent(emb_stream2, ev_serial, 0));
            serial_pending = false;
        }
        hipStream_t ds = pipe3 ? det_stream : s;
        if (pipe3 && call >= (unsigned)NSLOT) {
            // slot buffers are free again once M of the call NSLOT back is done.  NB the frames must be valid when the call is made:
            // making D wait for prior work on `s` would serialise the stages.
            HIPCHK(hipStreamWaitEvent(ds, ev_done[slot], 0));
        }
        if (ev_frames) {  // frt_pipeline_submit: the frames arrive on the copy stream
            HIPCHK(hipStreamWaitEvent(ds, ev_frames, 0));  // (crop + recogniser follow the detector through ev_det[slot])
            ev_frames = nullptr;
        }
        if (ev_ready) {  // frt_pipeline_run_dev_after: the caller's producer (upload / decode / resize on any stream) signals this event
            HIPCHK(hipStreamWaitEvent(ds, ev_ready, 0));
            ev_ready = nullptr;
        }
        if (input_sync && pipe3) {  // safe mode: everything queued on the caller's stream before this call happens-before the stages
            HIPCHK(hipEventRecord(ev_input, s));
            HIPCHK(hipStreamWaitEvent(ds, ev_input, 0));
        }
        CallRec cur = host_req;  // (staging set + host destinations when the call came through frt_pipeline_submit)
        host_req = CallRec{};
        cur.on = true;
        cur.call = call;
        cur.slot = slot;
        cur.n = n;
        cur.frames = frames_dev;
        cur.results = results_dev;
        cur.embeds = embeds_dev;
        cur.crops = crops_req;  // (one call only)
        crops_req = nullptr;
        const int akey = (align ? 1 : 0) | (cur.crops ? 2 : 0);
        run_part(GraphKey{0, frames_dev, nullptr, nullptr, n, slot, akey, 0u}, ds, [&](hipStream_t st) {
#ifdef FRT_TUNING
            // timing build: FRT_PIPE_ABLATE bit 0 = no detector network after the first calls (post-processing re-reads the old head outputs),
            // bit 1 = no recogniser network, bit 2 = no match: what each stage costs the pipelined step (profiles/r04/r04s_stage_ablation.txt)
            static const int pipe_abl = getenv("FRT_PIPE_ABLATE") ? atoi(getenv("FRT_PIPE_ABLATE")) : 0;
            if (!(pipe_abl & 1) || call < 8u)
#endif
            det->forward_frames(frames_dev, n, (size_t)g.frame_w * 3, (size_t)g.frame_w * g.frame_h * 3, st);
            det->postprocess(n, st, slot_boxes[slot], slot_nout[slot], slot_landmarks[slot]);  // straight into this call's slot
        });
        HIPCHK(hipEventRecord(det->ev_busy, ds));  // object-level detector calls wait for this (frt_detector::wait_idle)
        det->busy = true;
        if (pipe3) HIPCHK(hipEventRecord(ev_det[slot], ds));
        if (pairable) {
            // adaptive: hold this call back only while earlier recogniser passes are still in flight; fixed groups: always
            const bool hold = group > 0 || npend > 0 || (recogniser_busy() && (!cur.nsub || tickets_running() >= HOLD_MIN));
            if (hold) {
                pend[npend++] = cur;  // nothing else is queued for this call now (the caller's stream joins with the last partner's call)
                if (npend == gcap || (group < 0 && npend >= 2 && !recogniser_busy())) flush_pending();
                return;
            }
        }
        later_stages(&cur, 1, pipe3);
    }

    // the waiting calls' crop + recogniser + match: the group is complete, or the missing partners never came
    void flush_pending() {
        if (!npend) return;
        CallRec grp[MAXG];
        const int n = npend;
        for (int i = 0; i < n; ++i) grp[i] = pend[i];
        npend = 0;
        try {
            later_stages(grp, n, true);  // (a call is only ever deferred in the three-stream mode: its detector stage sits on det_stream)
        } catch (...) {
            // the held calls are lost; their tickets must not be answered from a staging set's STALE "results have left" event: mark them
            // failed and re-arm the event behind whatever did get queued, so that frt_pipeline_wait returns - with the error
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < grp[i].nsub; ++j) {
                    grp[i].sub[j].ab->failed = true;
                    (void)hipEventRecord(grp[i].sub[j].ab->ev_out, stream);
                }
            throw;
        }
    }
    bool is_pending(long ticket) const {
        for (int i = 0; i < npend; ++i)
            for (int j = 0; j < pend[i].nsub; ++j)
                if (pend[i].sub[j].ticket == ticket) return true;
        return false;
    }

    // E and M of one call, or of up to MAXG consecutive calls as ONE recogniser pass and ONE match call (pairing)
    void later_stages(const CallRec *c, int nc, bool pipe3) {
        hipStream_t s = stream;
        const DetGeom &g = det->g;
        int Fc[MAXG] = {}, Ftot = 0;
        for (int i = 0; i < nc; ++i) {
            Fc[i] = c[i].n * max_faces;
            Ftot += Fc[i];
        }
        (nc >= 2 ? paired_passes : single_passes) += 1;
        const int eset = (pipe3 && dual_embed && Ftot <= emb->max_batch) ? (int)(epass++ & 1u) : 0;  // activation set / stream of this recogniser pass
        hipStream_t es = pipe3 ? (eset ? emb_stream2 : emb_stream) : s;
        // match + pack follow the recogniser pass on ITS stream (they overlap the other set's pass and the next detector pass): a stream
        // of their own measured 0.6 % slower and is one more stream competing for the four hardware queues
        hipStream_t ms = es;
        float *chw = eset ? d_chw2 : d_chw;
        // embeddings and validity flags of the pass: the first call's slot (the calls of a group fit one slot together: run() checked)
        float *emb_slot = slot_embeds[c[0].slot];
        int *valid = slot_valid[c[0].slot];
        for (int i = 0; i < nc; ++i) {
            if (pipe3 && c[i].call >= (unsigned)NSLOT) HIPCHK(hipStreamWaitEvent(es, ev_done[c[i].slot], 0));
            if (pipe3) HIPCHK(hipStreamWaitEvent(es, ev_det[c[i].slot], 0));
        }
        const bool have_gallery = mat && mat->N > 0;
        const unsigned gen = mat ? mat->generation : 0u;
        const int akey = (align ? 1 : 0) | (c[0].crops ? 2 : 0);
        // (a pass on activation set k follows the previous pass on the same set: ordered by its stream)
        auto stage_e = [&](hipStream_t st) {
            int f_off = 0;
            for (int i = 0; i < nc; ++i) {
                const int sl = c[i].slot;
                if (align) {
                    ProfScope ps(2, "align_faces", (double)Fc[i] * 112 * 112 * 3, st);
                    launch_align_faces(c[i].frames, g.frame_h, g.frame_w, (size_t)g.frame_w * 3, (size_t)g.frame_w * g.frame_h * 3, slot_landmarks[sl],
                                       slot_nout[sl], max_faces, Fc[i], 0, c[i].crops, chw + (size_t)f_off * 3 * 112 * 112, valid + f_off, st);
                } else {
                    ProfScope ps(2, "crop_faces", (double)Fc[i] * 112 * 112 * 3, st);
                    launch_crop_faces(c[i].frames, g.frame_h, g.frame_w, (size_t)g.frame_w * 3, (size_t)g.frame_w * g.frame_h * 3, slot_boxes[sl],
                                      slot_nout[sl], max_faces, Fc[i], 0, 112, 112, c[i].crops, chw + (size_t)f_off * 3 * 112 * 112, valid + f_off, st);
                }
                f_off += Fc[i];
            }
#ifdef FRT_TUNING
            static const int pipe_abl = getenv("FRT_PIPE_ABLATE") ? atoi(getenv("FRT_PIPE_ABLATE")) : 0;
            if (!(pipe_abl & 2) || c[0].call < 8u)
#endif
            for (int f0 = 0; f0 < Ftot; f0 += emb->max_batch) {
                const int nf = std::min(emb->max_batch, Ftot - f0);
                emb->forward_set(eset, chw + (size_t)f0 * 3 * 112 * 112, nf, valid + f0, emb_slot + (size_t)f0 * 512, st);
            }
        };
        // (the fp32 pass is never captured: its single activation set is handed from pass to pass through the host-tracked event f32.done,
        //  which must be a real record on every pass - and a graph captured in one precision must not be replayed in the other)
        if (nc == 1 && !emb->fp32_mode) run_part(GraphKey{1, c[0].frames, nullptr, nullptr, c[0].n, c[0].slot, akey, (unsigned)eset}, es, stage_e);
        else stage_e(es);
        HIPCHK(hipEventRecord(emb->ev_busy[eset], es));
        emb->busy[eset] = true;
        if (pipe3) {
            HIPCHK(hipEventRecord(ev_emb[c[0].slot], es));
            HIPCHK(hipStreamWaitEvent(ms, ev_emb[c[0].slot], 0));
        }
        // consecutive calls' match stages sit on DIFFERENT streams (their recogniser passes') but share the matcher's scratch and this
        // pipeline's d_idx / d_sim: each one starts behind the previous one's end (they rarely meet: 0.3 ms every 3.3 ms, half a
        // period apart - which is exactly why an unordered pair showed up as one failing equality test in several hundred)
        // (the serial branch too: an object-level frt_matcher_top1_dev / topk_dev on another stream shares d_partial / the pair lists with this stage)
        if (mat && mat->busy) HIPCHK(hipStreamWaitEvent(ms, mat->ev_busy, 0));
        auto stage_m = [&](hipStream_t st) {
#ifdef FRT_TUNING
            static const int pipe_abl = getenv("FRT_PIPE_ABLATE") ? atoi(getenv("FRT_PIPE_ABLATE")) : 0;
            if (!(pipe_abl & 4) || c[0].call < 8u)
#endif
            if (have_gallery) mat->top1_dev(emb_slot, Ftot, d_idx, d_sim, st);
            int f_off = 0;
            for (int i = 0; i < nc; ++i) {
                const int sl = c[i].slot;
                {
                    ProfScope ps(2, "pack_results", (double)Fc[i], st);
                    // one launch per ticket of the call (merged submits): a ticket's records count ITS frames from zero
                    const int np = c[i].nsub > 1 ? c[i].nsub : 1;
                    for (int j = 0, fr = 0; j < np; ++j) {
                        const int nf = c[i].nsub > 1 ? c[i].sub[j].n : c[i].n, o = fr * max_faces;
                        launch_pack_results(slot_boxes[sl] + o, slot_nout[sl] + fr, valid + f_off + o, have_gallery ? d_idx + f_off + o : nullptr,
                                            have_gallery ? d_sim + f_off + o : nullptr, max_faces, nf * max_faces, c[i].results + o, st);
                        fr += nf;
                    }
                }
                if (c[i].embeds)
                    HIPCHK(hipMemcpyAsync(c[i].embeds, emb_slot + (size_t)f_off * 512, sizeof(float) * 512 * Fc[i], hipMemcpyDeviceToDevice, st));
                f_off += Fc[i];
            }
        };
        if (nc == 1) run_part(GraphKey{2, nullptr, c[0].results, c[0].embeds, c[0].n, c[0].slot, akey, gen}, ms, stage_m);
        else stage_m(ms);
        if (mat) {
            HIPCHK(hipEventRecord(mat->ev_busy, ms));
            mat->busy = true;
        }
        if (pipe3) {
            for (int i = 0; i < nc; ++i) HIPCHK(hipEventRecord(ev_done[c[i].slot], ms));
            HIPCHK(hipStreamWaitEvent(s, ev_done[c[0].slot], 0));  // the caller's stream joins here
        } else if (overlap) {
            HIPCHK(hipEventRecord(ev_serial, s));
            serial_pending = true;
        }
        // calls that came through frt_pipeline_submit: their downloads follow the join
        for (int i = 0; i < nc; ++i) {
            size_t o = 0;  // face slots in front of this ticket inside the call's device blocks (c[i].results / .embeds / .crops)
            for (int j = 0; j < c[i].nsub; ++j) {
                const Sub &t = c[i].sub[j];
                const size_t nf = (size_t)t.n * max_faces;
                HIPCHK(hipMemcpyAsync(t.h_results, c[i].results + o, sizeof(frt_face_result) * nf, hipMemcpyDeviceToHost, s));
                if (t.h_embeds) HIPCHK(hipMemcpyAsync(t.h_embeds, c[i].embeds + o * 512, sizeof(float) * 512 * nf, hipMemcpyDeviceToHost, s));
                if (t.h_crops) HIPCHK(hipMemcpyAsync(t.h_crops, c[i].crops + o * 112 * 112 * 3, nf * 112 * 112 * 3, hipMemcpyDeviceToHost, s));
                o += nf;
            }
            // "results have left" only behind the LAST download of the call: the first ticket's staging set carries every ticket's data
            for (int j = 0; j < c[i].nsub; ++j) HIPCHK(hipEventRecord(c[i].sub[j].ab->ev_out, s));
        }
    }
};

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

const char *frt_last_error(void) { return frthost::last_error().c_str(); }
const char *frt_version(void) { return "libfrt 0.1 (gfx950)"; }
long frt_set_wait_spin_us(long microseconds) {
    const long prev = wait_spin_us();
    g_wait_spin_us.store(std::max(0L, microseconds), std::memory_order_relaxed);
    return prev;
}
int frt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int frt_probe_sustained_mfma(int device, int mix, double seconds, double *tflops_out) {
    return guarded([&] {
        if (!tflops_out || mix < 0 || mix > 2 || !(seconds > 0.0) || seconds > 10.0) raise(FRT_ERR_INVALID, "frt_probe_sustained_mfma: bad argument");
        use_device(device);
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        const int n_wg = prop.multiProcessorCount;
        std::vector<uint16_t> h(8192 * 8);
        uint32_t x = 0x2545F491u;
        for (auto &v : h) {  // random fp16 values in (-1, 1): sign, exponent 8..14, random mantissa (real operand toggling, no inf / nan)
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            v = (uint16_t)(((x >> 3) & 0x8000u) | ((8u + (x >> 20) % 7u) << 10) | (x & 0x3ffu));
        }
        void *src = nullptr;
        float *out = nullptr;
        hipStream_t st = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        HIPCHK(hipMalloc(&src, h.size() * 2));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&out), (size_t)n_wg * 256 * sizeof(float)));
        HIPCHK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        const int iters = 2000;                 // ~ 1.2 - 1.6 ms per launch: long enough for the clock to settle at its power-managed value
        (void)launch_mfma_probe(mix, n_wg, iters, src, out, st);   // code load + first touch off the clock
        HIPCHK(hipStreamSynchronize(st));
        double flop = 0.0, ms_total = 0.0;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
        do {
            HIPCHK(hipEventRecord(e0, st));
            double f = 0.0;
            for (int i = 0; i < 8; ++i) f += launch_mfma_probe(mix, n_wg, iters, src, out, st);
            HIPCHK(hipEventRecord(e1, st));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            flop += f;
            ms_total += ms;
        } while (std::chrono::steady_clock::now() < t_end);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(st);
        (void)hipFree(src);
        (void)hipFree(out);
        *tflops_out = flop / (ms_total * 1e-3) / 1e12;
    });
}

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

// ------------------------------------------------------------------------------------------------------------- matcher
int frt_matcher_create(int device, frt_matcher **out) {
    return guarded([&] {
        if (!out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        use_device(device);
        std::unique_ptr<frt_matcher> m(new frt_matcher);
        m->device = device;
        HIPCHK(hipStreamCreate(&m->stream));
        HIPCHK(hipEventCreateWithFlags(&m->ev_busy, hipEventDisableTiming));
        *out = m.release();
    });
}

void frt_matcher_destroy(frt_matcher *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    // an unfinished streaming load runs on m->stream (ld.s == stream): abort it while the stream still exists
    m->load_abort();
    m->load_release_staging();
    m->ld.s = nullptr;
    if (m->stream) {
        (void)hipStreamSynchronize(m->stream);
        (void)hipStreamDestroy(m->stream);
    }
    if (m->ev_busy) (void)hipEventDestroy(m->ev_busy);
    for (void *p : {(void *)m->d_gallery, (void *)m->d_q, (void *)m->d_sim, (void *)m->d_idx, (void *)m->d_partial, (void *)m->d_full, (void *)m->d_g16, (void *)m->d_kth,
                    (void *)m->d_g8, (void *)m->d_g8_scale})
        if (p) (void)hipFree(p);
    m->free_screen_scratch();
    delete m;
}

static void check_gallery_shape(int num_row, int num_col) {
    if (num_row < 0) raise(FRT_ERR_INVALID, "MatMul::init: bad argument");
    if (num_col < 32 || num_col % 32) raise(FRT_ERR_INVALID, "MatMul::init: numCol must be a multiple of 32");
}

int frt_matcher_set_storage(frt_matcher *m, int fp16) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        m->want16 = fp16 != 0;
    });
}

int frt_matcher_set_screening(frt_matcher *m, int on) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        m->screen_on = on != 0;
    });
}

unsigned frt_matcher_generation(frt_matcher *m) {
    if (!m) return 0;
    std::lock_guard<std::mutex> lk(m->mu);
    return m->generation;
}

size_t frt_matcher_scan_bytes(frt_matcher *m) {
    if (!m) return 0;
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t n = (size_t)m->N, d = (size_t)m->D;
    if (m->screen && m->screen_on) return (m->d_g8 ? 1 : 2) * n * d;   // the coarse scan reads the shadow copy once per call
    return (m->store16 ? 2 : 4) * n * d;
}

int frt_matcher_init(frt_matcher *m, const float *gallery, int num_row, int num_col) {
    return guarded([&] {
        if (!m || (num_row > 0 && !gallery)) raise(FRT_ERR_INVALID, "MatMul::init: bad argument");
        check_gallery_shape(num_row, num_col);
        std::lock_guard<std::mutex> lk(m->mu);
        use_device(m->device);
        // one path for every gallery load: pinned staging chunks + asynchronous copies (idempotent: the previous device copy is
        // freed at commit - the reference leaks it on every /reload)
        m->load_begin(num_row, num_col);
        try {
            m->load_append(gallery, num_row);
            m->load_commit();
        } catch (...) {
            m->load_abort();
            throw;
        }
    });
}

int frt_matcher_gallery_begin(frt_matcher *m, int row_capacity, int num_col) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        check_gallery_shape(row_capacity, num_col);
        std::lock_guard<std::mutex> lk(m->mu);
        use_device(m->device);
        m->load_begin(row_capacity, num_col);
    });
}

int frt_matcher_gallery_append(frt_matcher *m, const void *rows, int n_rows) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        use_device(m->device);
        m->load_append(reinterpret_cast<const float *>(rows), n_rows);
    });
}

int frt_matcher_gallery_commit(frt_matcher *m) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        use_device(m->device);
        try {
            m->load_commit();
        } catch (...) {
            m->load_abort();
            throw;
        }
    });
}

int frt_matcher_num_rows(const frt_matcher *m) { return m ? m->N : 0; }

int frt_matcher_set_row_offset(frt_matcher *m, int row_offset) {
    return guarded([&] {
        if (!m || row_offset < 0) raise(FRT_ERR_INVALID, "set_row_offset: bad argument");
        std::lock_guard<std::mutex> lk(m->mu);
        m->row_offset = row_offset;
        ++m->generation;
    });
}

int frt_matcher_calculate(frt_matcher *m, const float *embeds, int embed_count, float *outputs) {
    return guarded([&] {
        if (!m || !embeds || !outputs) raise(FRT_ERR_INVALID, "MatMul::calculate: null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = m->stream;
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        const size_t need = (size_t)embed_count * m->N;
        if (need > m->full_cap) {
            if (m->d_full) (void)hipFree(m->d_full);
            m->d_full = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_full), need * sizeof(float)));
            m->full_cap = need;
        }
        HIPCHK(hipMemcpyAsync(m->d_q, embeds, sizeof(float) * (size_t)embed_count * m->D, hipMemcpyHostToDevice, s));
        for (int f0 = 0; f0 < embed_count; f0 += 128) {
            const int nf = std::min(128, embed_count - f0);
            if (m->store16)
                launch_match_full_h(m->d_g16, m->N, m->D, m->d_q + (size_t)f0 * m->D, nf, m->d_full + (size_t)f0 * m->N, s);
            else
                launch_match_full(m->d_gallery, m->N, m->D, m->d_q + (size_t)f0 * m->D, nf, m->d_full + (size_t)f0 * m->N, s);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(outputs, m->d_full, need * sizeof(float), hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_matcher_calculate_top1(frt_matcher *m, const float *embeds, int embed_count, float *outputs, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (!m || !embeds || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "calculate_top1: null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = m->stream;
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        HIPCHK(hipMemcpyAsync(m->d_q, embeds, sizeof(float) * (size_t)embed_count * m->D, hipMemcpyHostToDevice, s));
        if (outputs) {
            const size_t need = (size_t)embed_count * m->N;
            if (need > m->full_cap) {
                if (m->d_full) (void)hipFree(m->d_full);
                m->d_full = nullptr;
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_full), need * sizeof(float)));
                m->full_cap = need;
            }
            for (int f0 = 0; f0 < embed_count; f0 += 128) {
                const int nf = std::min(128, embed_count - f0);
                if (m->store16)
                    launch_match_full_h(m->d_g16, m->N, m->D, m->d_q + (size_t)f0 * m->D, nf, m->d_full + (size_t)f0 * m->N, s);
                else
                    launch_match_full(m->d_gallery, m->N, m->D, m->d_q + (size_t)f0 * m->D, nf, m->d_full + (size_t)f0 * m->N, s);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(outputs, m->d_full, need * sizeof(float), hipMemcpyDeviceToHost, s));  // (the top-1 search below runs under this copy's tail)
        }
        m->top1_dev(m->d_q, embed_count, m->d_idx, m->d_sim, s);
        HIPCHK(hipMemcpyAsync(idx_out, m->d_idx, sizeof(int32_t) * embed_count, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sim_out, m->d_sim, sizeof(float) * embed_count, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_pinned_alloc(size_t bytes, int device, void **out) {
    return guarded([&] {
        if (!out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (device >= 0) use_device(device);
        HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    });
}

void frt_pinned_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int frt_matcher_top1(frt_matcher *m, const float *embeds, int embed_count, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (!m || !embeds || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "top1: null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = m->stream;
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        HIPCHK(hipMemcpyAsync(m->d_q, embeds, sizeof(float) * (size_t)embed_count * m->D, hipMemcpyHostToDevice, s));
        m->top1_dev(m->d_q, embed_count, m->d_idx, m->d_sim, s);
        HIPCHK(hipMemcpyAsync(idx_out, m->d_idx, sizeof(int32_t) * embed_count, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sim_out, m->d_sim, sizeof(float) * embed_count, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

/* device-resident queries (sharded-gallery path, dist.py: the all-gathered embeddings never visit the host) */
int frt_matcher_top1_dev(frt_matcher *m, const void *embeds_dev, int embed_count, void *idx_dev, void *sim_dev, void *hip_stream) {
    return guarded([&] {
        if (!m || !embeds_dev || !idx_dev || !sim_dev) raise(FRT_ERR_INVALID, "top1_dev: null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        m->top1_dev(reinterpret_cast<const float *>(embeds_dev), embed_count, reinterpret_cast<int32_t *>(idx_dev), reinterpret_cast<float *>(sim_dev), s);
        HIPCHK(hipEventRecord(m->ev_busy, s));  // the scratch stays in use until this call has run
        m->busy = true;
    });
}

int frt_merge_top1(int n, const int32_t *idx_a, const float *sim_a, const int32_t *idx_b, const float *sim_b, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (n < 0 || !idx_a || !sim_a || !idx_b || !sim_b || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "merge: bad argument");
        for (int i = 0; i < n; ++i) {
            const bool a_ok = idx_a[i] >= 0, b_ok = idx_b[i] >= 0;
            bool take_b = false;
            if (!a_ok)
                take_b = b_ok;
            else if (b_ok)
                take_b = (sim_b[i] > sim_a[i]) || (sim_b[i] == sim_a[i] && idx_b[i] < idx_a[i]);
            idx_out[i] = take_b ? idx_b[i] : idx_a[i];
            sim_out[i] = take_b ? sim_b[i] : sim_a[i];
        }
    });
}

static void check_k(int k) {
    if (k < 1 || k > match_topk_max() || k > frt_matcher::KCAP) raise(FRT_ERR_INVALID, "top-k: k must be in 1..16");
}

int frt_matcher_topk(frt_matcher *m, const float *embeds, int embed_count, int k, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (!m || !embeds || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "topk: null argument");
        check_k(k);
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = m->stream;
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        HIPCHK(hipMemcpyAsync(m->d_q, embeds, sizeof(float) * (size_t)embed_count * m->D, hipMemcpyHostToDevice, s));
        m->topk_dev(m->d_q, embed_count, k, m->d_idx, m->d_sim, s);
        HIPCHK(hipMemcpyAsync(idx_out, m->d_idx, sizeof(int32_t) * (size_t)embed_count * k, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sim_out, m->d_sim, sizeof(float) * (size_t)embed_count * k, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_matcher_topk_dev(frt_matcher *m, const void *embeds_dev, int embeds_fp16, int embed_count, int k, void *idx_dev, void *sim_dev, void *hip_stream) {
    return guarded([&] {
        if (!m || !embeds_dev || !idx_dev || !sim_dev) raise(FRT_ERR_INVALID, "topk_dev: null argument");
        check_k(k);
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        const float *q = reinterpret_cast<const float *>(embeds_dev);
        if (embeds_fp16) {  // exact widening into the query scratch
            launch_half_to_float(reinterpret_cast<const half_t *>(embeds_dev), (long)embed_count * m->D, m->d_q, s);
            q = m->d_q;
        }
        m->topk_dev(q, embed_count, k, reinterpret_cast<int32_t *>(idx_dev), reinterpret_cast<float *>(sim_dev), s);
        HIPCHK(hipEventRecord(m->ev_busy, s));  // the scratch stays in use until this call has run
        m->busy = true;
    });
}

int frt_merge_topk(int shards, int n, int k, const int32_t *idx_all, const float *sim_all, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (shards < 1 || n < 0 || k < 1 || !idx_all || !sim_all || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "merge_topk: bad argument");
        std::vector<int> pos((size_t)shards);
        for (int q = 0; q < n; ++q) {
            std::fill(pos.begin(), pos.end(), 0);
            for (int o = 0; o < k; ++o) {
                int best = -1, bi = 0;
                float bv = 0.f;
                for (int sh = 0; sh < shards; ++sh) {
                    while (pos[(size_t)sh] < k && idx_all[((size_t)sh * n + q) * k + pos[(size_t)sh]] < 0) ++pos[(size_t)sh];  // empty slots
                    if (pos[(size_t)sh] >= k) continue;
                    const size_t e = ((size_t)sh * n + q) * k + pos[(size_t)sh];
                    const float v = sim_all[e];
                    const int i = idx_all[e];
                    if (best < 0 || v > bv || (v == bv && i < bi)) {
                        best = sh;
                        bv = v;
                        bi = i;
                    }
                }
                if (best < 0) {
                    idx_out[(size_t)q * k + o] = -1;
                    sim_out[(size_t)q * k + o] = -INFINITY;
                } else {
                    idx_out[(size_t)q * k + o] = bi;
                    sim_out[(size_t)q * k + o] = bv;
                    ++pos[(size_t)best];
                }
            }
        }
    });
}

int frt_merge_topk_dev(int shards, int n, int k, const void *idx_all_dev, const void *sim_all_dev, void *idx_out_dev, void *sim_out_dev, void *hip_stream) {
    return guarded([&] {
        if (shards < 1 || n < 0 || k < 1 || !idx_all_dev || !sim_all_dev || !idx_out_dev || !sim_out_dev) raise(FRT_ERR_INVALID, "merge_topk_dev: bad argument");
        if (n == 0) return;
        launch_merge_topk(reinterpret_cast<const int32_t *>(idx_all_dev), reinterpret_cast<const float *>(sim_all_dev), shards, n, k,
                          reinterpret_cast<int32_t *>(idx_out_dev), reinterpret_cast<float *>(sim_out_dev), reinterpret_cast<hipStream_t>(hip_stream));
        HIPCHK(hipGetLastError());
    });
}

int frt_embeds_to_half_dev(const void *embeds_dev, size_t n_values, void *half_out_dev, void *hip_stream) {
    return guarded([&] {
        if (!embeds_dev || !half_out_dev || n_values % 8) raise(FRT_ERR_INVALID, "embeds_to_half: bad argument (n_values must be a multiple of 8)");
        if (n_values == 0) return;
        launch_float_to_half(reinterpret_cast<const float *>(embeds_dev), (long)n_values, reinterpret_cast<half_t *>(half_out_dev),
                             reinterpret_cast<hipStream_t>(hip_stream));
        HIPCHK(hipGetLastError());
    });
}

// ------------------------------------------------------------------------------------------------------------ pipeline
int frt_pipeline_create(frt_detector *d, frt_embedder *e, frt_matcher *m, int max_frames, frt_pipeline **out) {
    return guarded([&] {
        if (!d || !e || !out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (max_frames < 1 || max_frames > d->max_batch) raise(FRT_ERR_CAPACITY, "pipeline: max_frames exceeds det_maxBatchSize");
        if (d->device != e->device || (m && m->device != d->device)) raise(FRT_ERR_INVALID, "pipeline: objects live on different devices");
        use_device(d->device);
        std::unique_ptr<frt_pipeline> p(new frt_pipeline);
        p->det = d; p->emb = e; p->mat = m;
        p->max_frames = max_frames;
        p->max_faces = d->g.max_faces;
        p->F_cap = max_frames * p->max_faces;
        // the pipeline's own join stream is created on first use: ROCm maps streams onto 4 hardware queues round-robin and streams that
        // share a queue serialise, so a stream nobody uses (callers usually pass theirs) should not take a slot among the stage streams
        p->stream = nullptr;
        // the stage streams are created at the highest stream priority: ROCm keeps a separate hardware-queue pool per priority, so they
        // never share a queue with the caller's (normal priority) stream, whose queue holds the pending joins of the batches in flight
        int prio_lo = 0, prio_hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        {
            const char *pe = getenv("FRT_PIPELINE_STREAM_PRIO");
            if (pe && pe[0] == '0') prio_hi = 0;   // "0": normal priority (stage streams share the caller's queue pool)
        }
        // hipStreamDefault (blocking), not hipStreamNonBlocking: a gallery reload between calls (hipFree / hipMalloc / synchronous
        // hipMemcpy on the legacy default stream) is then ordered against the stages still in flight without the caller
        // synchronising anything (tests/test_gpu_pipeline.py::test_gallery_reload_between_pipelined_calls); non-blocking
        // streams also measured 1 % slower
        p->copy_prio = prio_hi;
        auto mk = [&](hipStream_t *st) { HIPCHK(hipStreamCreateWithPriority(st, hipStreamDefault, prio_hi)); };
        {
            // FRT_PIPELINE_DET_PRIO=lo / normal: the detector's stream below the recogniser's (A/B: does the hardware then give the recogniser -
            // the longer stage - the CUs first and let the detector fill its gaps?)
            const char *dp = getenv("FRT_PIPELINE_DET_PRIO");
            if (dp && dp[0] == 'l') HIPCHK(hipStreamCreateWithPriority(&p->det_stream, hipStreamDefault, prio_lo));
            else if (dp && dp[0] == 'n') HIPCHK(hipStreamCreateWithPriority(&p->det_stream, hipStreamDefault, 0));
            else mk(&p->det_stream);
        }
        mk(&p->emb_stream);
        mk(&p->emb_stream2);
        {
            const char *de = getenv("FRT_PIPELINE_DUAL_EMBED");
            p->dual_embed = !(de && de[0] == '0');
        }
        if (p->dual_embed) {
            e->ensure_alt();
            p->d_chw2 = p->arena.alloc<float>((size_t)max_frames * d->g.max_faces * 3 * 112 * 112);
        }
        const size_t F = (size_t)p->F_cap;
        HIPCHK(hipEventCreateWithFlags(&p->ev_serial, hipEventDisableTiming));
        for (int i = 0; i < frt_pipeline::NSLOT; ++i) {
            HIPCHK(hipEventCreateWithFlags(&p->ev_det[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p->ev_emb[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p->ev_done[i], hipEventDisableTiming));
            p->slot_embeds[i] = p->arena.alloc<float>(F * 512);
            p->slot_valid[i] = p->arena.alloc<int>(F);
            p->slot_boxes[i] = p->arena.alloc<frt_bbox>(F);
            p->slot_nout[i] = p->arena.alloc<int>((size_t)max_frames);
            if (d->has_landmarks) p->slot_landmarks[i] = p->arena.alloc<float>(F * 10);
        }
        {
            const char *e = getenv("FRT_PIPELINE_OVERLAP");
            p->overlap = !(e && e[0] == '0');
            const char *gph = getenv("FRT_PIPELINE_GRAPH");
            p->use_graphs = gph && gph[0] == '1';  // opt-in: measured no gain on this workload (see the note at run_part)
        }
        HIPCHK(hipEventCreateWithFlags(&p->ev_input, hipEventDisableTiming));
        p->d_chw = p->arena.alloc<float>(F * 3 * 112 * 112);
        p->d_sim = p->arena.alloc<float>(F);
        p->d_idx = p->arena.alloc<int32_t>(F);
        if (p->overlap) {  // create-time self-check of the stage streams (~1 ms); FRT_PIPELINE_SELFCHECK=0 skips it
            const char *sc = getenv("FRT_PIPELINE_SELFCHECK");
            if (!(sc && sc[0] == '0')) p->self_check(false);
        }
        *out = p.release();
    });
}

static void pipeline_flush_locked(frt_pipeline *p);
static void pipeline_start_held(frt_pipeline *p);

void frt_pipeline_destroy(frt_pipeline *p) {
    if (!p) return;
    (void)hipSetDevice(p->det->device);
    if (p->npend || p->held.on) {  // pairing / merging: calls still waiting for partners run now - a submitted batch is never dropped
        try {
            std::lock_guard<std::mutex> lk(p->run_mu);
            pipeline_flush_locked(p);
        } catch (...) {
        }
    }
    if (p->det_stream) (void)hipStreamSynchronize(p->det_stream);
    if (p->emb_stream) (void)hipStreamSynchronize(p->emb_stream);
    if (p->emb_stream2) (void)hipStreamSynchronize(p->emb_stream2);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    p->drop_graphs();
    if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    if (p->det_stream) (void)hipStreamDestroy(p->det_stream);
    if (p->emb_stream) (void)hipStreamDestroy(p->emb_stream);
    if (p->emb_stream2) (void)hipStreamDestroy(p->emb_stream2);
    if (p->ev_serial) (void)hipEventDestroy(p->ev_serial);
    if (p->ev_input) (void)hipEventDestroy(p->ev_input);
    if (p->copy_stream) {
        (void)hipStreamSynchronize(p->copy_stream);
        (void)hipStreamDestroy(p->copy_stream);
    }
    for (frt_pipeline::AsyncBuf &b : p->abuf) {
        if (b.ev_h2d) (void)hipEventDestroy(b.ev_h2d);
        if (b.ev_out) (void)hipEventDestroy(b.ev_out);
    }
    for (int i = 0; i < frt_pipeline::NSLOT; ++i) {
        if (p->ev_det[i]) (void)hipEventDestroy(p->ev_det[i]);
        if (p->ev_emb[i]) (void)hipEventDestroy(p->ev_emb[i]);
        if (p->ev_done[i]) (void)hipEventDestroy(p->ev_done[i]);
    }
    p->arena.release();
    delete p;
}

// Caller holds p->run_mu.
static void pipeline_lock_run(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev) {
    if (n_frames < 1 || n_frames > p->max_frames) raise(FRT_ERR_CAPACITY, "pipeline: more frames than max_frames");
    std::lock_guard<std::mutex> l1(p->det->mu);
    std::lock_guard<std::mutex> l2(p->emb->mu);
    std::unique_lock<std::mutex> l3;
    if (p->mat) {
        l3 = std::unique_lock<std::mutex>(p->mat->mu);
        if (p->mat->N > 0) p->mat->ensure_queries(p->F_cap);
    }
    p->run(reinterpret_cast<const uint8_t *>(frames_dev), n_frames, reinterpret_cast<frt_face_result *>(results_dev),
           reinterpret_cast<float *>(embeds_dev));
}

int frt_pipeline_run_dev(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev) {
    return guarded([&] {
        if (!p || !frames_dev || !results_dev) raise(FRT_ERR_INVALID, "null argument");
        use_device(p->det->device);
        std::lock_guard<std::mutex> lk(p->run_mu);
        pipeline_start_held(p);  // (submits held back at the host boundary go first)
        pipeline_lock_run(p, frames_dev, n_frames, results_dev, embeds_dev);
    });
}

int frt_pipeline_run_dev_after(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev, void *ready_event) {
    return guarded([&] {
        if (!p || !frames_dev || !results_dev) raise(FRT_ERR_INVALID, "null argument");
        use_device(p->det->device);
        std::lock_guard<std::mutex> lk(p->run_mu);
        pipeline_start_held(p);
        p->ev_ready = reinterpret_cast<hipEvent_t>(ready_event);
        try {
            pipeline_lock_run(p, frames_dev, n_frames, results_dev, embeds_dev);
        } catch (...) {
            p->ev_ready = nullptr;
            throw;
        }
    });
}

int frt_pipeline_check_overlap(frt_pipeline *p, float *ratio_out) {
    int rc = guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        use_device(p->det->device);
        std::lock_guard<std::mutex> la(p->async_mu);  // same order as pipeline_submit_impl: async_mu, then run_mu
        std::lock_guard<std::mutex> lk(p->run_mu);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        if (p->stream) HIPCHK(hipStreamSynchronize(p->stream));
        p->ensure_stream();
        p->ensure_async();  // the upload stream of frt_pipeline_submit / run takes part
        p->self_check(true);
        if (ratio_out) *ratio_out = p->overlap_ratio;
    });
    if (rc == FRT_OK && p && !p->warning.empty()) frthost::last_error() = p->warning;  // FRT_OK + a message: a warning, not a failure
    return rc;
}

int frt_pipeline_set_input_sync(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        p->input_sync = enable != 0;
    });
}

int frt_pipeline_sync(frt_pipeline *p) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        use_device(p->det->device);
        {
            std::lock_guard<std::mutex> lk(p->run_mu);
            pipeline_flush_locked(p);
        }
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->emb->check_se_error();
    });
}

int frt_pipeline_set_stream(frt_pipeline *p, void *hip_stream) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : p->own_stream;  // null own_stream: created at the next run
    });
}

int frt_pipeline_set_overlap(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->overlap = enable != 0;
        p->seq = 0;
        p->drop_graphs();
    });
}

int frt_pipeline_set_graph(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->use_graphs = enable != 0;
        p->drop_graphs();
    });
}

int frt_pipeline_set_align(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        if (enable && !p->det->has_landmarks) raise(FRT_ERR_FORMAT, "pipeline: alignment needs a detector blob with the LandmarkHead");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->align = enable != 0;
    });
}

int frt_pipeline_set_pairing(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        p->group = enable < 0 ? -1 : (enable == 0 ? 0 : std::min(std::max(enable, 2), (int)frt_pipeline::MAXG));
        p->adaptive_dev = enable == -2;
        p->merge_submits = enable != -3;
    });
}

int frt_pipeline_merge_stats(frt_pipeline *p, long *merged_calls, long *merged_tickets) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        if (merged_calls) *merged_calls = p->merged_calls;
        if (merged_tickets) *merged_tickets = p->merged_tickets;
    });
}

int frt_pipeline_graph_stats(frt_pipeline *p, long *captured, long *replayed) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        if (captured) *captured = p->graphs_captured;
        if (replayed) *replayed = p->graphs_replayed;
    });
}

int frt_pipeline_pairing_stats(frt_pipeline *p, long *paired_passes, long *single_passes) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        if (paired_passes) *paired_passes = p->paired_passes;
        if (single_passes) *single_passes = p->single_passes;
    });
}

static void pipeline_lock_run(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev);

// Caller holds p->run_mu: the held submits (merged at the host boundary) go out as ONE call.  Never throws: a failure is left on the tickets
// (AsyncBuf::failed, reported by frt_pipeline_wait) - the caller of the moment may be somebody else's submit or wait.
static void pipeline_start_held(frt_pipeline *p) {
    if (!p->held.on) return;
    frt_pipeline::Held h = p->held;
    p->held = frt_pipeline::Held{};
    p->ev_frames = h.base->ev_h2d;  // recorded behind the last ticket's upload
    p->crops_req = h.want_crops ? h.base->d_crops : nullptr;
    p->serial_call = false;
    p->host_req = frt_pipeline::CallRec{};
    p->host_req.nsub = h.nsub;
    for (int j = 0; j < h.nsub; ++j) p->host_req.sub[j] = h.sub[j];
    try {
        pipeline_lock_run(p, h.base->d_frames, h.n, h.base->d_results, h.want_embeds ? h.base->d_embeds : nullptr);
        p->merged_calls += h.nsub > 1;
        p->merged_tickets += h.nsub > 1 ? h.nsub : 0;
    } catch (const std::exception &e) {
        p->held_error = e.what();
        p->ev_frames = nullptr;
        p->crops_req = nullptr;
        p->host_req = frt_pipeline::CallRec{};
        for (int j = 0; j < h.nsub; ++j) {
            h.sub[j].ab->failed = true;
            (void)hipEventRecord(h.sub[j].ab->ev_out, p->stream);
        }
    }
}

// queue one batch through a staging set; caller holds neither mutex
static long pipeline_submit_impl(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out, bool synchronous = false,
                                 uint8_t *crops_host = nullptr) {
    if (n_frames < 1 || n_frames > p->max_frames) raise(FRT_ERR_CAPACITY, "pipeline: more frames than max_frames");
    use_device(p->det->device);
    std::lock_guard<std::mutex> lk(p->async_mu);   // staging sets + ticket order
    std::lock_guard<std::mutex> lr(p->run_mu);     // the stage enqueue itself (shared with frt_pipeline_run_dev)
    p->ensure_stream();
    p->ensure_async();
    const long ticket = p->next_ticket;
    frt_pipeline::AsyncBuf &b = p->abuf[ticket % frt_pipeline::NBUF];
    if (b.ticket >= 0) {
        // (a ticket that is still held back has no "results have left" event yet: its set's event is its previous occupant's)
        for (int j = 0; j < p->held.nsub; ++j)
            if (p->held.on && p->held.sub[j].ticket == b.ticket) pipeline_start_held(p);
        if (p->is_pending(b.ticket)) pipeline_flush_locked(p);
        wait_event_spinning(b.ev_out);  // the staging set is free once its previous batch has left
    }
    b.failed = false;
    hipStream_t s = p->stream;
    const size_t fbytes = (size_t)p->det->g.frame_h * p->det->g.frame_w * 3;
    // ---- adaptive merging at the host boundary (frt_pipeline::Held): join the held call, or become one when the detector is busy
    {
        const int K = p->max_faces;
        const bool mergeable = p->group < 0 && p->merge_submits && p->overlap && g_prof_kind == 0;
        frt_pipeline::Sub me;
        me.ab = &b;
        me.h_results = results;
        me.h_embeds = embeds_out;
        me.h_crops = crops_host;
        me.n = n_frames;
        me.ticket = ticket;
        auto join = [&](frt_pipeline::Held &h) {  // this ticket's frames behind the held ones, in the FIRST ticket's staging set
            HIPCHK(hipMemcpyAsync(h.base->d_frames + fbytes * (size_t)h.n, frames, fbytes * n_frames, hipMemcpyHostToDevice, p->copy_stream));
            HIPCHK(hipEventRecord(h.base->ev_h2d, p->copy_stream));
            h.sub[h.nsub++] = me;
            h.n += n_frames;
            h.want_embeds = h.want_embeds || embeds_out;
            h.want_crops = h.want_crops || crops_host;
            b.ticket = ticket;
            p->next_ticket = ticket + 1;
        };
        if (p->held.on) {
            const int nt = p->held.n + n_frames;
            if (mergeable && p->held.nsub < frt_pipeline::MAXSUB && nt <= p->max_frames && nt * K <= p->emb->max_batch) {
                join(p->held);
                if (p->held.nsub == frt_pipeline::MAXSUB || 2 * p->held.n > p->max_frames || !p->backed_up() || p->tickets_running() < frt_pipeline::HOLD_MIN)
                    pipeline_start_held(p);
                return ticket;
            }
            pipeline_start_held(p);  // cannot join: first in, first out
        }
        if (mergeable && 2 * n_frames <= p->max_frames && 2 * n_frames * K <= p->emb->max_batch && p->backed_up() && p->tickets_running() >= frt_pipeline::HOLD_MIN) {
            p->held.on = true;
            p->held.base = &b;
            join(p->held);
            return ticket;
        }
    }
    // A synchronous call that finds nothing else in flight (the reference's request / reply shape: one frame, one caller) has nothing to
    // overlap with: upload, detector, recogniser, match and download go down ONE stream - no stream-to-stream event hand-overs on its
    // critical path (5 of them otherwise; one 4-face call 1.02 -> 0.94 ms, profiles/r03/r03u_sync_overlap.txt).  Calls that arrive while
    // another is in flight take the stage streams as before (and are ordered behind this one through ev_serial).
    bool lone = synchronous && p->overlap && !p->npend && !p->held.on;
    for (int i = 0; lone && i < frt_pipeline::NBUF; ++i)
        if (p->abuf[i].ticket >= 0 && i != (int)(ticket % frt_pipeline::NBUF) && hipEventQuery(p->abuf[i].ev_out) != hipSuccess) lone = false;
    if (lone) {
        HIPCHK(hipMemcpyAsync(b.d_frames, frames, fbytes * n_frames, hipMemcpyHostToDevice, s));
    } else {
        HIPCHK(hipMemcpyAsync(b.d_frames, frames, fbytes * n_frames, hipMemcpyHostToDevice, p->copy_stream));
        HIPCHK(hipEventRecord(b.ev_h2d, p->copy_stream));
        p->ev_frames = b.ev_h2d;  // the stages that read the frames (detector, crop) wait for the copy; the caller's stream does not
    }
    p->serial_call = lone;
    p->crops_req = crops_host ? b.d_crops : nullptr;
    // the downloads and the "results have left" event are queued by the pipeline behind this call's match stage - now, or (pairing) with the next call
    p->host_req = frt_pipeline::CallRec{};
    p->host_req.nsub = 1;
    p->host_req.sub[0].ab = &b;
    p->host_req.sub[0].h_results = results;
    p->host_req.sub[0].h_embeds = embeds_out;
    p->host_req.sub[0].h_crops = crops_host;
    p->host_req.sub[0].n = n_frames;
    p->host_req.sub[0].ticket = ticket;
    try {
        pipeline_lock_run(p, b.d_frames, n_frames, b.d_results, embeds_out ? b.d_embeds : nullptr);
    } catch (...) {
        p->ev_frames = nullptr;
        p->crops_req = nullptr;
        p->host_req = frt_pipeline::CallRec{};
        p->serial_call = false;
        throw;
    }
    p->serial_call = false;
    b.ticket = ticket;
    p->next_ticket = ticket + 1;
    return ticket;
}

// Caller holds p->run_mu: queue the later stages of a call that is waiting for a partner (pairing).
static void pipeline_flush_locked(frt_pipeline *p) {
    pipeline_start_held(p);  // (takes the object mutexes itself)
    if (!p->npend) return;
    std::lock_guard<std::mutex> l1(p->det->mu);
    std::lock_guard<std::mutex> l2(p->emb->mu);
    std::unique_lock<std::mutex> l3;
    if (p->mat) l3 = std::unique_lock<std::mutex>(p->mat->mu);
    p->flush_pending();
}

static void pipeline_wait_impl(frt_pipeline *p, long ticket) {
    use_device(p->det->device);
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(p->async_mu);
        if (ticket < 0 || ticket >= p->next_ticket) raise(FRT_ERR_INVALID, "pipeline: unknown ticket");
        frt_pipeline::AsyncBuf &b = p->abuf[ticket % frt_pipeline::NBUF];
        if (b.ticket > ticket) return;  // its staging set was reused, which submit only does after that batch completed
        {
            std::lock_guard<std::mutex> lr(p->run_mu);
            // pairing: the partners that would share its recogniser pass have not come; adaptive pairing: calls held back behind a busy
            // recogniser go out as soon as a waiting caller finds it idle (they would be running by now had they not been held)
            if (p->held.on) {  // submits merged at the host boundary: one of its tickets is being waited for, or the detector has gone idle
                bool mine = false;
                for (int j = 0; j < p->held.nsub; ++j) mine = mine || p->held.sub[j].ticket == ticket;
                if (mine || !p->backed_up() || p->tickets_running() < frt_pipeline::HOLD_MIN) pipeline_start_held(p);
            }
            if (p->is_pending(ticket) || (p->npend && p->group < 0 && !p->recogniser_busy())) pipeline_flush_locked(p);
        }
        ev = b.ev_out;
    }
    wait_event_spinning(ev);
    {
        std::lock_guard<std::mutex> lk(p->async_mu);
        frt_pipeline::AsyncBuf &b = p->abuf[ticket % frt_pipeline::NBUF];
        if (b.ticket == ticket && b.failed)
            raise(FRT_ERR_DEVICE, "pipeline: the held stages of this call could not be queued" + (p->held_error.empty() ? std::string() : ": " + p->held_error));
    }
    p->emb->check_se_error();
}

// Synchronous host entry point.  Thread-safe: every call takes its own staging set (device frames / results / embeddings) under the
// pipeline's mutexes, so concurrent callers (the reference's Crow server is .multithreaded(), src/app.cpp:367) never share a buffer;
// with several threads calling, their batches overlap in the stage pipeline exactly like submit()/wait() batches do.
int frt_pipeline_run(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out) {
    return guarded([&] {
        if (!p || !frames || !results) raise(FRT_ERR_INVALID, "null argument");
        const long t = pipeline_submit_impl(p, frames, n_frames, results, embeds_out, true);
        pipeline_wait_impl(p, t);
    });
}

int frt_pipeline_submit(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out, long *ticket_out) {
    return guarded([&] {
        if (!p || !frames || !results || !ticket_out) raise(FRT_ERR_INVALID, "null argument");
        *ticket_out = pipeline_submit_impl(p, frames, n_frames, results, embeds_out);
    });
}

int frt_pipeline_submit_crops(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out, uint8_t *crops_out,
                              long *ticket_out) {
    return guarded([&] {
        if (!p || !frames || !results || !ticket_out) raise(FRT_ERR_INVALID, "null argument");
        *ticket_out = pipeline_submit_impl(p, frames, n_frames, results, embeds_out, false, crops_out);
    });
}

int frt_pipeline_wait(frt_pipeline *p, long ticket) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        pipeline_wait_impl(p, ticket);
    });
}

// ----------------------------------------------------------------------------------------------------------- profiling
int frt_profile_enable(int kind) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (kind < 0) {  // pause: stop recording, keep what was recorded for frt_profile_collect
        g_prof_kind = 0;
        return FRT_OK;
    }
    g_prof_kind = kind;
    for (ProfRec &r : g_prof) {
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof.clear();
    if (kind >= 0)  // (also on "off": a caller that is about to time a region switches profiling off first - the pool is filled outside it)
        while (g_prof_pool.size() < 512) {  // a profiled pipeline call brackets ~ 120 launches
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            g_prof_pool.push_back(e);
        }
    return FRT_OK;
}

int frt_profile_collect(char *names_out, size_t names_cap, double *ms_out, double *work_out, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    std::string names;
    for (ProfRec &r : g_prof) {
        if (n < cap) {
            float ms = 0.f;
            (void)hipEventSynchronize(r.b);
            if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = -1.f;
            if (ms_out) ms_out[n] = ms;
            if (work_out) work_out[n] = r.work;
            names += r.name;
            names += '\n';
            ++n;
        }
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof.clear();
    if (names_out && names_cap) {
        const size_t k = std::min(names_cap - 1, names.size());
        std::memcpy(names_out, names.data(), k);
        names_out[k] = 0;
    }
    return n;
}

}  // extern "C"
