// Throughput / latency of the DROP-IN itself (round-2 VERDICT item 8): the reference's /inference call sequence, src/app.cpp:304-310,
//     outputBbox = detector.findFace(frame); recognizer.forward(frame, outputBbox);
//     output_sims = recognizer.featureMatching(); std::tie(names, sims) = recognizer.getOutputs(output_sims);
// verbatim through the C++ shells (include/frt/*.h), one frame per call like the reference, against an N-row gallery - with the full
// [F x N] similarity matrix coming back to the host exactly as MatMul::calculate specifies (src/matmul.cpp:69-72) - next to the same
// loop with the fused matchTop1() extension.  T threads model Crow's multithreaded server (src/app.cpp:367); every thread owns its
// objects (the reference's classes are not thread-safe), on the device `devices[t % n]` - the one-process / several-devices shape.
//   dropin_bench <det.frtw> <rec.frtw> <frames.bin (u8 BGR [n][rows][cols][3])> <n_frames> <rows> <cols> <gallery rows N> <threads> <iters> <devices e.g. 0,0,1>
//                [shared|own] [coalesce frames] [window us]
// "shared": ONE detector and ONE recogniser (one gallery on the device) used by all T threads - the reference's own shape (its handlers
// capture the two global objects by reference, src/app.cpp:52-57,243,293); the shells keep every call's results per calling thread.
// coalesce frames > 0 (shared only): recognizer.coalesceWith(detector, frames, window) - concurrent requests share one device batch.
// Prints one JSON line.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>

#include "frt/arcface.h"
#include "frt/retinaface.h"

typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Stat {
    double find = 0, forward = 0, match = 0, outputs = 0, total = 0, top1 = 0;
    long frames = 0, faces = 0;
    std::vector<double> lat;
};

int main(int argc, char **argv) {
    if (argc < 11 || argc > 14) {
        std::fprintf(stderr, "usage: see the header comment\n");
        return 2;
    }
    const int n_frames = std::atoi(argv[4]), rows = std::atoi(argv[5]), cols = std::atoi(argv[6]), N = std::atoi(argv[7]), T = std::atoi(argv[8]),
              iters = std::atoi(argv[9]);
    std::vector<int> devices;
    for (const char *p = argv[10]; *p;) {
        devices.push_back(std::atoi(p));
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
    }
    if (devices.empty() || T < 1 || n_frames < 1) return 2;
    const bool shared = argc > 11 && std::string(argv[11]) == "shared";
    const int co_frames = argc > 12 ? std::atoi(argv[12]) : 0, co_window = argc > 13 ? std::atoi(argv[13]) : 100;
    if (co_frames > 0 && !shared) return 2;
    const int n_sets = shared ? 1 : T;
    std::ifstream f(argv[3], std::ios::binary);
    std::vector<unsigned char> fb((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (fb.size() < (size_t)n_frames * rows * cols * 3) return 2;
    // gallery: N unit rows (xorshift -> sum of uniforms ~ normal -> normalised), the same for every thread
    std::vector<float> gal((size_t)N * 512);
    {
        unsigned long long x = 88172645463325252ull;
        for (int r = 0; r < N; ++r) {
            double n2 = 0;
            float *row = &gal[(size_t)r * 512];
            for (int k = 0; k < 512; ++k) {
                float v = 0;
                for (int j = 0; j < 4; ++j) {
                    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                    v += (float)(x >> 40) * (1.0f / 16777216.0f) - 0.5f;
                }
                row[k] = v;
                n2 += (double)v * v;
            }
            const float inv = (float)(1.0 / std::sqrt(n2));
            for (int k = 0; k < 512; ++k) row[k] *= inv;
        }
    }
    TRTLogger gLogger;
    std::vector<Stat> st((size_t)T);
    std::vector<double> load_ms((size_t)T, 0.0);
    std::vector<int> mismatch((size_t)T, 0);
    struct Objs {
        RetinaFace *det;
        ArcFaceIR50 *rec;
    };
    std::vector<Objs> objs((size_t)n_sets);
    // construction as in src/app.cpp:52-57 and gallery load as in src/db.cpp:316-346, one set of objects per thread.  Sequential: the
    // reference's `static int classCount` (src/arcface.h:39) is process-wide, so every instance is loaded from a count of zero and all
    // of them end up agreeing on classCount == N.
    for (int t = 0; t < n_sets; ++t) {
        const int dev = devices[(size_t)t % devices.size()];
        const int det_batch = co_frames > 0 ? co_frames : 1, rec_batch = co_frames > 0 ? 4 * co_frames : 4;
        objs[(size_t)t].det = new RetinaFace(gLogger, argv[1], cols, rows, "input_det", {"output_det0", "output_det1"}, {3, rows, cols}, det_batch, 4, 0.4f, 0.6f, dev);
        objs[(size_t)t].rec = new ArcFaceIR50(gLogger, argv[2], cols, rows, "input", "output", {3, 112, 112}, 512, rec_batch, 4, 0.65f, dev);
        const Clock::time_point t0 = Clock::now();
        ArcFaceIR50 &rec = *objs[(size_t)t].rec;
        ArcFaceIR50::classCount = 0;
        rec.initKnownEmbeds(N);
        for (int i = 0; i < N; ++i) rec.addEmbedding(std::string(), &gal[(size_t)i * 512]);
        rec.initMatMul();
        load_ms[(size_t)t] = ms_since(t0);
        if (co_frames > 0) rec.coalesceWith(*objs[(size_t)t].det, co_frames, co_window);
    }
    // mode 0: featureMatching + getOutputs with the matrix materialised (the reference's contract); 1: same calls, matrix not
    // materialised (setMaterializeSimilarities(false)); 2: matchTop1 extension
    double wall[3] = {0, 0, 0};
    std::vector<Stat> res[3];
    for (int mode = 0; mode < 3; ++mode) {
        st.assign((size_t)T, Stat());
        for (Objs &o : objs) o.rec->setMaterializeSimilarities(mode == 0);
        const Clock::time_point w0 = Clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t, mode] {
                RetinaFace &detector = *objs[shared ? 0 : (size_t)t].det;
                ArcFaceIR50 &recognizer = *objs[shared ? 0 : (size_t)t].rec;
                Stat &s = st[(size_t)t];
                std::vector<std::string> names;
                std::vector<float> sims;
                for (int it = -2; it < iters; ++it) {  // two warm-up calls
                    cv::Mat frame(rows, cols, CV_8UC3, &fb[(size_t)((it + 2 + t) % n_frames) * rows * cols * 3]);
                    const Clock::time_point t0 = Clock::now();
                    std::vector<struct Bbox> outputBbox = detector.findFace(frame);
                    const double a = ms_since(t0);
                    if (outputBbox.empty()) continue;
                    recognizer.forward(frame, outputBbox);
                    const double b = ms_since(t0);
                    double c, d;
                    if (mode < 2) {
                        float *output_sims = recognizer.featureMatching();
                        c = ms_since(t0);
                        std::tie(names, sims) = recognizer.getOutputs(output_sims);
                        d = ms_since(t0);
                        if (mode == 0 && it == 0) {  // the fast path equals std::max_element over the materialised matrix
                            for (size_t i = 0; i < outputBbox.size(); ++i) {
                                const float *row = output_sims + i * (size_t)ArcFaceIR50::classCount;
                                if (*std::max_element(row, row + ArcFaceIR50::classCount) != sims[i]) ++mismatch[(size_t)t];
                            }
                        }
                    } else {
                        std::tie(names, sims) = recognizer.matchTop1();
                        c = d = ms_since(t0);
                    }
                    if (it < 0) continue;
                    s.find += a; s.forward += b - a; s.match += c - b; s.outputs += d - c; s.total += d;
                    s.lat.push_back(d);
                    s.frames += 1;
                    s.faces += (long)outputBbox.size();
                }
            });
        for (std::thread &x : th) x.join();
        wall[mode] = ms_since(w0);
        res[mode] = st;
    }
    int mism = 0;
    for (int v : mismatch) mism += v;
    long co_b = 0, co_f = 0;
    objs[0].det->coalesceStats(co_b, co_f);
    std::printf("{\"gallery_rows\": %d, \"threads\": %d, \"devices\": \"%s\", \"objects\": \"%s\", \"coalesce_frames\": %d, \"coalesce_window_us\": %d, "
                "\"coalesced_batches\": %ld, \"coalesced_frames\": %ld, \"iters_per_thread\": %d, \"gallery_load_ms_per_thread\": %.1f, \"fastpath_mismatches\": %d",
                N, T, argv[10], shared ? "shared" : "one set per thread", co_frames, co_window, co_b, co_f, iters, load_ms[0], mism);
    const char *tag[3] = {"featureMatching_getOutputs", "featureMatching_getOutputs_no_matrix", "matchTop1"};
    for (int mode = 0; mode < 3; ++mode) {
        Stat a;
        std::vector<double> lat;
        for (const Stat &s : res[mode]) {
            a.find += s.find; a.forward += s.forward; a.match += s.match; a.outputs += s.outputs; a.total += s.total;
            a.frames += s.frames; a.faces += s.faces;
            lat.insert(lat.end(), s.lat.begin(), s.lat.end());
        }
        std::sort(lat.begin(), lat.end());
        const double fr = (double)std::max(a.frames, 1L);
        // per-thread wall = sum of its call latencies (threads run concurrently): aggregate rate = faces / (total / T)
        std::printf(", \"%s\": {\"wall_ms_incl_warmup\": %.1f, \"frames\": %ld, \"faces\": %ld, \"faces_per_sec\": %.1f, \"frames_per_sec\": %.1f, \"latency_ms_median\": %.3f, \"latency_ms_p95\": %.3f, "
                    "\"findFace_ms\": %.3f, \"forward_ms\": %.3f, \"match_ms\": %.3f, \"getOutputs_ms\": %.3f}",
                    tag[mode], wall[mode], a.frames, a.faces, 1e3 * a.faces / (a.total / T), 1e3 * a.frames / (a.total / T), lat.empty() ? 0.0 : lat[lat.size() / 2],
                    lat.empty() ? 0.0 : lat[(size_t)(lat.size() * 0.95)], a.find / fr, a.forward / fr, a.match / fr, a.outputs / fr);
    }
    std::printf("}\n");
    for (Objs &o : objs) {
        delete o.det;
        delete o.rec;
    }
    return mism ? 3 : 0;
}
