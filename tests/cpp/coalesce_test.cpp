// Request coalescing behind the drop-in shells (include/frt/coalesce.h, frt_coalescer_*): T threads share ONE detector and ONE recogniser
// (the reference's shape: global objects captured by reference in every handler, src/app.cpp:52-57,293) and run the /inference call
// sequence (src/app.cpp:304-310) on their own frames - first plain (every call goes to the device on its own, calls serialise on the
// objects), then after recognizer.coalesceWith(detector): per call the boxes must be identical, the crops (croppedFaces[i].face, what the
// reply JPEG-encodes, src/app.cpp:328) and the recogniser input tensors (faceMat) byte-identical, the names identical, similarities and
// embeddings equal up to the kernels' batch-size classes (1 - cos <= 1e-5, |dsim| <= 1e-5).
//   coalesce_test <det.frtw> <rec.frtw> <frames.bin> <n_frames> <rows> <cols> <gallery.bin> <n> <threads> <coalesce frames>
// Exit code 0 + "coalesce ok ..." on success.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>

#include "frt/arcface.h"
#include "frt/retinaface.h"

struct CallResult {
    std::vector<Bbox> boxes;
    std::vector<std::string> names;
    std::vector<float> sims, embeds;
    std::vector<std::vector<unsigned char>> crops;
    std::vector<std::vector<float>> tensors;
};

static std::vector<char> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc != 11) return 2;
    const int n_frames = std::atoi(argv[4]), rows = std::atoi(argv[5]), cols = std::atoi(argv[6]), n = std::atoi(argv[8]), T = std::atoi(argv[9]),
              cf = std::atoi(argv[10]);
    std::vector<char> fb = slurp(argv[3]), gb = slurp(argv[7]);
    if (fb.size() < (size_t)n_frames * rows * cols * 3 || gb.size() < (size_t)n * 512 * 4) return 2;
    TRTLogger gLogger;
    RetinaFace detector(gLogger, argv[1], cols, rows, "input_det", {"output_det0", "output_det1"}, {3, rows, cols}, cf, 4, 0.4f, 0.6f);
    ArcFaceIR50 recognizer(gLogger, argv[2], cols, rows, "input", "output", {3, 112, 112}, 512, 4 * cf, 4, 0.65f);
    recognizer.initKnownEmbeds(n);
    const float *g = reinterpret_cast<const float *>(gb.data());
    for (int i = 0; i < n; ++i) recognizer.addEmbedding(std::to_string(i), const_cast<float *>(g + (size_t)i * 512));
    recognizer.initMatMul();
    recognizer.setMaterializeSimilarities(false);

    std::vector<std::vector<CallResult>> res[2];
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) recognizer.coalesceWith(detector, cf, 200);
        res[pass].assign((size_t)T, std::vector<CallResult>());
        std::atomic<int> failed(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t, pass] {
                try {
                    for (int it = 0; it < n_frames; ++it) {
                        cv::Mat frame(rows, cols, CV_8UC3, &fb[(size_t)((it + t) % n_frames) * rows * cols * 3]);
                        CallResult r;
                        // src/app.cpp:304-310
                        std::vector<struct Bbox> outputBbox = detector.findFace(frame);
                        r.boxes = outputBbox;
                        if (!outputBbox.empty()) {
                            recognizer.forward(frame, outputBbox);
                            float *output_sims = recognizer.featureMatching();
                            std::tie(r.names, r.sims) = recognizer.getOutputs(output_sims);
                            r.embeds.assign(recognizer.embeddings(), recognizer.embeddings() + outputBbox.size() * 512);
                            for (size_t i = 0; i < recognizer.croppedFaces.size(); ++i) {
                                const CroppedFace &c = recognizer.croppedFaces[i];
                                r.crops.emplace_back(c.face.data, c.face.data + 112 * 112 * 3);
                                r.tensors.emplace_back(c.faceMat.ptr<float>(0), c.faceMat.ptr<float>(0) + 3 * 112 * 112);
                            }
                        }
                        res[pass][(size_t)t].push_back(r);
                    }
                } catch (const char *s) {
                    std::fprintf(stderr, "thread %d: %s\n", t, s);
                    ++failed;
                } catch (const std::exception &e) {
                    std::fprintf(stderr, "thread %d: %s\n", t, e.what());
                    ++failed;
                }
            });
        for (std::thread &x : th) x.join();
        if (failed) return 4;
    }
    long batches = 0, frames = 0;
    if (!detector.coalesceStats(batches, frames) || frames != (long)T * n_frames) {
        std::fprintf(stderr, "coalescer carried %ld frames, expected %ld\n", frames, (long)T * n_frames);
        return 5;
    }
    double worst_cos = 0, worst_sim = 0;
    long faces = 0;
    for (int t = 0; t < T; ++t)
        for (int it = 0; it < n_frames; ++it) {
            const CallResult &a = res[0][(size_t)t][(size_t)it], &b = res[1][(size_t)t][(size_t)it];
            if (a.boxes.size() != b.boxes.size() || a.names != b.names) return 6;
            for (size_t i = 0; i < a.boxes.size(); ++i) {
                if (std::memcmp(&a.boxes[i], &b.boxes[i], sizeof(Bbox)) != 0) return 7;
                if (a.crops[i] != b.crops[i]) return 8;
                if (std::memcmp(a.tensors[i].data(), b.tensors[i].data(), a.tensors[i].size() * 4) != 0) return 9;
                double dot = 0;
                for (int k = 0; k < 512; ++k) dot += (double)a.embeds[i * 512 + k] * b.embeds[i * 512 + k];
                worst_cos = std::max(worst_cos, 1.0 - dot);
                worst_sim = std::max(worst_sim, (double)std::fabs(a.sims[i] - b.sims[i]));
                ++faces;
            }
        }
    if (worst_cos > 1e-5 || worst_sim > 1e-5 || faces == 0) {
        std::fprintf(stderr, "1 - cos %.3g, |dsim| %.3g, faces %ld\n", worst_cos, worst_sim, faces);
        return 10;
    }
    std::printf("coalesce ok: %d threads x %d frames, %ld faces, %ld batches for %ld frames, 1 - cos <= %.3g, |dsim| <= %.3g\n", T, n_frames, faces, batches,
                frames, worst_cos, worst_sim);
    return batches < frames ? 0 : 11;  // (11: nothing was ever coalesced)
}
