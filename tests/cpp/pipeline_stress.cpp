// Sanitizer stress job over libfrt's HOST code with a real device behind it (round-1 VERDICT, hygiene item 9): the pipeline's
// slot / staging-set / ticket bookkeeping, the per-object mutexes and stage events, the streaming gallery loader.  Built with
// clang++ -fsanitize=address,undefined and linked against libfrt_asan.so (host translation units of libfrt instrumented the same way,
// device code untouched) by tests/test_gpu_sanitizers.py.
//   pipeline_stress <det.frtw> <rec.frtw> <frames.bin: 2 batches of B frames u8 HxWx3> <B> <H> <W> <gallery.bin fp32 [N][512]> <N>
// Four threads call frt_pipeline_run concurrently (one of them mixes in object-level detector / matcher calls), the main thread keeps
// tickets in flight through submit / wait and reloads the gallery in the middle; then six threads send single frames through the request
// coalescer (frt_coalescer_*).  Every answer must equal the quiet-device answer.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include "frt.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        const int rc_ = (call);                                                             \
        if (rc_ != FRT_OK) {                                                                \
            std::fprintf(stderr, "%s failed: %d %s\n", #call, rc_, frt_last_error());      \
            std::exit(2);                                                                   \
        }                                                                                   \
    } while (0)

static std::vector<char> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc != 9) return 64;
    const int B = std::atoi(argv[4]), H = std::atoi(argv[5]), W = std::atoi(argv[6]), N = std::atoi(argv[8]), K = 4;
    const std::vector<char> fb = slurp(argv[3]), gb = slurp(argv[7]);
    const size_t batch_bytes = (size_t)B * H * W * 3;
    if (fb.size() != 2 * batch_bytes || gb.size() != (size_t)N * 512 * 4) return 65;
    const uint8_t *frames[2] = {reinterpret_cast<const uint8_t *>(fb.data()), reinterpret_cast<const uint8_t *>(fb.data()) + batch_bytes};
    const float *gallery = reinterpret_cast<const float *>(gb.data());

    frt_detector *det = nullptr;
    frt_embedder *emb = nullptr;
    frt_matcher *mat = nullptr;
    frt_pipeline *pipe = nullptr;
    CHECK(frt_detector_create(argv[1], W, H, 3, H, W, B, K, 0.4f, 0.6f, 0, &det));
    CHECK(frt_embedder_create(argv[2], 3, 112, 112, 512, B * K, 0, &emb));
    CHECK(frt_matcher_create(0, &mat));
    // streaming load in ragged pieces (addEmbedding-style single rows, then blocks)
    CHECK(frt_matcher_gallery_begin(mat, N, 512));
    for (int i = 0; i < 5; ++i) CHECK(frt_matcher_gallery_append(mat, gallery + (size_t)i * 512, 1));
    CHECK(frt_matcher_gallery_append(mat, gallery + 5 * 512, N - 5));
    CHECK(frt_matcher_gallery_commit(mat));
    CHECK(frt_pipeline_create(det, emb, mat, B, &pipe));

    const int F = B * K;
    std::vector<frt_face_result> want[2];
    for (int k = 0; k < 2; ++k) {
        want[k].resize(F);
        CHECK(frt_pipeline_run(pipe, frames[k], B, want[k].data(), nullptr));
    }
    std::vector<frt_bbox> want_boxes(K);
    int want_n = 0;
    CHECK(frt_detector_find_faces(det, frames[1], H, W, (size_t)W * 3, want_boxes.data(), &want_n));
    std::vector<float> q(2 * 512);
    std::memcpy(q.data(), gallery + 7 * 512, 512 * 4);
    std::memcpy(q.data() + 512, gallery + (size_t)(N - 3) * 512, 512 * 4);
    int32_t want_idx[2];
    float want_sim[2];
    CHECK(frt_matcher_top1(mat, q.data(), 2, want_idx, want_sim));

    std::atomic<int> bad{0};
    auto same = [&](const std::vector<frt_face_result> &a, const std::vector<frt_face_result> &b) {
        return std::memcmp(a.data(), b.data(), sizeof(frt_face_result) * a.size()) == 0;
    };
    auto worker = [&](int t) {
        std::vector<frt_face_result> res(F);
        std::vector<float> embeds((size_t)F * 512);
        for (int it = 0; it < 12; ++it) {
            const int k = (t + it) & 1;
            CHECK(frt_pipeline_run(pipe, frames[k], B, res.data(), (it & 3) == 0 ? embeds.data() : nullptr));
            if (!same(res, want[k])) bad.fetch_add(1);
            if (t == 3) {  // object-level calls while other threads' stages are in flight
                std::vector<frt_bbox> boxes(K);
                int n = 0;
                CHECK(frt_detector_find_faces(det, frames[1], H, W, (size_t)W * 3, boxes.data(), &n));
                if (n != want_n || std::memcmp(boxes.data(), want_boxes.data(), sizeof(frt_bbox) * n)) bad.fetch_add(1);
                int32_t idx[2];
                float sim[2];
                CHECK(frt_matcher_top1(mat, q.data(), 2, idx, sim));
                if (idx[0] != want_idx[0] || idx[1] != want_idx[1] || sim[0] != want_sim[0]) bad.fetch_add(1);
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t) th.emplace_back(worker, t);
    {  // main thread: tickets in flight, waited out of order, a gallery reload in the middle
        std::vector<std::vector<frt_face_result>> res(10, std::vector<frt_face_result>(F));
        long tickets[10];
        for (int i = 0; i < 10; ++i) {
            CHECK(frt_pipeline_submit(pipe, frames[i & 1], B, res[i].data(), nullptr, &tickets[i]));
            if (i == 4) {
                CHECK(frt_matcher_gallery_begin(mat, N, 512));
                CHECK(frt_matcher_gallery_append(mat, gallery, N));
                CHECK(frt_matcher_gallery_commit(mat));
            }
            if (i >= 3) CHECK(frt_pipeline_wait(pipe, tickets[i - 3]));
        }
        for (int i = 9; i >= 7; --i) CHECK(frt_pipeline_wait(pipe, tickets[i]));
        for (int i = 0; i < 10; ++i)
            if (!same(res[i], want[i & 1])) bad.fetch_add(1);
        if (frt_pipeline_wait(pipe, tickets[9] + 100) == FRT_OK) bad.fetch_add(1);  // unknown ticket must be an error
    }
    for (std::thread &t : th) t.join();
    // ---- round 4: the request coalescer (csrc/frt_coalesce.cpp: dispatcher + completer threads, four staging sets, per-batch condition
    //      variables) under six caller threads, each sending single frames; every frame's boxes / matches must be its own quiet answer
    {
        frt_coalescer *co = nullptr;
        CHECK(frt_coalescer_create(det, emb, mat, B, 150, &co));
        const size_t fbytes = (size_t)H * W * 3;
        auto cworker = [&](int t) {
            std::vector<frt_face_result> res(K);
            std::vector<float> embeds((size_t)K * 512);
            std::vector<uint8_t> crops((size_t)K * 112 * 112 * 3);
            for (int it = 0; it < 10; ++it) {
                const int k = (t + it) & 1, fidx = (t * 3 + it) % B;
                int n = 0;
                CHECK(frt_coalescer_infer_crops(co, frames[k] + (size_t)fidx * fbytes, H, W, (size_t)W * 3, res.data(), (it & 1) ? embeds.data() : nullptr,
                                                (it % 3) == 0 ? crops.data() : nullptr, &n));
                const frt_face_result *w = &want[k][(size_t)fidx * K];
                for (int j = 0; j < K; ++j)
                    if (std::memcmp(&res[j].box, &w[j].box, sizeof(frt_bbox)) || res[j].match_idx != w[j].match_idx || res[j].valid != w[j].valid) bad.fetch_add(1);
            }
        };
        std::vector<std::thread> ct;
        for (int t = 0; t < 6; ++t) ct.emplace_back(cworker, t);
        for (std::thread &t : ct) t.join();
        long nb = 0, nf = 0;
        CHECK(frt_coalescer_stats(co, &nb, &nf));
        if (nf != 60 || nb < 1 || nb > 60) bad.fetch_add(1);
        if (frt_coalescer_infer(co, frames[0], H / 2, W, (size_t)W * 3, want[0].data(), nullptr, nullptr) == FRT_OK) bad.fetch_add(1);  // wrong frame size
        frt_coalescer_destroy(co);
    }
    frt_pipeline_destroy(pipe);
    frt_matcher_destroy(mat);
    frt_embedder_destroy(emb);
    frt_detector_destroy(det);
    std::printf("stress %s (%d mismatches)\n", bad.load() ? "FAILED" : "ok", bad.load());
    std::fflush(stdout);
    // _Exit, not return: ROCm's ASan runtime asserts inside libamdhip64's own static finaliser ("dev_runtime_unloaded_" in
    // sanitizer_allocator_device.h) when the HIP runtime frees host memory after ASan's device hooks are gone - not libfrt's code,
    // and every libfrt object is already destroyed above.
    std::_Exit(bad.load() ? 1 : 0);
}
