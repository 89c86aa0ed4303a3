// ONE process, all (or the listed) devices: the reference's own deployment shape - a single Crow process owns the machine
// (src/app.cpp:52-57, 367) - for the batched path: one host thread per device drives that device's pipeline (frt_pipeline_submit / wait,
// three batches in flight, its own 32 frames per step and a full gallery replica: weak scaling, BASELINE configs[3]) and after every step
// the per-face result records of all devices are exchanged with ONE ncclAllGather per device issued from one thread
// (frt_comm_create_all + frt_comm_all_gather_multi: RCCL bound by libfrt.so itself, no Python, no torch, no MPI).  bench.py measures the
// one-process-per-GPU form (torch.distributed.run); this is the same step without a launcher.
//   multi_device_pipeline <det.frtw> <rec.frtw> <frames.bin (u8 BGR [n][640][640][3])> <frames per step per device> <gallery rows> <steps> <devices e.g. 0,1,2,3 | all>
// Prints one JSON line; exit code 0 only if every device received every other device's records of the last step.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "frt.h"

#define CK(x)                                                                          \
    do {                                                                               \
        if ((x) != 0) {                                                                \
            std::fprintf(stderr, "FAILED %s: %s\n", #x, frt_last_error());             \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)
#define HK(x)                                                                          \
    do {                                                                               \
        if ((x) != hipSuccess) {                                                       \
            std::fprintf(stderr, "FAILED %s\n", #x);                                   \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

struct Barrier {  // (C++11: no std::barrier)
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    long gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    template <class F>
    void arrive(F &&last) {  // `last` runs once per generation, on the thread that arrives last, before anybody is released
        std::unique_lock<std::mutex> lk(mu);
        const long g = gen;
        if (++waiting == n) {
            last();
            waiting = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};

struct Dev {
    int device;
    frt_detector *det = nullptr;
    frt_embedder *emb = nullptr;
    frt_matcher *mat = nullptr;
    frt_pipeline *pipe = nullptr;
    uint8_t *h_frames[2] = {nullptr, nullptr};
    frt_face_result *h_res[4] = {};
    void *d_send = nullptr, *d_recv = nullptr;
    hipStream_t side = nullptr;
    long faces = 0;
};

int main(int argc, char **argv) {
    if (argc != 8) {
        std::fprintf(stderr, "usage: see the header comment\n");
        return 2;
    }
    const int B = std::atoi(argv[4]), N = std::atoi(argv[5]), steps = std::atoi(argv[6]), K = 4, H = 640, W = 640;
    std::vector<int> devices;
    if (std::string(argv[7]) == "all") {
        for (int d = 0; d < frt_device_count(); ++d) devices.push_back(d);
    } else {
        for (const char *p = argv[7]; *p;) {
            devices.push_back(std::atoi(p));
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
    }
    const int D = (int)devices.size();
    if (D < 1 || B < 1 || steps < 4) return 2;
    std::ifstream f(argv[3], std::ios::binary);
    std::vector<unsigned char> fb((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const size_t fbytes = (size_t)H * W * 3;
    const int n_frames = (int)(fb.size() / fbytes);
    if (n_frames < 1) return 2;
    std::vector<float> gal((size_t)N * 512);
    {
        unsigned long long x = 88172645463325252ull;
        for (int r = 0; r < N; ++r) {
            double n2 = 0;
            float *row = &gal[(size_t)r * 512];
            for (int k = 0; k < 512; ++k) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                row[k] = (float)(x >> 40) * (1.0f / 16777216.0f) - 0.5f;
                n2 += (double)row[k] * row[k];
            }
            const float inv = (float)(1.0 / std::sqrt(n2));
            for (int k = 0; k < 512; ++k) row[k] *= inv;
        }
    }
    const int F = B * K;
    const size_t rec_bytes = sizeof(frt_face_result) * (size_t)F;
    std::vector<Dev> dev((size_t)D);
    for (int i = 0; i < D; ++i) {
        Dev &d = dev[(size_t)i];
        d.device = devices[(size_t)i];
        CK(frt_detector_create(argv[1], W, H, 3, H, W, B, K, 0.4f, 0.6f, d.device, &d.det));
        CK(frt_embedder_create(argv[2], 3, 112, 112, 512, F, d.device, &d.emb));
        CK(frt_matcher_create(d.device, &d.mat));
        CK(frt_matcher_init(d.mat, gal.data(), N, 512));
        CK(frt_pipeline_create(d.det, d.emb, d.mat, B, &d.pipe));
        for (int j = 0; j < 2; ++j) {
            void *p = nullptr;
            CK(frt_pinned_alloc(fbytes * B, d.device, &p));
            d.h_frames[j] = static_cast<uint8_t *>(p);
            for (int b = 0; b < B; ++b) std::memcpy(d.h_frames[j] + (size_t)b * fbytes, &fb[(size_t)((i * B + b + j * 7) % n_frames) * fbytes], fbytes);
        }
        for (int j = 0; j < 4; ++j) {
            void *p = nullptr;
            CK(frt_pinned_alloc(rec_bytes, d.device, &p));
            d.h_res[j] = static_cast<frt_face_result *>(p);
        }
        HK(hipSetDevice(d.device));
        HK(hipMalloc(&d.d_send, rec_bytes));
        HK(hipMalloc(&d.d_recv, rec_bytes * D));
        HK(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
    }
    // the communicators come last (they own streams; see INTEGRATION.md "Streams and hardware queues")
    std::vector<frt_comm *> comms((size_t)D, nullptr);
    CK(frt_comm_create_all(D, devices.data(), comms.data()));

    Barrier bar(D);
    std::vector<const void *> send((size_t)D);
    std::vector<void *> recv((size_t)D), streams((size_t)D);
    for (int i = 0; i < D; ++i) {
        send[(size_t)i] = dev[(size_t)i].d_send;
        recv[(size_t)i] = dev[(size_t)i].d_recv;
        streams[(size_t)i] = dev[(size_t)i].side;
    }
    auto run = [&](int n_steps, bool count) {
        std::vector<std::thread> th;
        for (int i = 0; i < D; ++i)
            th.emplace_back([&, i] {
                Dev &d = dev[(size_t)i];
                HK(hipSetDevice(d.device));
                long tickets[4] = {-1, -1, -1, -1};
                for (int s = 0; s < n_steps + 3; ++s) {
                    if (s >= 3) {  // step s - 3 completes: its records go to the exchange
                        const int done = s - 3, slot = done % 4;
                        CK(frt_pipeline_wait(d.pipe, tickets[slot]));
                        if (count)
                            for (int k = 0; k < F; ++k) d.faces += d.h_res[slot][k].valid == 1;
                        HK(hipStreamSynchronize(d.side));  // the previous exchange has left d_send
                        HK(hipMemcpyAsync(d.d_send, d.h_res[slot], rec_bytes, hipMemcpyHostToDevice, d.side));
                        bar.arrive([&] {  // every device's records of this step are queued: one grouped ncclAllGather for all devices
                            CK(frt_comm_all_gather_multi(D, comms.data(), send.data(), recv.data(), rec_bytes, streams.data()));
                        });
                    }
                    if (s < n_steps) CK(frt_pipeline_submit(d.pipe, d.h_frames[s & 1], B, d.h_res[s % 4], nullptr, &tickets[s % 4]));
                }
                HK(hipStreamSynchronize(d.side));
            });
        for (std::thread &t : th) t.join();
    };
    run(6, false);
    const auto t0 = std::chrono::steady_clock::now();
    run(steps, true);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // every device holds every device's records of the last step
    int bad = 0;
    long faces = 0;
    for (int i = 0; i < D; ++i) {
        faces += dev[(size_t)i].faces;
        std::vector<unsigned char> h(rec_bytes * D);
        HK(hipSetDevice(dev[(size_t)i].device));
        HK(hipMemcpy(h.data(), dev[(size_t)i].d_recv, h.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < D; ++r)
            if (std::memcmp(&h[(size_t)r * rec_bytes], dev[(size_t)r].h_res[(steps - 1) % 4], rec_bytes) != 0) ++bad;
    }
    std::printf("{\"shape\": \"one process, one host thread per device, grouped ncclAllGather of every step's records\", \"devices\": %d, \"frames_per_step_per_device\": %d, "
                "\"gallery_rows\": %d, \"steps\": %d, \"faces\": %ld, \"faces_per_sec\": %.1f, \"ms_per_step\": %.4f, \"gather_mismatches\": %d}\n",
                D, B, N, steps, faces, 1e3 * faces / ms, ms / steps, bad);
    for (int i = 0; i < D; ++i) {
        Dev &d = dev[(size_t)i];
        frt_comm_destroy(comms[(size_t)i]);
        frt_pipeline_destroy(d.pipe);
        frt_matcher_destroy(d.mat);
        frt_embedder_destroy(d.emb);
        frt_detector_destroy(d.det);
    }
    return bad ? 3 : 0;
}
