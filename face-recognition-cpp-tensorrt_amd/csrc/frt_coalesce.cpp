// Request coalescer: concurrent ONE-frame requests -> pipeline batches.
//
// The reference serves one frame per request (src/app.cpp:293-352: findFace -> forward -> featureMatching -> getOutputs) from a Crow
// server started with .multithreaded() (src/app.cpp:367).  Through the drop-in shells that call sequence costs 1.35 ms per frame and
// reaches 5.9 k faces/s with 8 request threads (profiles/r03/r03zf_dropin_bench.json) where frt_pipeline_submit / wait sustains 39 k: one
// frame is 4 faces, and a 4-face pass leaves the chip idle.  This object gathers the frames of requests that are in the building at
// the same time into ONE frt_pipeline_submit:
//   * a caller copies its frame into the next slot of the open batch's pinned staging buffer (the callers' memcpys run in parallel)
//     and sleeps on the batch;
//   * a dispatcher thread closes the open batch when it is full, or when its first frame has waited `window_us` and the pipeline has
//     room (fewer than kInflight (3) batches queued) - so an idle server answers a lone request after at most the window, and a loaded
//     one fills its batches while the previous ones run (the batch size adapts to the load, there is no fixed batch);
//   * a completer thread waits for the tickets in order and wakes the callers of a finished batch, which copy their slots' results out.
// Results: boxes are the detector's (bit-identical across batch sizes, tests/test_gpu_detector.py), embeddings / similarities are
// those of the batch size the frame happened to travel in (kernels are chosen by batch size; 1 - cos <= 1e-5 against the lone call,
// same top-1 rows - tests/test_gpu_coalesce.py).
// Uses only the public C ABI (include/frt.h).
#include <chrono>
#include <condition_variable>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/frt.h"
#include "frt_host.hpp"

using frthost::guarded;
using frthost::raise;

namespace {
typedef std::chrono::steady_clock Clock;
}

struct frt_coalescer {
    static constexpr int kBatches = 4;   // staging sets (the pipeline itself holds up to 4 batches in flight)
    static constexpr size_t kCropBytes = 112 * 112 * 3;
    int kInflight = 3;  // batches queued on the pipeline before the dispatcher lets the open one grow (FRT_COALESCE_INFLIGHT: 1 .. 3; measured at 8 / 32 threads: 1 -> 10.2 / 19.6 k faces/s, 2 -> 12.9 / 25.8 k, 3 -> 13.9 / 27.9 k)
    enum State { FREE, OPEN, CLOSED, DONE };
    struct Batch {
        uint8_t *h_frames = nullptr;
        frt_face_result *h_results = nullptr;
        float *h_embeds = nullptr;
        uint8_t *h_crops = nullptr;
        State state = FREE;
        int joined = 0, filled = 0, readers = 0;
        bool want_crops = false;  // some caller of this batch asked for the u8 crops
        long ticket = -1;
        int rc = FRT_OK;
        std::string err;
        Clock::time_point first;
        std::condition_variable cv_done;
    };
    frt_pipeline *pipe = nullptr;
    int device = 0;
    int frame_w = 0, frame_h = 0, max_frames = 0, max_faces = 0;
    size_t fbytes = 0;
    long window_us = 0;
    Batch batch[kBatches];
    int open = -1;  // index of the OPEN batch (-1: none free right now)
    int inflight = 0;
    std::deque<int> queue;  // submitted batches, in ticket order
    bool stop = false;
    bool disp_done = false;  // the dispatcher has exited: no further ticket will be queued (the completer may leave once the queue is empty)
    int active = 0;          // callers inside frt_coalescer_infer_crops (destroy waits for them before it frees anything)
    std::mutex mu;
    std::condition_variable cv_disp, cv_comp, cv_space, cv_idle;
    std::thread t_disp, t_comp;
    long n_batches = 0, n_frames = 0;

    void open_next_locked() {
        if (open >= 0) return;
        for (int i = 0; i < kBatches; ++i)
            if (batch[i].state == FREE) {
                batch[i].state = OPEN;
                batch[i].joined = batch[i].filled = batch[i].readers = 0;
                batch[i].want_crops = false;
                batch[i].rc = FRT_OK;
                batch[i].err.clear();
                open = i;
                cv_space.notify_all();
                return;
            }
    }

    void dispatcher() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_disp.wait(lk, [&] { return stop || (open >= 0 && batch[open].joined > 0); });
            if (stop) return;
            Batch &b = batch[open];
            // grow the batch while the pipeline is busy anyway; once there is room, only until the first frame has waited the window
            for (;;) {
                if (stop) return;
                if (b.joined >= max_frames) break;
                if (inflight < kInflight) {
                    const long waited = (long)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - b.first).count();
                    if (waited >= window_us) break;
                    cv_disp.wait_for(lk, std::chrono::microseconds(window_us - waited));
                } else {
                    cv_disp.wait(lk);
                }
            }
            const int bi = open;
            b.state = CLOSED;
            open = -1;
            open_next_locked();  // later arrivals join the next batch
            cv_disp.wait(lk, [&] { return stop || b.filled == b.joined; });  // every joined frame has been copied into the staging buffer
            if (stop) return;
            const int n = b.joined;
            lk.unlock();
            long ticket = -1;
            const int rc = frt_pipeline_submit_crops(pipe, b.h_frames, n, b.h_results, b.h_embeds, b.want_crops ? b.h_crops : nullptr, &ticket);
            lk.lock();
            ++n_batches;
            n_frames += n;
            if (rc != FRT_OK) {
                b.rc = rc;
                b.err = frt_last_error();
                b.state = DONE;
                b.readers = n;
                b.cv_done.notify_all();
                continue;
            }
            b.ticket = ticket;
            ++inflight;
            queue.push_back(bi);
            cv_comp.notify_one();
        }
    }

    void completer() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            // (shutdown: every ticket the dispatcher submitted is waited for - the device work and the D2H copies into the pinned buffers
            //  must have finished before destroy frees them; the dispatcher is joined first and sets disp_done)
            cv_comp.wait(lk, [&] { return disp_done || !queue.empty(); });
            if (queue.empty()) return;
            const int bi = queue.front();
            queue.pop_front();
            Batch &b = batch[bi];
            lk.unlock();
            const int rc = frt_pipeline_wait(pipe, b.ticket);
            lk.lock();
            if (rc != FRT_OK) {
                b.rc = rc;
                b.err = frt_last_error();
            }
            --inflight;
            b.state = DONE;
            b.readers = b.joined;
            b.cv_done.notify_all();
            cv_disp.notify_all();
        }
    }
};

extern "C" {

int frt_coalescer_create(frt_detector *d, frt_embedder *e, frt_matcher *m, int max_frames, int window_us, frt_coalescer **out) {
    return guarded([&] {
        if (!d || !e || !out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        int fw = 0, fh = 0, mb = 0, mf = 0, dev = 0;
        if (frt_detector_geometry(d, &fw, &fh, &mb, &mf, &dev) != FRT_OK) raise(FRT_ERR_INVALID, "coalescer: bad detector");
        if (max_frames < 1 || max_frames > mb) raise(FRT_ERR_CAPACITY, "coalescer: max_frames exceeds the detector's max_batch");
        if (window_us < 0 || window_us > 1000000) raise(FRT_ERR_INVALID, "coalescer: window_us out of range");
        std::unique_ptr<frt_coalescer> c(new frt_coalescer());
        c->device = dev;
        c->frame_w = fw;
        c->frame_h = fh;
        c->max_frames = max_frames;
        c->max_faces = mf;
        c->fbytes = (size_t)fw * fh * 3;
        c->window_us = window_us;
#ifdef FRT_TUNING  // (measurement builds only: batches in flight behind the coalescer)
        if (const char *e = getenv("FRT_COALESCE_INFLIGHT")) c->kInflight = std::max(1, std::min(3, atoi(e)));
#endif
        auto cleanup = [&] {
            for (auto &b : c->batch) {
                frt_pinned_free(b.h_frames);
                frt_pinned_free(b.h_results);
                frt_pinned_free(b.h_embeds);
                frt_pinned_free(b.h_crops);
            }
            frt_pipeline_destroy(c->pipe);
        };
        int rc = frt_pipeline_create(d, e, m, max_frames, &c->pipe);
        if (rc != FRT_OK) raise(rc, frt_last_error());
        const size_t F = (size_t)max_frames * mf;
        for (auto &b : c->batch) {
            void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
            if (frt_pinned_alloc(c->fbytes * max_frames, dev, &p0) != FRT_OK || frt_pinned_alloc(sizeof(frt_face_result) * F, dev, &p1) != FRT_OK ||
                frt_pinned_alloc(sizeof(float) * 512 * F, dev, &p2) != FRT_OK || frt_pinned_alloc(frt_coalescer::kCropBytes * F, dev, &p3) != FRT_OK) {
                frt_pinned_free(p0);
                frt_pinned_free(p1);
                frt_pinned_free(p2);
                frt_pinned_free(p3);
                const std::string msg = frt_last_error();
                cleanup();
                raise(FRT_ERR_DEVICE, msg);
            }
            b.h_frames = static_cast<uint8_t *>(p0);
            b.h_results = static_cast<frt_face_result *>(p1);
            b.h_embeds = static_cast<float *>(p2);
            b.h_crops = static_cast<uint8_t *>(p3);
        }
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->open_next_locked();
        }
        frt_coalescer *raw = c.get();
        c->t_disp = std::thread([raw] { raw->dispatcher(); });
        c->t_comp = std::thread([raw] { raw->completer(); });
        *out = c.release();
    });
}

void frt_coalescer_destroy(frt_coalescer *c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->stop = true;
    }
    c->cv_disp.notify_all();
    c->cv_space.notify_all();
    // 1. the dispatcher first: it may be inside frt_pipeline_submit_crops - the ticket it gets is queued before it sees `stop`
    if (c->t_disp.joinable()) c->t_disp.join();
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->disp_done = true;
    }
    c->cv_comp.notify_all();
    // 2. the completer drains every submitted ticket (frt_pipeline_wait: kernels and result copies of those batches have finished)
    if (c->t_comp.joinable()) c->t_comp.join();
    {
        std::unique_lock<std::mutex> lk(c->mu);
        // 3. callers still parked on a batch that never ran (destroying an object in use is the caller's bug; do not leave them asleep)
        for (auto &b : c->batch) {
            if (b.state == frt_coalescer::OPEN || b.state == frt_coalescer::CLOSED) {
                b.rc = FRT_ERR_INVALID;
                b.err = "coalescer destroyed with requests in flight";
                b.state = frt_coalescer::DONE;
                b.readers = b.joined;
            }
            b.cv_done.notify_all();
        }
        c->open = -1;
        // 4. ... and wait until the last of them has left frt_coalescer_infer_crops: they still read the batch's buffers and this mutex
        c->cv_idle.wait(lk, [&] { return c->active == 0; });
    }
    frt_pipeline_destroy(c->pipe);
    for (auto &b : c->batch) {
        frt_pinned_free(b.h_frames);
        frt_pinned_free(b.h_results);
        frt_pinned_free(b.h_embeds);
        frt_pinned_free(b.h_crops);
    }
    delete c;
}

int frt_coalescer_infer_crops(frt_coalescer *c, const uint8_t *bgr, int rows, int cols, size_t row_stride, frt_face_result *results, float *embeds_out,
                              uint8_t *crops_out, int *n_boxes) {
    return guarded([&] {
        if (!c || !bgr || !results) raise(FRT_ERR_INVALID, "null argument");
        if (rows != c->frame_h || cols != c->frame_w) raise(FRT_ERR_INVALID, "coalescer: frame size differs from the detector's frame size");
        if (row_stride < (size_t)cols * 3) raise(FRT_ERR_INVALID, "coalescer: row stride smaller than a row");
        std::unique_lock<std::mutex> lk(c->mu);
        if (c->stop) raise(FRT_ERR_INVALID, "coalescer: shutting down");
        struct Active {  // counted while this call may touch the coalescer; frt_coalescer_destroy waits for zero
            frt_coalescer *c;
            explicit Active(frt_coalescer *c_) : c(c_) { ++c->active; }   // (under c->mu)
            ~Active() {
                std::lock_guard<std::mutex> g(c->mu);
                if (--c->active == 0) c->cv_idle.notify_all();
            }
        } active_guard(c);
        c->cv_space.wait(lk, [&] { return c->stop || (c->open >= 0 && c->batch[c->open].joined < c->max_frames); });
        if (c->stop) {
            lk.unlock();  // (the guard's destructor takes the mutex)
            raise(FRT_ERR_INVALID, "coalescer: shutting down");
        }
        const int bi = c->open;
        frt_coalescer::Batch &b = c->batch[bi];
        const int slot = b.joined++;
        if (slot == 0) b.first = Clock::now();
        if (crops_out) b.want_crops = true;
        c->cv_disp.notify_all();
        lk.unlock();
        uint8_t *dst = b.h_frames + (size_t)slot * c->fbytes;
        if (row_stride == (size_t)cols * 3) {
            std::memcpy(dst, bgr, c->fbytes);
        } else {
            for (int r = 0; r < rows; ++r) std::memcpy(dst + (size_t)r * cols * 3, bgr + (size_t)r * row_stride, (size_t)cols * 3);
        }
        lk.lock();
        ++b.filled;
        c->cv_disp.notify_all();
        b.cv_done.wait(lk, [&] { return b.state == frt_coalescer::DONE; });
        const int rc = b.rc;
        const std::string err = b.err;
        int nb = 0;
        lk.unlock();  // the batch stays DONE until its last reader has left: the copies below need no lock (32 callers x 150 KB of crops)
        if (rc == FRT_OK) {
            const frt_face_result *src = b.h_results + (size_t)slot * c->max_faces;
            for (int k = 0; k < c->max_faces; ++k) {
                results[k] = src[k];
                results[k].frame = 0;  // the caller's one frame
                if (src[k].box.score > 0.f) nb = k + 1;  // unused slots read as zeros (frt_face_result)
            }
            if (embeds_out) std::memcpy(embeds_out, b.h_embeds + (size_t)slot * c->max_faces * 512, sizeof(float) * 512 * c->max_faces);
            if (crops_out) std::memcpy(crops_out, b.h_crops + (size_t)slot * c->max_faces * frt_coalescer::kCropBytes, frt_coalescer::kCropBytes * c->max_faces);
        }
        lk.lock();
        if (--b.readers == 0) {  // last reader out: the staging set is free again
            b.state = frt_coalescer::FREE;
            if (!c->stop) c->open_next_locked();
            c->cv_disp.notify_all();
        }
        lk.unlock();
        if (rc != FRT_OK) raise(rc, err);
        if (n_boxes) *n_boxes = nb;
    });
}

int frt_coalescer_infer(frt_coalescer *c, const uint8_t *bgr, int rows, int cols, size_t row_stride, frt_face_result *results, float *embeds_out,
                        int *n_boxes) {
    return frt_coalescer_infer_crops(c, bgr, rows, cols, row_stride, results, embeds_out, nullptr, n_boxes);
}

int frt_coalescer_stats(frt_coalescer *c, long *batches_out, long *frames_out) {
    return guarded([&] {
        if (!c) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(c->mu);
        if (batches_out) *batches_out = c->n_batches;
        if (frames_out) *frames_out = c->n_frames;
    });
}

}  // extern "C"
