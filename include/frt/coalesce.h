// Shared plumbing of the two drop-in shells (retinaface.h, arcface.h) for
//   (1) per-thread call state: the reference's classes keep the results of a call in the object (croppedFaces, m_embed, the similarity
//       buffer), and its server runs .multithreaded() (src/app.cpp:367) with ONE detector and ONE recogniser captured by reference in every
//       handler (src/app.cpp:52-57, 243, 293) - two requests at once overwrite each other's results there.  Here every calling thread
//       has its own copy of that state behind the same member names, so one pair of objects (and ONE gallery on the device) can be
//       shared by all request threads;
//   (2) opt-in request coalescing (frt_coalescer_*, include/frt.h): findFace() of concurrent requests travel through the device as one
//       pipeline batch - detector, crop, recogniser and top-1 match - and leave each thread a record that the same thread's forward() /
//       featureMatching() / getOutputs() on the same frame and boxes then answer from, without touching the device again.
//       Enable with  recognizer.coalesceWith(detector)  or, without touching the application,  FRT_COALESCE=<frames>[:<window_us>]
//       in the environment (e.g. FRT_COALESCE=32:100); default off.
#ifndef FRT_COALESCE_H
#define FRT_COALESCE_H

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace frtdetail {

inline uint64_t nextObjectId() {
    static std::atomic<uint64_t> n(1);
    return n++;
}

// Objects that have been destroyed: their per-thread slots (which may hold page-locked buffers) are dropped by each thread on its next
// perThread() access after the destruction (a thread cannot reach into another thread's thread_local storage).
struct DeadObjects {
    std::mutex mu;
    std::vector<uint64_t> ids;       // sorted
    std::atomic<uint64_t> epoch{0};  // bumped per destruction
};
inline DeadObjects &deadObjects() {
    static DeadObjects d;
    return d;
}
inline void retireObject(uint64_t id) {
    DeadObjects &d = deadObjects();
    std::lock_guard<std::mutex> lk(d.mu);
    d.ids.insert(std::upper_bound(d.ids.begin(), d.ids.end(), id), id);
    d.epoch.fetch_add(1, std::memory_order_release);
}

// the calling thread's instance of S for the object `id` (created on first use; destroyed with the thread, or on this thread's first
// access after the object itself was destroyed)
template <class S>
S &perThread(uint64_t id) {
    static thread_local std::vector<std::pair<uint64_t, std::unique_ptr<S>>> slots;
    static thread_local uint64_t seen_epoch = 0;
    DeadObjects &d = deadObjects();
    const uint64_t ep = d.epoch.load(std::memory_order_acquire);
    if (ep != seen_epoch) {
        std::lock_guard<std::mutex> lk(d.mu);
        slots.erase(std::remove_if(slots.begin(), slots.end(),
                                   [&](const std::pair<uint64_t, std::unique_ptr<S>> &e) { return std::binary_search(d.ids.begin(), d.ids.end(), e.first); }),
                    slots.end());
        seen_epoch = ep;
    }
    for (auto &e : slots)
        if (e.first == id) return *e.second;
    slots.emplace_back(id, std::unique_ptr<S>(new S()));
    return *slots.back().second;
}

// A std::vector data member of the reference's class, one instance per calling thread.  Covers what application code does with
// `recognizer.croppedFaces` (size / index / iterate / clear; src/app.cpp:316-329) and converts to the vector itself.
template <class T, class Owner, std::vector<T> &(*Get)(const Owner *)>
class PerThreadVector {
  public:
    explicit PerThreadVector(const Owner *o) : o_(o) {}
    std::vector<T> &get() const { return Get(o_); }
    operator std::vector<T> &() const { return get(); }
    size_t size() const { return get().size(); }
    bool empty() const { return get().empty(); }
    T &operator[](size_t i) const { return get()[i]; }
    T &at(size_t i) const { return get().at(i); }
    T &front() const { return get().front(); }
    T &back() const { return get().back(); }
    typename std::vector<T>::iterator begin() const { return get().begin(); }
    typename std::vector<T>::iterator end() const { return get().end(); }
    void clear() const { get().clear(); }
    void reserve(size_t n) const { get().reserve(n); }
    void push_back(const T &v) const { get().push_back(v); }

  private:
    const Owner *o_;
};

// ---- coalescing
struct CoalesceLink {  // shared by the detector shell and the recogniser shell; whichever dies first shuts the coalescer down
    std::mutex mu;
    std::condition_variable cv;
    frt_coalescer *c = nullptr;
    frt_matcher *mat = nullptr;  // the gallery the coalesced batches are matched against (for frt_matcher_generation)
    int maxFaces = 0;
    int users = 0;               // calls inside frt_coalescer_* through this link
    ~CoalesceLink() { shutdown(); }
    // a call takes the coalescer for its duration: shutdown() - run by the OTHER shell's destructor, possibly on another thread - waits for it
    frt_coalescer *acquire() {
        std::lock_guard<std::mutex> lk(mu);
        if (!c) return nullptr;
        ++users;
        return c;
    }
    void release() {
        std::lock_guard<std::mutex> lk(mu);
        if (--users == 0) cv.notify_all();
    }
    bool alive() {
        std::lock_guard<std::mutex> lk(mu);
        return c != nullptr;
    }
    void shutdown() {
        std::unique_lock<std::mutex> lk(mu);
        frt_coalescer *dead = c;
        c = nullptr;  // no new call gets it
        cv.wait(lk, [&] { return users == 0; });
        lk.unlock();
        if (dead) frt_coalescer_destroy(dead);
    }
};
struct CoalesceUse {  // RAII: acquire / release
    CoalesceLink *l;
    frt_coalescer *c;
    explicit CoalesceUse(CoalesceLink *link) : l(link), c(link ? link->acquire() : nullptr) {}
    ~CoalesceUse() {
        if (c) l->release();
    }
    CoalesceUse(const CoalesceUse &) = delete;
    CoalesceUse &operator=(const CoalesceUse &) = delete;
};

// what a coalesced findFace() left behind for the same thread's forward() / featureMatching()
struct FrameRecord {
    const CoalesceLink *link = nullptr;
    const unsigned char *data = nullptr;
    int rows = 0, cols = 0;
    uint64_t print = 0;  // fingerprint of sampled pixels: the frame must still be the one that was analysed (64 x 8 sampled bytes: a frame
                         // modified IN PLACE between findFace() and forward() - boxes drawn on it - can go unnoticed; forward() then returns the
                         // crops of the frame as it was analysed, which is what the boxes belong to)
    unsigned galleryGen = 0;  // frt_matcher_generation around the batch; 0 = the gallery changed meanwhile (match_idx / match_sim not usable)
    std::vector<Bbox> boxes;
    std::vector<frt_face_result> res;
    std::vector<float> embeds;          // [n][512]
    std::vector<unsigned char> crops;   // [n][112][112][3]
};
inline FrameRecord &frameRecord() {
    static thread_local FrameRecord r;
    return r;
}
inline uint64_t framePrint(const unsigned char *data, int rows, int cols, size_t step) {
    uint64_t h = 1469598103934665603ull;
    const size_t row_bytes = (size_t)cols * 3;
    for (int i = 0; i < 64; ++i) {  // 64 x 8 bytes spread over the frame
        const size_t r = (size_t)i * (size_t)(rows - 1) / 63, c = ((size_t)i * 2654435761u) % (row_bytes > 8 ? row_bytes - 8 : 1);
        const unsigned char *p = data + r * step + c;
        for (int k = 0; k < 8 && c + k < row_bytes; ++k) h = (h ^ p[k]) * 1099511628211ull;
    }
    return h;
}

struct CoalesceEnv {
    int frames = 0, window_us = 100;
};
inline const CoalesceEnv &coalesceEnv() {  // FRT_COALESCE=<frames>[:<window_us>]
    static const CoalesceEnv e = [] {
        CoalesceEnv v;
        if (const char *s = std::getenv("FRT_COALESCE")) {
            v.frames = std::atoi(s);
            if (v.frames < 0) v.frames = 0;
            if (v.frames > 256) v.frames = 256;
            for (const char *p = s; *p; ++p)
                if (*p == ':') {
                    v.window_us = std::atoi(p + 1);
                    break;
                }
            if (v.window_us < 0) v.window_us = 0;
        }
        return v;
    }();
    return e;
}

// objects created while FRT_COALESCE is set, waiting for their partner (same device and frame size)
struct Pending {
    int device, fw, fh, maxBatch;
    frt_detector *det;
    frt_embedder *emb;
    frt_matcher *mat;
    std::shared_ptr<CoalesceLink> *slot;  // the shell's link member
};
struct Registry {
    std::mutex mu;
    std::vector<Pending> dets, recs;
};
inline Registry &registry() {
    static Registry r;
    return r;
}
inline std::shared_ptr<CoalesceLink> makeLink(frt_detector *d, frt_embedder *e, frt_matcher *m, int frames, int window_us) {
    std::shared_ptr<CoalesceLink> l(new CoalesceLink());
    int mb = 0, mf = 0;
    checkFrtStatus(frt_detector_geometry(d, nullptr, nullptr, &mb, &mf, nullptr));
    if (frames <= 0 || frames > mb) frames = mb;
    checkFrtStatus(frt_coalescer_create(d, e, m, frames, window_us, &l->c));
    l->mat = m;
    l->maxFaces = mf;
    return l;
}
// called by both constructors under FRT_COALESCE: link with a waiting partner, or wait for one
inline void autoLink(bool isDet, const Pending &me) {
    Registry &r = registry();
    std::lock_guard<std::mutex> lk(r.mu);
    std::vector<Pending> &others = isDet ? r.recs : r.dets;
    for (size_t i = 0; i < others.size(); ++i) {
        Pending &o = others[i];
        if (o.device != me.device || o.fw != me.fw || o.fh != me.fh) continue;
        const Pending &d = isDet ? me : o, &rc = isDet ? o : me;
        std::shared_ptr<CoalesceLink> l = makeLink(d.det, rc.emb, rc.mat, coalesceEnv().frames, coalesceEnv().window_us);
        std::atomic_store(d.slot, l);   // (request threads read the shells' link members with atomic_load)
        std::atomic_store(rc.slot, l);
        others.erase(others.begin() + (long)i);
        return;
    }
    (isDet ? r.dets : r.recs).push_back(me);
}
inline void autoUnlink(bool isDet, std::shared_ptr<CoalesceLink> *slot) {
    Registry &r = registry();
    std::lock_guard<std::mutex> lk(r.mu);
    std::vector<Pending> &mine = isDet ? r.dets : r.recs;
    for (size_t i = 0; i < mine.size(); ++i)
        if (mine[i].slot == slot) {
            mine.erase(mine.begin() + (long)i);
            break;
        }
}

}  // namespace frtdetail

#endif  // FRT_COALESCE_H
