// libfrt.so host side, shared by its translation units (frt_api.cpp, frt_detector.cpp, frt_embedder.cpp, frt_matcher.cpp, frt_pipeline.cpp): device-memory
// arena, host waits, HIP-event profiling hooks.  Internal: nothing here is part of the C ABI (include/frt.h).
#pragma once
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
std::string &last_error();  // (frt_api.cpp)
}
using frthost::guarded;
using frthost::raise;
using frthost::use_device;

namespace frti {


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
extern std::atomic<long> g_wait_spin_us;  // (defined in frt_api.cpp)
inline long wait_spin_us() {
    long v = g_wait_spin_us.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("FRT_WAIT_SPIN_US");
        v = e ? std::max(0L, atol(e)) : 200L;
        g_wait_spin_us.store(v, std::memory_order_relaxed);
    }
    return v;
}
inline long wait_poll_us() {
    static const long v = [] {
        const char *e = getenv("FRT_WAIT_POLL_US");
        return e ? std::max(0L, atol(e)) : 2000000L;
    }();
    return v;
}
template <class Query>
inline bool spin_until_done(Query &&query) {
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
inline void wait_event_spinning(hipEvent_t ev) {
    if (!spin_until_done([&] { return hipEventQuery(ev); })) HIPCHK(hipEventSynchronize(ev));
}
inline void sync_stream_spinning(hipStream_t st) {
    if (!spin_until_done([&] { return hipStreamQuery(st); })) HIPCHK(hipStreamSynchronize(st));
}

// ------------------------------------------------------------------------------------------------ profiling (HIP events)
struct ProfRec {
    std::string name;
    hipEvent_t a, b;
    double work;
};
extern std::mutex g_prof_mu;  // (the profiling state is defined in frt_api.cpp)
extern int g_prof_kind;
extern std::vector<ProfRec> g_prof;
// Events are created when profiling is switched on, not between the two records of a bracket: the first hipEventCreate calls of a process
// take ~ 100 us each (pool set-up), and a host stall between "record a" and the launch it brackets is GPU idle time INSIDE the bracket
// (it showed up as conv_s2c64_kernel - the second bracket of a pass - at 224 us "live" against 97 us in rocprofv3's trace).
extern std::vector<hipEvent_t> g_prof_pool;
inline hipEvent_t prof_event() {
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


}  // namespace frti
using namespace frti;
