// Multi-GPU exchange step of the hot path from C++: RCCL (librccl, the ROCm collectives library over xGMI) behind the C ABI.
//
// The reference is ONE single-GPU C++ server process (src/app.cpp:52-57, 367); north_star shards whole frames over the 8 GPUs of a node
// with "RCCL all-gather of embeddings" as the only exchange.  A maintainer's C++ shell reaches the other GPUs through these entry points -
// no Python, no torch.distributed: one communicator per device (one process per GPU, or one process driving several devices from
// threads), ncclAllGather of byte blocks (result records, fp16 embeddings, top-k winners) on a stream.
//
// librccl is bound at run time (dlopen "librccl.so.1"): a single-GPU deployment does not need it installed, and a process that already
// carries one (PyTorch bundles its own) shares that copy instead of loading a second.  Types come from <rccl/rccl.h>.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "frt_host.hpp"

using frthost::guarded;
using frthost::raise;
using frthost::use_device;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char *e = dlerror();  // (one call: dlerror() clears the state it returns)
            r.error = std::string("librccl not found: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!r.error.empty()) raise(FRT_ERR_DEVICE, r.error);
    return r;
}

void ncclchk(ncclResult_t e, const char *what) {
    if (e != ncclSuccess) raise(FRT_ERR_DEVICE, std::string("RCCL: ") + what + ": " + rccl().GetErrorString(e));
}

// ---- bounded bootstrap.  ncclGetUniqueId (opens the bootstrap's listening socket), ncclCommInitRank (every rank connects to it and to
// its ring neighbours) and ncclCommInitAll are BLOCKING calls with no timeout of their own: a rank that never arrives, an interface RCCL
// picked that the peers cannot reach, or a socket stall leaves the caller hung - on an 8-GPU node that is a hung job, not an error.  Each
// of them runs on a helper thread here; the caller waits for it up to frt_comm_set_bootstrap_timeout seconds (default 180; env
// FRT_COMM_BOOTSTRAP_TIMEOUT_S; <= 0: forever) and gets FRT_ERR_DEVICE with a message when the time is up.  Everything the helper touches
// lives in a shared heap record, so a helper that comes back late finds valid memory, sees that it was given up on and tears down what it
// made (ncclCommAbort).  (RCCL's own non-blocking initialisation - ncclCommInitRankConfig with blocking = 0 - would make every later
// collective on the communicator non-blocking too; the exchange step wants ordinary stream-ordered calls.)
std::atomic<long> g_bootstrap_timeout_ms{-1};
long bootstrap_timeout_ms() {
    long v = g_bootstrap_timeout_ms.load(std::memory_order_relaxed);
    if (v != -1) return v;
    const char *e = getenv("FRT_COMM_BOOTSTRAP_TIMEOUT_S");
    v = e ? (long)(atof(e) * 1000.0) : 180000L;
    g_bootstrap_timeout_ms.store(v, std::memory_order_relaxed);
    return v;
}
struct Bootstrap {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclResult_t res = ncclSuccess;
    ncclUniqueId uid;
    std::vector<ncclComm_t> comms;
    std::vector<int> devices;
};
// runs body(record) on a helper thread; true = finished in time (result in record->res)
bool bounded(const std::shared_ptr<Bootstrap> &b, std::function<ncclResult_t(Bootstrap &)> body, std::function<void(Bootstrap &)> undo) {
    std::thread([b, body, undo] {
        const ncclResult_t r = body(*b);
        std::unique_lock<std::mutex> lk(b->mu);
        b->res = r;
        b->done = true;
        if (b->abandoned) {
            lk.unlock();
            if (r == ncclSuccess && undo) undo(*b);  // nobody is waiting for what this call made any more
            return;
        }
        b->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(b->mu);
    const long ms = bootstrap_timeout_ms();
    if (ms <= 0) b->cv.wait(lk, [&] { return b->done; });
    else if (!b->cv.wait_for(lk, std::chrono::milliseconds(ms), [&] { return b->done; })) b->abandoned = true;
    return b->done;
}
[[noreturn]] void stalled(const char *what) {
    char buf[640];
    snprintf(buf, sizeof(buf),
             "RCCL: %s did not return within %.0f s (frt_comm_set_bootstrap_timeout / FRT_COMM_BOOTSTRAP_TIMEOUT_S): the bootstrap is stalled - a rank "
             "of the communicator never called in, or RCCL chose a network interface its peers cannot reach (single node: NCCL_SOCKET_IFNAME=lo; "
             "NCCL_DEBUG=INFO shows the interface).  The call was given up on; its helper thread tears the communicator down if it ever returns",
             what, bootstrap_timeout_ms() / 1000.0);
    raise(FRT_ERR_DEVICE, buf);
}

}  // namespace

struct frt_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    // The exchange runs on a stream of its own so that it never sits in front of a pipeline stage: created HERE, i.e. after the caller's
    // pipelines exist (create the communicator last: a stream created before a pipeline's stage streams can change how ROCm maps those onto
    // hardware queues).
    hipStream_t side = nullptr;
    std::mutex mu;
};

extern "C" {

double frt_comm_set_bootstrap_timeout(double seconds) {
    const double prev = bootstrap_timeout_ms() / 1000.0;
    g_bootstrap_timeout_ms.store(seconds > 0 ? (long)(seconds * 1000.0 + 0.5) : 0L, std::memory_order_relaxed);
    return prev;
}

int frt_comm_get_unique_id(uint8_t *id_out) {
    return guarded([&] {
        if (!id_out) raise(FRT_ERR_INVALID, "null argument");
        static_assert(sizeof(ncclUniqueId) == FRT_COMM_ID_BYTES, "FRT_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
        Rccl &r = rccl();
        auto b = std::make_shared<Bootstrap>();
        if (!bounded(b, [&r](Bootstrap &x) { return r.GetUniqueId(&x.uid); }, nullptr)) stalled("ncclGetUniqueId");
        ncclchk(b->res, "ncclGetUniqueId");
        std::memcpy(id_out, &b->uid, sizeof(b->uid));
    });
}

int frt_comm_create(const uint8_t *id, int rank, int world, int device, frt_comm **out) {
    return guarded([&] {
        if (!id || !out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (world < 1 || rank < 0 || rank >= world) raise(FRT_ERR_INVALID, "comm: bad rank / world size");
        use_device(device);
        std::unique_ptr<frt_comm> c(new frt_comm);
        c->device = device;
        c->rank = rank;
        c->world = world;
        Rccl &r = rccl();
        auto b = std::make_shared<Bootstrap>();
        std::memcpy(&b->uid, id, sizeof(b->uid));
        b->comms.assign(1, nullptr);
        if (!bounded(
                b,
                [&r, world, rank, device](Bootstrap &x) {
                    if (hipSetDevice(device) != hipSuccess) return ncclUnhandledCudaError;  // (the helper thread has no current device yet)
                    return r.CommInitRank(&x.comms[0], world, x.uid, rank);
                },
                [&r](Bootstrap &x) { (void)r.CommAbort(x.comms[0]); }))
            stalled("ncclCommInitRank");
        ncclchk(b->res, "ncclCommInitRank");
        c->comm = b->comms[0];
        HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        *out = c.release();
    });
}

int frt_comm_create_all(int n_devices, const int *devices, frt_comm **out) {
    return guarded([&] {
        if (!devices || !out || n_devices < 1) raise(FRT_ERR_INVALID, "comm: bad argument");
        for (int i = 0; i < n_devices; ++i) out[i] = nullptr;
        Rccl &r = rccl();
        auto b = std::make_shared<Bootstrap>();
        b->comms.assign((size_t)n_devices, nullptr);
        b->devices.assign(devices, devices + n_devices);
        if (!bounded(
                b, [&r](Bootstrap &x) { return r.CommInitAll(x.comms.data(), (int)x.devices.size(), x.devices.data()); },
                [&r](Bootstrap &x) {
                    for (ncclComm_t cm : x.comms)
                        if (cm) (void)r.CommAbort(cm);
                }))
            stalled("ncclCommInitAll");
        ncclchk(b->res, "ncclCommInitAll");
        const std::vector<ncclComm_t> &comms = b->comms;
        for (int i = 0; i < n_devices; ++i) {
            frt_comm *c = new frt_comm;
            c->device = devices[i];
            c->rank = i;
            c->world = n_devices;
            c->comm = comms[(size_t)i];
            out[i] = c;
            use_device(devices[i]);
            HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        }
    });
}

void frt_comm_destroy(frt_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    if (c->comm) {
        try {
            (void)rccl().CommDestroy(c->comm);
        } catch (...) {
        }
    }
    delete c;
}

int frt_comm_rank(const frt_comm *c) { return c ? c->rank : -1; }
int frt_comm_world(const frt_comm *c) { return c ? c->world : 0; }
void *frt_comm_stream(frt_comm *c) { return c ? reinterpret_cast<void *>(c->side) : nullptr; }

int frt_comm_all_gather(frt_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *hip_stream) {
    return guarded([&] {
        if (!c || !send_dev || !recv_dev) raise(FRT_ERR_INVALID, "all_gather: null argument");
        if (bytes_per_rank == 0) return;
        std::lock_guard<std::mutex> lk(c->mu);
        use_device(c->device);
        hipStream_t st = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->side;
        ncclchk(rccl().AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, c->comm, st), "ncclAllGather");
    });
}

int frt_comm_all_gather_multi(int n, frt_comm *const *comms, const void *const *send_dev, void *const *recv_dev, size_t bytes_per_rank,
                              void *const *hip_streams) {
    return guarded([&] {
        if (n < 1 || !comms || !send_dev || !recv_dev) raise(FRT_ERR_INVALID, "all_gather_multi: bad argument");
        if (bytes_per_rank == 0) return;
        Rccl &r = rccl();
        ncclchk(r.GroupStart(), "ncclGroupStart");
        ncclResult_t first = ncclSuccess;
        for (int i = 0; i < n; ++i) {
            frt_comm *c = comms[i];
            if (!c || !send_dev[i] || !recv_dev[i]) {
                first = ncclInvalidArgument;
                break;
            }
            (void)hipSetDevice(c->device);
            hipStream_t st = hip_streams && hip_streams[i] ? reinterpret_cast<hipStream_t>(hip_streams[i]) : c->side;
            const ncclResult_t e = r.AllGather(send_dev[i], recv_dev[i], bytes_per_rank, ncclUint8, c->comm, st);
            if (e != ncclSuccess && first == ncclSuccess) first = e;
        }
        const ncclResult_t ge = r.GroupEnd();
        ncclchk(first, "ncclAllGather (group)");
        ncclchk(ge, "ncclGroupEnd");
    });
}

int frt_comm_sync(frt_comm *c) {
    return guarded([&] {
        if (!c) raise(FRT_ERR_INVALID, "null argument");
        use_device(c->device);
        HIPCHK(hipStreamSynchronize(c->side));
    });
}

}  // extern "C"
