// RCCL through the C ABI from plain C++ (no Python, no torch): what a maintainer's multi-GPU shell binds (include/frt.h frt_comm_*).
// Runs on however many devices are visible: one communicator per device created in ONE process (frt_comm_create_all), an all-gather of
// per-device byte blocks issued from one thread (frt_comm_all_gather_multi) and, on device 0, a 1-rank communicator made from a unique id
// (frt_comm_get_unique_id + frt_comm_create) - the one-process-per-GPU form.  Prints "comm ok <ndev>".
// `comm_test stall`: the bound on RCCL's bootstrap - rank 0 of a TWO-rank communicator whose other rank never calls in must come back from
// frt_comm_create with an error after frt_comm_set_bootstrap_timeout seconds instead of hanging.  Prints "stall ok <seconds waited>".
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "frt.h"

#define CK(x)                                                                \
    do {                                                                     \
        if ((x) != 0) {                                                      \
            std::printf("FAILED %s: %s\n", #x, frt_last_error());            \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(int argc, char **argv) {
    const int ndev = frt_device_count();
    if (ndev < 1) return 2;
    if (argc > 1 && !std::strcmp(argv[1], "stall")) {
        frt_comm_set_bootstrap_timeout(4.0);
        uint8_t id[FRT_COMM_ID_BYTES];
        CK(frt_comm_get_unique_id(id));
        frt_comm *c = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = frt_comm_create(id, 0, 2, 0, &c);  // rank 1 never comes
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc == 0 || c || dt < 3.5 || dt > 60.0 || !std::strstr(frt_last_error(), "did not return within")) {
            std::printf("FAILED stall: rc %d after %.1f s: %s\n", rc, dt, frt_last_error());
            return 6;
        }
        std::printf("stall ok %.1f\n", dt);
        std::fflush(stdout);
        _exit(0);  // (the abandoned helper thread still sits in RCCL's bootstrap: no destructors, no atexit handlers)
    }
    const size_t B = 4096;
    // ---- one process per GPU form (here: world 1 on device 0)
    {
        uint8_t id[FRT_COMM_ID_BYTES];
        CK(frt_comm_get_unique_id(id));
        frt_comm *c = nullptr;
        CK(frt_comm_create(id, 0, 1, 0, &c));
        if (frt_comm_rank(c) != 0 || frt_comm_world(c) != 1 || !frt_comm_stream(c)) return 3;
        std::vector<uint8_t> h(B), back(B, 0);
        for (size_t i = 0; i < B; ++i) h[i] = (uint8_t)(i * 7 + 3);
        void *ds = nullptr, *dr = nullptr;
        hipSetDevice(0);
        hipMalloc(&ds, B);
        hipMalloc(&dr, B);
        hipMemcpy(ds, h.data(), B, hipMemcpyHostToDevice);
        CK(frt_comm_all_gather(c, ds, dr, B, nullptr));  // on the communicator's own stream
        CK(frt_comm_sync(c));
        hipMemcpy(back.data(), dr, B, hipMemcpyDeviceToHost);
        if (std::memcmp(h.data(), back.data(), B)) return 4;
        hipFree(ds);
        hipFree(dr);
        frt_comm_destroy(c);
    }
    // ---- one process, all visible devices
    {
        std::vector<int> devs;
        for (int d = 0; d < ndev; ++d) devs.push_back(d);
        std::vector<frt_comm *> comms((size_t)ndev, nullptr);
        CK(frt_comm_create_all(ndev, devs.data(), comms.data()));
        std::vector<void *> send((size_t)ndev), recv((size_t)ndev);
        for (int d = 0; d < ndev; ++d) {
            hipSetDevice(d);
            hipMalloc(&send[(size_t)d], B);
            hipMalloc(&recv[(size_t)d], B * ndev);
            hipMemset(send[(size_t)d], 0x10 + d, B);
            hipDeviceSynchronize();
        }
        CK(frt_comm_all_gather_multi(ndev, comms.data(), send.data(), recv.data(), B, nullptr));
        for (int d = 0; d < ndev; ++d) {
            CK(frt_comm_sync(comms[(size_t)d]));
            std::vector<uint8_t> h(B * ndev);
            hipSetDevice(d);
            hipMemcpy(h.data(), recv[(size_t)d], B * ndev, hipMemcpyDeviceToHost);
            for (int r = 0; r < ndev; ++r)
                for (size_t i = 0; i < B; ++i)
                    if (h[(size_t)r * B + i] != 0x10 + r) return 5;
        }
        for (int d = 0; d < ndev; ++d) {
            hipSetDevice(d);
            hipFree(send[(size_t)d]);
            hipFree(recv[(size_t)d]);
            frt_comm_destroy(comms[(size_t)d]);
        }
    }
    std::printf("comm ok %d\n", ndev);
    return 0;
}
