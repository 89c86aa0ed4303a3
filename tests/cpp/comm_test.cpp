// RCCL through the C ABI from plain C++ (no Python, no torch): what a maintainer's multi-GPU shell binds (include/frt.h frt_comm_*).
// Runs on however many devices are visible: one communicator per device created in ONE process (frt_comm_create_all), an all-gather of
// per-device byte blocks issued from one thread (frt_comm_all_gather_multi) and, on device 0, a 1-rank communicator made from a unique id
// (frt_comm_get_unique_id + frt_comm_create) - the one-process-per-GPU form.  Prints "comm ok <ndev>".
#include <hip/hip_runtime.h>

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

int main() {
    const int ndev = frt_device_count();
    if (ndev < 1) return 2;
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
