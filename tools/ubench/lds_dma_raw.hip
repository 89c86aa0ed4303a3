// Does a ds_read issued right behind "s_waitcnt vmcnt(0)" see the bytes of the LDS-DMA (global_load_lds_dwordx4) the wait covers?
// Round-4 observation in dwpw_wave_kernel: once, in one build, it did not (kernels_det_wave.hip).  This probe hammers exactly that pattern:
// every wave (one-wave workgroups, many per CU, all CUs busy) repeatedly DMAs 8 KB of a pattern that changes per iteration into the SAME LDS
// bytes, waits with vmcnt(0), optionally fences, and reads words other lanes' DMA slots wrote with ds_read2_b32; any word that is not the current
// iteration's pattern is a stale (or torn) read.  Variants: 0 nothing between wait and read, 1 s_nop 0, 2 s_barrier, 3 s_sleep 2,
// 4 counted wait vmcnt(8) with 8 younger DMAs to another region in flight (the kernel's real shape), 5 = 4 + s_barrier.
//   hipcc -O3 --offload-arch=gfx950 -o lds_dma_raw lds_dma_raw.hip && ./lds_dma_raw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int PIECES = 8;        // 1 KB each
constexpr int NPAT = 8;          // distinct source patterns
template <int V>
__global__ __launch_bounds__(64) void probe(const unsigned *src, unsigned long long *bad, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * PIECES * 256];
    const int lane = threadIdx.x;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)lds;
    // zero once
    for (int i = lane; i < 2 * PIECES * 256; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    unsigned long long nbad = 0;
    const unsigned voff = lane * 16;
    for (int it = 0; it < iters; ++it) {
        const unsigned *s = src + (size_t)((it + blockIdx.x) % NPAT) * (PIECES * 256);
        const unsigned want_base = (unsigned)((it + blockIdx.x) % NPAT) << 24;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const unsigned *sp = s + p * 256;
            const unsigned dst = lds0 + p * 1024;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sp), "s"(dst) : "memory");
        }
        if (V >= 4) {  // younger DMAs into the second region (never read): the wait below is a counted one
            const unsigned *s2 = src + (size_t)((it + 3) % NPAT) * (PIECES * 256);
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const unsigned *sp = s2 + p * 256;
                const unsigned dst = lds0 + (PIECES + p) * 1024;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sp), "s"(dst) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (V == 1) asm volatile("s_nop 0" ::: "memory");
        if (V == 2 || V == 5) asm volatile("s_barrier" ::: "memory");
        if (V == 3) asm volatile("s_sleep 2" ::: "memory");
        // read words written by other lanes' slots: piece p, lane' = 63 - lane, dwords 0 and 3 of its 16 bytes (+ the last piece first)
        unsigned got[PIECES][2];
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const int pp = PIECES - 1 - p;
            const unsigned addr = lds0 + pp * 1024 + (63 - lane) * 16;
            unsigned long long v;
            asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:3" : "=v"(v) : "v"(addr) : "memory");
            got[pp][0] = (unsigned)v;   // (filled after the wait below)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
            got[pp][0] = (unsigned)v;
            got[pp][1] = (unsigned)(v >> 32);
        }
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const unsigned w0 = want_base | (unsigned)(p * 256 + (63 - lane) * 4), w3 = w0 + 3;
            if (got[p][0] != w0) ++nbad;
            if (got[p][1] != w3) ++nbad;
        }
        if (V >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (WAR: the next iteration's DMA overwrites bytes whose reads have completed - lgkmcnt(0) above)
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    std::vector<unsigned> h((size_t)NPAT * PIECES * 256);
    for (int k = 0; k < NPAT; ++k)
        for (int i = 0; i < PIECES * 256; ++i) h[(size_t)k * PIECES * 256 + i] = ((unsigned)k << 24) | (unsigned)i;
    unsigned *d; unsigned long long *bad;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&bad, 8));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 2000, grid = 256 * 16;
    auto run = [&](auto kern, const char *name) {
        CK(hipMemset(bad, 0, 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, bad, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long b; CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
        printf("%-44s stale/torn words: %llu of %.3g checked   (%.1f ms)\n", name, b, 5.0 * grid * 64 * iters * PIECES * 2, ms);
    };
    run(probe<0>, "vmcnt(0) ; ds_read");
    run(probe<1>, "vmcnt(0) ; s_nop 0 ; ds_read");
    run(probe<2>, "vmcnt(0) ; s_barrier ; ds_read");
    run(probe<3>, "vmcnt(0) ; s_sleep 2 ; ds_read");
    run(probe<4>, "vmcnt(8) (8 younger DMAs) ; ds_read");
    run(probe<5>, "vmcnt(8) ; s_barrier ; ds_read");
    return 0;
}
