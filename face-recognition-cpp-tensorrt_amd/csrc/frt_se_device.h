// Device helpers of the SE tail (IR-SE), shared by the kernels that run it in their epilogue (kernels_arc.hip, kernels_arc_s2.hip).
#pragma once
#include "frt_kernels.h"

constexpr int SE_SPLIT = 4;  // pixel ranges per face (partial sums, summed in fixed order: deterministic)

// fc1 -> ReLU -> fc2 -> sigmoid on the pooled vector sp[C] (LDS), 256 threads.  Both layers are a few thousand MACs: what costs is
// the dependent chain, so every hidden unit gets 256 / R threads that each take a contiguous run of channels (all loads of a thread
// are independent 16-byte loads in flight at once), then a fixed-order shuffle reduction; the output layer is a thread per channel
// with its R weights as float4 loads.  (A wave per hidden unit walking the channels and a scalar loop over R: 5 - 11 us per call.)
__device__ __forceinline__ void se_fc1(const float *sp, float *shid, const float *__restrict__ w1, int C) {  // 256 threads; caller syncs afterwards
    const int R = C / 16;            // hidden units: 4 .. 32
    const int G = 256 / R;           // threads per hidden unit: 64 .. 8 (a power of two, inside one wave)
    const int per = C / G;           // channels per thread: 1, 4, 16, 64
    const int h = threadIdx.x / G, g = threadIdx.x % G;
    float a = 0.f;
    if (per == 1) {
        a = w1[(long)h * C + g] * sp[g];
    } else {
        const float *wp = w1 + (long)h * C + g * per;
        const float *xp = sp + g * per;
        for (int i = 0; i < per; i += 4) {
            const floatx4 w = *reinterpret_cast<const floatx4 *>(wp + i);
            a = fmaf(w[0], xp[i], a);
            a = fmaf(w[1], xp[i + 1], a);
            a = fmaf(w[2], xp[i + 2], a);
            a = fmaf(w[3], xp[i + 3], a);
        }
    }
    for (int off = G >> 1; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (g == 0) shid[h] = fmaxf(a, 0.f);
}
__device__ __forceinline__ float se_fc2(const float *shid, const float *__restrict__ w2, int C, int c) {  // gate of channel c
    const int R = C / 16;
    const float *wp = w2 + (long)c * R;
    float o = 0.f;
    for (int i = 0; i < R; i += 4) {
        const floatx4 w = *reinterpret_cast<const floatx4 *>(wp + i);
        o = fmaf(w[0], shid[i], o);
        o = fmaf(w[1], shid[i + 1], o);
        o = fmaf(w[2], shid[i + 2], o);
        o = fmaf(w[3], shid[i + 3], o);
    }
    return 1.f / (1.f + expf(-o));
}
