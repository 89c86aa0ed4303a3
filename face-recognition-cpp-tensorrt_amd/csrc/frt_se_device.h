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

// The IR-SE unit tail (model_irse.py:22-45, 58-66) inside conv2's epilogue: y = BN(conv2) * gate + shortcut, z = BN_next(y), where
// gate = sigmoid(fc2(relu(fc1(mean over the image of BN(conv2))))) needs the WHOLE image and all channels.  For the strip kernels
// (4 waves x 32 couts, NT pixel tiles per wave in MFMA accumulators, strips that are row ranges of one image or NI whole small images):
//   pass 1  transposes the accumulators once just to sum the (fp16-rounded, as the stand-alone path stores them) BN outputs per
//           channel and image; the partial sums go to pool[part of the image][face][channel] as device-scope stores;
//   meet    the workgroups of a face (parts x cout tiles; adjacent in launch order) count themselves in; the last one resets the
//           counter and publishes the launch number in the face's flag, everybody waits for it (bounded spin; on timeout the error word is raised);
//   gate    every workgroup adds the partial sums in part order - so the result does not depend on arrival order - and runs fc1 and
//           its 128 channels of fc2 itself (a few thousand MACs: cheaper than another hand-over);
//   pass 2  transposes again and writes y and z - exactly se_apply_kernel's arithmetic, without the res tensor's round trip,
//           the pooling pass and the apply pass (10 + 13 us and two dependent launches per unit before).
// No device-scope fence anywhere (see se_pool_gate_kernel); nothing a workgroup waits for depends on a workgroup that is dispatched
// more than (parts x cout tiles - 1) x 8 block indices later, so the wait cannot starve the launch.
// ep: this wave's [32][36] fp32 transpose tile; scratch: >= (576 + NI * 128) floats of LDS behind the four tiles; slot_pixel(slot, m, il):
// false for dead slots, else the flattened output pixel m and the strip-local image index il < n_img <= NI.
template <int NT, int NI, typename SlotPixel>
__device__ __forceinline__ void se_tail_epilogue(const ConvMfmaArgs &p, const floatx16 (&acc)[NT], float *ep, float *scratch, int part, int n_parts,
                                                 int n_co_tiles, int img0, int n_img, int HW, int co_base, int cow, SlotPixel slot_pixel) {
    constexpr int EROW = 36;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hi = lane >> 5, chunk = lane & 3;
    const int cch = co_base + cow + chunk * 8;
    const int C = p.Cout;
    floatx4 q0[2], q1[2];
    q0[0] = *reinterpret_cast<const floatx4 *>(p.p0 + cch);
    q0[1] = *reinterpret_cast<const floatx4 *>(p.p0 + cch + 4);
    q1[0] = *reinterpret_cast<const floatx4 *>(p.p1 + cch);
    q1[1] = *reinterpret_cast<const floatx4 *>(p.p1 + cch + 4);
    float se_sum[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) se_sum[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            *reinterpret_cast<floatx4 *>(ep + r * EROW + 8 * g + 4 * hi) = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int px = (lane >> 2) + 16 * it;
            const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8);
            const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8 + 4);
            long m;
            int il;
            if (!slot_pixel(j * 32 + px, m, il)) continue;
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (NI == 1 || il == i) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) se_sum[i][e] += (float)(half_t)(v[e] * q0[e >> 2][e & 3] + q1[e >> 2][e & 3]);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) se_sum[i][e] += __shfl_xor(se_sum[i][e], off);
    if ((lane >> 2) == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < n_img && img0 + i < p.B) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    __hip_atomic_store(&p.se_pool[((long)part * p.B + img0 + i) * C + cch + e], se_sum[i][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // acknowledged before the arrival is announced
    __syncthreads();
    if (tid < NI && tid < n_img && img0 + tid < p.B) {
        const int img = img0 + tid;
        const int expect = n_parts * n_co_tiles;
        int *flag = p.se_counter + p.se_flag_off;
        const int prev = __hip_atomic_fetch_add(&p.se_counter[img], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == expect - 1) {
            __hip_atomic_store(&p.se_counter[img], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&flag[img], p.se_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spin = 0;
            while (__hip_atomic_load(&flag[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.se_epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > (1 << 25)) {
                    // seconds: the hand-over is broken.  Raise the error word (mapped host memory: every synchronising entry point of
                    // the embedder / pipeline checks it and fails with FRT_ERR_DEVICE) and leave the loop; the gate of this face is
                    // then stale, but the HIP context - shared by every pipeline of the process - survives (a trap would kill it).
                    __hip_atomic_store(p.se_error, p.se_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
    }
    __syncthreads();
    // the first tile's shortcut values are requested here and land under the gate arithmetic
    auto load_sc = [&](int j, half8 (&dst)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            long m;
            int il;
            const bool ok = slot_pixel(j * 32 + (lane >> 2) + 16 * it, m, il);
            dst[it] = *reinterpret_cast<const half8 *>(p.sc + (ok ? m : 0) * C + cch);
        }
    };
    half8 scs[2][2];  // shortcut values one pixel tile ahead (all tiles at once: 56 live registers, spills)
    load_sc(0, scs[0]);
    float *sp = scratch, *shid = sp + 512, *sgate = sp + 576;  // sgate: [NI][128]
    for (int i = 0; i < NI; ++i) {
        if (i < n_img && img0 + i < p.B) {
            for (int c = tid; c < C; c += 256) {
                float t = 0.f;
                for (int q = 0; q < n_parts; ++q)
                    t += __hip_atomic_load(&p.se_pool[((long)q * p.B + img0 + i) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sp[c] = t / (float)HW;
            }
        }
        __syncthreads();
        se_fc1(sp, shid, p.se_w1, C);
        __syncthreads();
        if (tid < 128) sgate[i * 128 + tid] = se_fc2(shid, p.se_w2, C, co_base + tid);
        __syncthreads();
    }
    floatx4 q2[2], q3[2];
    q2[0] = *reinterpret_cast<const floatx4 *>(p.p2 + cch);
    q2[1] = *reinterpret_cast<const floatx4 *>(p.p2 + cch + 4);
    q3[0] = *reinterpret_cast<const floatx4 *>(p.p3 + cch);
    q3[1] = *reinterpret_cast<const floatx4 *>(p.p3 + cch + 4);
    float g8[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) g8[i][e] = sgate[i * 128 + cow + chunk * 8 + e];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j + 1 < NT) load_sc(j + 1, scs[(j + 1) & 1]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            *reinterpret_cast<floatx4 *>(ep + r * EROW + 8 * g + 4 * hi) = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int px = (lane >> 2) + 16 * it;
            const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8);
            const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8 + 4);
            long m;
            int il;
            if (!slot_pixel(j * 32 + px, m, il)) continue;
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            half8 y8, z8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float res = (float)(half_t)(v[e] * q0[e >> 2][e & 3] + q1[e >> 2][e & 3]);
                const float gte = NI == 1 ? g8[0][e] : (il == 0 ? g8[0][e] : g8[NI - 1][e]);
                const float y = res * gte + (float)scs[j & 1][it][e];
                y8[e] = (half_t)y;
                z8[e] = (half_t)(y * q2[e >> 2][e & 3] + q3[e >> 2][e & 3]);
            }
            *reinterpret_cast<half8 *>(p.out0 + m * C + cch) = y8;
            *reinterpret_cast<half8 *>(p.out1 + m * C + cch) = z8;
        }
    }
}
