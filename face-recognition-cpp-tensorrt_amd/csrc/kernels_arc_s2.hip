// ArcFace IR-50: the stride-2 3x3 convolutions (second conv of the first unit of every stage, model_irse.py:48-65 with stride 2:
// 128 -> 128 at 56 -> 28, 256 -> 256 at 28 -> 14, 512 -> 512 at 14 -> 7), fp16 NHWC, fp32 accumulation on v_mfma_f32_32x32x16_f16.
//
// They ran on the im2col LDS-DMA kernel (kernels_arc.hip: conv_glds_kernel), which moves every input pixel L2 -> LDS once per tap that
// touches it and measured 58 - 83 us per layer against 37 - 42 us for the LDS-resident strip kernel on stride-1 layers with the same
// 29.6 GFLOP.  This is the strip kernel's idea carried over to stride 2.
//
// A strip kernel keeps the strip's input patch resident in LDS and makes a tap a constant address offset: slot s -> patch row s + off.
// With stride 2 consecutive output pixels are TWO input pixels apart (and LDS rows 2 * 144 B apart collide pairwise in the 64 banks),
// and the patch of R output rows is 4x larger than at stride 1.  Both problems go away when the patch is staged de-interleaved by the
// parity of the input row and column - four PHASE PLANES, each on the OUTPUT grid ((R+1) x (Wo+1) pixels, pixel pitch 144 B):
//     input row 2*oy + kh - 1:   kh = 0 -> odd rows,  plane row oy        kh = 1 -> even rows, plane row oy      kh = 2 -> odd rows, plane row oy + 1
// (same for columns), so inside a plane a tap is again "slot + constant" and consecutive slots are consecutive 144-byte rows
// (conflict-free ds_read_b128).  The nine taps split over the planes as 4 (odd, odd) + 1 (even, even) + 2 (odd, even) + 2 (even, odd);
// the K loop walks, per 64-channel chunk, plane by plane in that order.  Each plane has its own LDS buffer (4 x <= 36 KB): the moment
// every wave has finished a plane (one workgroup barrier per plane), the SAME plane of the next channel chunk is fetched into it by
// LDS-DMA - five or more (chunk, tap) steps before it is needed.  VMEM operations retire in order, so the wait in front of a plane is
// an exact compile-time count: two younger plane fetches plus four weight-fragment loads per step issued since.
// Weights: fragment-ordered copy in THIS kernel's step order (host-packed, [32-cout block][chunk][step][kk][lane][8 halfs]); each
// wave streams its 32 couts' fragments straight into registers, two steps ahead, one contiguous kilobyte per load.
// One workgroup (4 waves x 32 couts, NT pixel tiles each) per CU: the four plane buffers take up to 144 KB of the 160 KB LDS.
#include "frt_kernels.h"
#include "frt_se_device.h"

#include <algorithm>
#include <cstdio>
#include <vector>

#include <type_traits>

namespace {

constexpr int PROW = 144;  // bytes per plane pixel (128 data + 16 pad)

// step -> (plane, row offset, column offset); plane order (odd,odd), (even,even), (odd,even), (even,odd)
__device__ constexpr int kSub[9] = {0, 0, 0, 0, 1, 2, 2, 3, 3};
__device__ constexpr int kDr[9] = {0, 0, 1, 1, 0, 0, 1, 0, 0};
__device__ constexpr int kDc[9] = {0, 1, 0, 1, 0, 0, 0, 0, 1};
__device__ constexpr int kLen[4] = {4, 1, 2, 2};   // steps per plane
__device__ constexpr int kRowPar[4] = {1, 0, 1, 0};  // plane -> input row parity (1: odd rows 2*i - 1, 0: even rows 2*i)
__device__ constexpr int kColPar[4] = {1, 0, 0, 1};

// SCF: the unit's 1x1 stride-2 shortcut convolution (model_irse.py:52-54: Conv2d(in, depth, (1, 1), stride) + BatchNorm2d on the unit's RAW
// input) computed HERE instead of by a launch of its own: its input pixels x[2 oy][2 ox] are exactly the (even, even) phase plane, so
// when the last chunk of the 3x3 loop releases the four plane buffers they are refilled with the (even, even) planes of the shortcut's
// (up to four) 64-channel chunks - the DMAs that used to fetch zeros at the tail - and 4 MFMAs per chunk and pixel tile into a second
// accumulator set follow the main loop.  The shortcut tensor's HBM round trip (write + read, fp16-rounded) and a launch disappear.
template <int NT, int PP, bool SEP = false, bool SCF = false>  // NT pixel tiles per strip; PP = LDS-DMA pieces (1 KB per wave) per thread per plane; SEP: IR-SE tail in the epilogue
__global__ __launch_bounds__(256, 1) void conv_s2_kernel(ConvMfmaArgs p, int R, int n_img, int linear) {
    constexpr int PLANE_B = PP * 4096;
    constexpr int LA = 2, WR = 3, BFD = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo, Wq = Wo + 1;
    const int NPp = n_img * (R + 1) * Wq;  // pixels of one plane
    const int strips_per_img = Ho / R;
    const int n_valid = n_img * R * Wo;

#ifdef FRT_ABLATE
    // timing build (FRT_S2_STAMPS=1): phase stamps (100 MHz constant clock) of wave 0 of the first and the last workgroup into p.outf
    unsigned long long *stamps = (p.outf && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
                                     ? reinterpret_cast<unsigned long long *>(p.outf) + (blockIdx.x ? 8 : 0) : nullptr;
#define S2_STAMP(i) do { if (stamps && lane == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define S2_STAMP(i) do { } while (0)
#endif
    S2_STAMP(0);
    const int n_co_tiles = p.Cout >> 7;
    const int nblk = gridDim.x, bq = nblk >> 3, brem = nblk & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
    const int co_tile = lid % n_co_tiles, strip = lid / n_co_tiles;
    const int co_base = co_tile * 128, cow = wave * 32;
    const int img0 = (strip / strips_per_img) * n_img;
    const int oy0 = (strip % strips_per_img) * R;
    const int n_chunks = p.Cin >> 6;

    // ---- plane DMA descriptors: piece q of plane s covers 16-byte chunk g = (q*4 + wave)*64 + lane of the plane image.
    //      A piece's pixel (image, plane row, plane column) is the same in all four planes and in the shortcut plane - it is decomposed ONCE per
    //      piece, with reciprocal multiplies + one correction step instead of integer divisions by run-time values (round 5: phase stamps showed
    //      11.8 - 15 us of a 35 - 48 us launch in front of the first DMA: 36 descriptors x 3 divisions per thread; FRT_S2_STAMPS, tuning build)
    int poff[4][PP];
    const int n_sc = SCF ? (p.Csc >> 6) : 0;  // 64-channel chunks of the shortcut conv: 1, 2 or 4 (<= the four plane buffers)
    int poff_sc[SCF ? PP : 1];
    {
        const int plane_px = (R + 1) * Wq;
        const float inv_plane = 1.0f / (float)plane_px, inv_row = 1.0f / (float)Wq;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const int g = (q * 4 + wave) * 64 + lane;
            const int px = g / 9, pos = g - px * 9;
            int il = (int)(((float)px + 0.5f) * inv_plane);
            int rem = px - il * plane_px;
            if (rem < 0) { --il; rem += plane_px; } else if (rem >= plane_px) { ++il; rem -= plane_px; }
            int i = (int)(((float)rem + 0.5f) * inv_row);
            int j = rem - i * Wq;
            if (j < 0) { --i; j += Wq; } else if (j >= Wq) { ++i; j -= Wq; }
            const int b = img0 + il;
            const bool live = pos < 8 && px < NPp && b < p.B;
            const int ye = 2 * (oy0 + i), xe = 2 * j;  // the (even, even) input pixel of this plane pixel
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int y = ye - kRowPar[s], x = xe - kColPar[s];
                poff[s][q] = (live && y >= 0 && y < H && x >= 0 && x < W) ? ((b * H + y) * W + x) * p.Cin + pos * 8 : -1;
            }
            if constexpr (SCF) poff_sc[q] = (live && ye < H && xe < W) ? ((b * H + ye) * W + xe) * p.Csc + pos * 8 : -1;
        }
    }
    const half_t *wfrag = p.wf + ((long)((co_base + cow) >> 5) * n_chunks) * (9 * 4 * 512) + lane * 8;
    int pbase[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int sl = j * 32 + r;
        int pidx = 0;
        if (linear) {
            pidx = sl < R * Wq ? sl : 0;
        } else if (sl < n_valid) {
            const int per = R * Wo;
            int il = (int)(((float)sl + 0.5f) * (1.0f / (float)per));
            int rem = sl - il * per;
            if (rem < 0) { --il; rem += per; } else if (rem >= per) { ++il; rem -= per; }
            int rr = (int)(((float)rem + 0.5f) * (1.0f / (float)Wo));
            int cc = rem - rr * Wo;
            if (cc < 0) { --rr; cc += Wo; } else if (cc >= Wo) { ++rr; cc -= Wo; }
            pidx = (il * (R + 1) + rr) * Wq + cc;
        }
        pbase[j] = pidx * PROW + hi * 16;
    }

    half8 areg[WR][4];
    auto load_w = [&](int c, int step, auto slot_c) {  // wave-uniform c, step; clamped at the tail (values unused there)
        constexpr int S = decltype(slot_c)::value;
        const int woff = c < n_chunks ? (c * 9 + step) * (4 * 512) : 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) areg[S][kk] = *reinterpret_cast<const half8 *>(wfrag + woff + kk * 512);
    };
    auto issue_plane = [&](int c, auto sub_c) {
        constexpr int S = decltype(sub_c)::value;
        const bool real = c < n_chunks;
        char *pl = smem + S * PLANE_B + wave * 1024;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const half_t *src = (real && poff[S][q] >= 0) ? p.x + (unsigned)(poff[S][q] + (c << 6)) : p.zeros;
            if constexpr (SCF) {  // behind the last chunk: buffer S takes the (even, even) plane of shortcut chunk S
                if (!real && S < n_sc && poff_sc[q] >= 0) src = p.scx + (unsigned)(poff_sc[q] + (S << 6));
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(pl + q * 4096), 16, 0, 0);
        }
    };

    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // prologue: all four planes of chunk 0, then the weight fragments of steps 0 and 1
    issue_plane(0, std::integral_constant<int, 0>{});
    issue_plane(0, std::integral_constant<int, 1>{});
    issue_plane(0, std::integral_constant<int, 2>{});
    issue_plane(0, std::integral_constant<int, 3>{});
    load_w(0, 0, std::integral_constant<int, 0>{});
    load_w(0, 1, std::integral_constant<int, 1>{});
    half8 bf[BFD][NT];
    S2_STAMP(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LA) : "memory");  // this wave's plane pieces have landed (younger: the fragment loads)
    __builtin_amdgcn_s_barrier();                                    // ... and everybody else's
    S2_STAMP(2);
#pragma unroll
    for (int k2 = 0; k2 < BFD; ++k2)
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[k2][j] = *reinterpret_cast<const half8 *>(smem + pbase[j] + k2 * 32);  // step 0: plane 0, offsets (0, 0)

    auto step = [&](int c, auto step_c) {
        constexpr int ST = decltype(step_c)::value;
        constexpr int NS = (ST + 1) % 9;
        constexpr int AS = ST % WR;
        constexpr bool LAST_OF_PLANE = kSub[NS] != kSub[ST];
        const int dp = kSub[ST] * PLANE_B + (kDr[ST] * Wq + kDc[ST]) * PROW;
        const int dpn = kSub[NS] * PLANE_B + (kDr[NS] * Wq + kDc[NS]) * PROW;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1;
            if (kk == 4 - BFD && LAST_OF_PLANE) {
                // plane hand-over: every read of this plane's buffer has been issued (the refills from here on read the NEXT plane);
                // wait for them, for this wave's pieces of the next plane (issued one chunk ago: since then two younger plane fetches
                // and four fragment loads per step), meet, then refill THIS plane's buffer with the next chunk's data.
                constexpr int NEXT = kSub[NS];
                constexpr int YOUNGER = 4 * (9 - kLen[NEXT]) + 2 * PP;
                static_assert(YOUNGER <= 63, "vmcnt field");
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                issue_plane(c + 1, std::integral_constant<int, kSub[ST]>{});
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[AS][kk], bf[cur][j], acc[j], 0, 0, 0);
                if (kk + BFD < 4) bf[cur][j] = *reinterpret_cast<const half8 *>(smem + pbase[j] + dp + (kk + BFD) * 32);
                else bf[cur][j] = *reinterpret_cast<const half8 *>(smem + pbase[j] + dpn + (kk + BFD - 4) * 32);
                if (kk == 0 && j == (NT > 1 ? 1 : 0)) {  // weight fragments of step t+2 into the ring slot step t-1 used
                    constexpr int T2 = ST + LA;
                    load_w(T2 < 9 ? c : c + 1, T2 % 9, std::integral_constant<int, (T2 % 9) % WR>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int c = 0; c < n_chunks; ++c) {
        step(c, std::integral_constant<int, 0>{});
        step(c, std::integral_constant<int, 1>{});
        step(c, std::integral_constant<int, 2>{});
        step(c, std::integral_constant<int, 3>{});
        step(c, std::integral_constant<int, 4>{});
        step(c, std::integral_constant<int, 5>{});
        step(c, std::integral_constant<int, 6>{});
        step(c, std::integral_constant<int, 7>{});
        step(c, std::integral_constant<int, 8>{});
    }
    S2_STAMP(3);
    floatx16 acc_sc[SCF ? NT : 1];
    if constexpr (SCF) {  // this wave's shortcut weight fragments: [32-cout block][chunk][kk][lane][8 halfs], requested under the tail DMAs
        half8 wsc[4][4];
        const half_t *wsf = p.wscf + ((long)((co_base + cow) >> 5) * n_sc) * (4 * 512) + lane * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wsc[c][kk] = *reinterpret_cast<const half8 *>(wsf + ((c < n_sc ? c : 0) * 4 + kk) * 512);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_sc[j][e] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the shortcut planes (the tail's DMAs) and the fragments have landed
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < n_sc) {  // (wave-uniform)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const half8 b = *reinterpret_cast<const half8 *>(smem + pbase[j] + c * PLANE_B + kk * 32);
                        acc_sc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wsc[c][kk], b, acc_sc[j], 0, 0, 0);
                    }
            }
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's dummy DMAs still target LDS
    }
    __syncthreads();
    S2_STAMP(4);

    // ------------------------------------------------------------------ epilogue (per wave: 32 couts x NT pixel tiles) through LDS
    constexpr int EROW = 36;  // floats per pixel row (32 + 4 pad)
    float *ep = reinterpret_cast<float *>(smem) + wave * (32 * EROW);
    const int chunk = lane & 3;
    const int cch = co_base + cow + chunk * 8;
    floatx4 q0[2], q1[2], q2[2], q3[2];
    q0[0] = *reinterpret_cast<const floatx4 *>(p.p0 + cch);
    q0[1] = *reinterpret_cast<const floatx4 *>(p.p0 + cch + 4);
    q1[0] = *reinterpret_cast<const floatx4 *>(p.p1 + cch);
    q1[1] = *reinterpret_cast<const floatx4 *>(p.p1 + cch + 4);
    const bool two = p.mode == EPI_BN_ADD_BN && p.out1;
    if (two) {
        q2[0] = *reinterpret_cast<const floatx4 *>(p.p2 + cch);
        q2[1] = *reinterpret_cast<const floatx4 *>(p.p2 + cch + 4);
        q3[0] = *reinterpret_cast<const floatx4 *>(p.p3 + cch);
        q3[1] = *reinterpret_cast<const floatx4 *>(p.p3 + cch + 4);
    }
    const long Mtot = (long)p.B * Ho * Wo;
    const float inv_wq = 1.0f / (float)Wq;
    auto slot_pixel = [&](int sl, long &m) -> bool {  // pixel slot -> flattened output pixel index; false for dead slots
        if (linear) {
            const int rr = (int)(((float)sl + 0.5f) * inv_wq);  // exact for sl < 2^20
            const int cc = sl - rr * Wq;
            m = ((long)img0 * Ho + oy0 + rr) * Wo + cc;
            return rr < R && cc < Wo && m < Mtot;
        }
        // compact enumeration: strips of whole images (R == Ho), contiguous in the flattened (image, row, column) index
        m = ((long)img0 * Ho + oy0) * Wo + sl;
        return sl < n_valid && m < Mtot;
    };
    if constexpr (SEP) {  // IR-SE: the whole SE tail here (frt_se_device.h)
        const int per_img = R * Wo;
        se_tail_epilogue<NT, (NT == 4 ? 2 : 1)>(p, acc, ep, reinterpret_cast<float *>(smem + 4 * 32 * EROW * 4), strip % strips_per_img, strips_per_img,
                                               n_co_tiles, img0, n_img, Ho * Wo, co_base, cow, [&](int sl, long &m, int &il) -> bool {
                                                   il = (NT == 4 && !linear && sl >= per_img) ? 1 : 0;
                                                   return slot_pixel(sl, m);
                                               });
        return;
    }
    half8 sc8[SCF ? 1 : NT][2];
    floatx4 qs0[2], qs1[2];
    if constexpr (SCF) {  // the shortcut's folded BatchNorm
        qs0[0] = *reinterpret_cast<const floatx4 *>(p.psc0 + cch);
        qs0[1] = *reinterpret_cast<const floatx4 *>(p.psc0 + cch + 4);
        qs1[0] = *reinterpret_cast<const floatx4 *>(p.psc1 + cch);
        qs1[1] = *reinterpret_cast<const floatx4 *>(p.psc1 + cch + 4);
    } else if (p.mode == EPI_BN_ADD_BN) {  // the shortcut has the output's geometry; all loads in flight before the transposes
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int sl = j * 32 + (lane >> 2) + 16 * it;
                long m;
                const bool ok = slot_pixel(sl, m);
                sc8[j][it] = *reinterpret_cast<const half8 *>(p.sc + (ok ? m : 0) * p.Cout + cch);
            }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        float v[2][8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 t = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            *reinterpret_cast<floatx4 *>(ep + r * EROW + 8 * g + 4 * hi) = t;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int px = (lane >> 2) + 16 * it;
            const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8);
            const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8 + 4);
            const float t[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[it][e] = t[e] * q0[e >> 2][e & 3] + q1[e >> 2][e & 3];
        }
        if constexpr (SCF) {  // the shortcut conv's accumulators through the same tile (a wave's LDS operations execute in order)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 t = {acc_sc[j][4 * g], acc_sc[j][4 * g + 1], acc_sc[j][4 * g + 2], acc_sc[j][4 * g + 3]};
                *reinterpret_cast<floatx4 *>(ep + r * EROW + 8 * g + 4 * hi) = t;
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int px = (lane >> 2) + 16 * it;
                const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8);
                const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8 + 4);
                const float t[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] += t[e] * qs0[e >> 2][e & 3] + qs1[e >> 2][e & 3];
            }
        } else if (p.mode == EPI_BN_ADD_BN) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] += (float)sc8[j][it][e];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int sl = j * 32 + (lane >> 2) + 16 * it;
            long m;
            if (!slot_pixel(sl, m)) continue;
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)v[it][e];
            *reinterpret_cast<half8 *>(p.out0 + m * p.Cout + cch) = o;
            if (two) {
                half8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[it][e] * q2[e >> 2][e & 3] + q3[e >> 2][e & 3]);
                *reinterpret_cast<half8 *>(p.out1 + m * p.Cout + cch) = z;
            }
        }
    }
#ifdef FRT_ABLATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S2_STAMP(5);
#endif
}

// ------------------------------------------------------------------------------------------------ 64 -> 64, stride 2 (112 -> 56)
// The second conv of the very first unit has ONE 64-channel chunk and 64 output channels: 72 MFMAs per wave between a patch load and
// an epilogue, nothing to software-pipeline - the layer is a stream of 360 MB (205 MB in, shortcut, two outputs).  The overlap comes
// from co-resident workgroups instead: small persistent ones (2 waves = the two 32-cout blocks) at three per CU, each walking a
// contiguous range of output rows (one row = 2 pixel tiles per step).
//  * the wave's 36 weight fragments (fragment-ordered, 36 contiguous KB) stay in 144 registers for the kernel's lifetime: a first
//    version that fetched them per output row moved 516 MB through L2 and was bound by exactly that (42 us with everything else off);
//  * input rows are staged in their NATURAL order (three 113-pixel row slots of 128 B pixels, no phase planes): a row is 14 LDS-DMA
//    instructions of one contiguous kilobyte each (phase planes made every request a 128-byte piece at a 256-byte stride: 2.6 TB/s
//    with nothing else running).  The 16-byte pieces of pixel x sit at piece ^ ((x + 1) / 2 % 8) - applied on the DMA's SOURCE side,
//    the request stays one contiguous kilobyte - so the stride-2 fragment reads are two-way instead of eight-way bank conflicts;
//  * row 2 oy + 1 of one step is row 2 oy' - 1 of the next: slot(y) = (y + 1) % 3, two new rows per step;
//  * the next step's rows are requested right behind the barrier that ends the MFMA loop and land under the epilogue (its transpose
//    tiles have their own 9 KB: 52.6 KB per workgroup - 53 248 is the most that still fits three per CU).
// 135 us (im2col kernel) -> 100 - 105 us = 3.5 TB/s on the 360 MB.  What bounds it now is the dependent chain of a step (row fetch ->
// 72 MFMAs -> shortcut load -> stores) at three chains per CU; the staging alone streams at 4.8 TB/s (43 us).
// Next to co-runners (the pipelined benchmark) a launch lasts 220 - 335 us: with their LDS taken fewer of the three chains per CU are
// resident.  Handing the rows out dynamically (chunks of 4 from a device counter instead of a static split) did not help - 126 us
// alone, 241 us live - so it is not a load-balance effect; at the headline this kernel and the im2col kernel it replaces are equal.
constexpr int C64_ROW = 113 * 128;               // one input row: x = -1 .. 111, 128 bytes per pixel (x = 112 is only read by dead pixel slots)
constexpr int C64_LDS = 3 * C64_ROW + 2 * 32 * 36 * 4;  // + the epilogue tiles with the channel parameters in their pad columns (the dead pixel slots' read overrun, pixel index <= 128, lands there)
__global__ __launch_bounds__(128, 2) void conv_s2c64_kernel(ConvMfmaArgs p, int n_rows) {
    constexpr int Wo = 56, Ho = 56, W = 112, H = 112;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int nblk = gridDim.x, bq = nblk >> 3, brem = nblk & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;  // neighbouring ranges share an L2
    const int row_begin = (int)((long)lid * n_rows / nblk), row_end = (int)((long)(lid + 1) * n_rows / nblk);

    constexpr int EROW = 36;  // floats per pixel row of the epilogue tile (32 + 4 pad)
    float *ep = reinterpret_cast<float *>(smem + 3 * C64_ROW) + wave * (32 * EROW);
    const int chunk = lane & 3;
    const int cch = wave * 32 + chunk * 8;  // this lane's 8 couts in the epilogue

    half8 areg[9][4];  // [kh * 3 + kw][kk]: the fragment-ordered copy in tap order (wf, not the phase-plane order)
    {
        const half_t *wfrag = p.wf + (long)wave * (9 * 4 * 512) + lane * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) areg[t][kk] = *reinterpret_cast<const half8 *>(wfrag + (t * 4 + kk) * 512);
    }
    // channel parameters p0 | p1 | p2 | p3 (64 floats each) live in the 4 pad columns of the two epilogue tiles: parameter t of array a
    // at tile (a >> 1), row ((a & 1) * 64 + t) >> 2, column 32 + (t & 3)
    auto prm = [&](int a, int t) -> float * {
        return reinterpret_cast<float *>(smem + 3 * C64_ROW) + (a >> 1) * (32 * EROW) + ((((a & 1) << 6) + t) >> 2) * EROW + 32 + (t & 3);
    };
    if (tid < 64) {
        const bool two0 = p.mode == EPI_BN_ADD_BN && p.out1;
        *prm(0, tid) = p.p0[tid];
        *prm(1, tid) = p.p1[tid];
        *prm(2, tid) = two0 ? p.p2[tid] : 0.f;
        *prm(3, tid) = two0 ? p.p3[tid] : 0.f;
    }
    if (tid < 24) {  // the x = -1 column of the three slots stays zero (the DMA writes pixels 1 .. 112 only)
        const int sl = tid >> 3, c = tid & 7;
        *reinterpret_cast<floatx4 *>(smem + sl * C64_ROW + c * 16) = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    // DMA: instruction k of a row moves pixels 8 k .. 8 k + 7 (x) = LDS pixels 1 + 8 k ..; lane = (pixel, LDS chunk c), source piece c ^ swizzle
    const int dpx = lane >> 3, dc = lane & 7;
    const int src_even = dpx * 64 + ((dc ^ (((1 + dpx) >> 1) & 7)) << 3);      // k even: ((1 + 8 k + dpx) >> 1) & 7 = ((1 + dpx) >> 1) & 7
    const int src_odd = dpx * 64 + ((dc ^ ((((1 + dpx) >> 1) + 4) & 7)) << 3);  // k odd: + 4
    auto fetch_row = [&](const half_t *img, int y, int k0, int k1) {
        char *dst = smem + ((y + 1) % 3) * C64_ROW + 128;
        const bool row_ok = y >= 0;
        const half_t *rowp = img + y * (W * 64);
#pragma unroll
        for (int k = 0; k < 14; ++k) {
            if (k < k0 || k >= k1) continue;
            const half_t *src = row_ok ? rowp + k * 512 + ((k & 1) ? src_odd : src_even) : p.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + k * 1024), 16, 0, 0);
        }
    };
    auto fetch = [&](int row, bool fresh) {  // input rows of output row `row`: wave 0 the even row 2 oy, wave 1 the odd row 2 oy + 1
        const int b = row / Ho, oy = row - b * Ho;
        const half_t *img = p.x + (long)b * (H * W * 64);
        if (fresh) fetch_row(img, 2 * oy - 1, wave * 7, wave * 7 + 7);  // otherwise inherited from the previous step
        fetch_row(img, 2 * oy + wave, 0, 14);
    };

    if (row_begin < row_end) fetch(row_begin, true);
    // fragment read: pixel x = 2 sl + kw - 1 -> LDS pixel 2 sl + kw, piece pc = 2 kk + hi at chunk pc ^ ((sl + (kw >> 1)) & 7)
    int boff[2][2];  // [tile][kw >> 1]: byte offset of kk = 0 inside a row slot, without the kw pixel offset
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sl = j * 32 + r;
            boff[j][h] = sl * 256 + ((hi ^ ((sl + h) & 7)) << 4);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int row = row_begin; row < row_end; ++row) {
        const int b = row / Ho, oy = row - b * Ho;
        __syncthreads();  // this wave's rows have landed (waited for at the end of the previous step): now everybody's have

        floatx16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const char *rowb = smem + ((2 * oy + kh) % 3) * C64_ROW;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const half8 bf = *reinterpret_cast<const half8 *>(rowb + ((boff[j][kw >> 1] ^ (kk << 5)) + kw * 128));
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[kh * 3 + kw][kk], bf, acc[j], 0, 0, 0);
                    }
        }
        __syncthreads();  // every row read has been consumed: rows 2 oy - 1 and 2 oy are free
        if (row + 1 < row_end) fetch(row + 1, oy + 1 == Ho);

        // epilogue (per wave: 32 couts x 2 pixel tiles) through a wave-private fp32 tile in LDS, as in the kernel above
        const bool two = p.mode == EPI_BN_ADD_BN && p.out1;
        const long m0 = ((long)b * Ho + oy) * Wo;
        const long s0 = p.mode == EPI_BN_ADD_BN ? ((long)b * p.sc_h + oy * p.sc_stride) * p.sc_w : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // per tile: the shortcut and the channel parameters are (re)loaded here - with the 144 weight registers the kernel has no
            // room to keep them across the MFMA loop (requesting the shortcut in front of the row fetch, so that it does not return
            // behind it, was measured: 16 spilled registers, no gain); a compiler barrier keeps them from being hoisted
            asm volatile("" ::: "memory");
            half8 sc8[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int sl = j * 32 + (lane >> 2) + 16 * it;
                if (p.mode == EPI_BN_ADD_BN) sc8[it] = *reinterpret_cast<const half8 *>(p.sc + (s0 + (sl < Wo ? sl * p.sc_stride : 0)) * 64 + cch);
            }
            floatx4 q0[2], q1[2], q2[2], q3[2];  // channel parameters from their LDS copy
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                q0[h] = *reinterpret_cast<const floatx4 *>(prm(0, cch + 4 * h));
                q1[h] = *reinterpret_cast<const floatx4 *>(prm(1, cch + 4 * h));
                q2[h] = *reinterpret_cast<const floatx4 *>(prm(2, cch + 4 * h));
                q3[h] = *reinterpret_cast<const floatx4 *>(prm(3, cch + 4 * h));
            }
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
                const int sl = j * 32 + px;
                if (sl >= Wo) continue;
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * q0[e >> 2][e & 3] + q1[e >> 2][e & 3];
                if (p.mode == EPI_BN_ADD_BN) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)sc8[it][e];
                }
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                *reinterpret_cast<half8 *>(p.out0 + (m0 + sl) * 64 + cch) = o;
                if (two) {
                    half8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[e] * q2[e >> 2][e & 3] + q3[e >> 2][e & 3]);
                    *reinterpret_cast<half8 *>(p.out1 + (m0 + sl) * 64 + cch) = z;
                }
            }
        }
        // wait for the next step's rows but not for this step's stores: memory operations retire in order and the stores (4 per output
        // tensor, issued by every lane group) are the youngest ones - anything else the compiler added only makes the wait stricter
        if (two) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
}

bool s2c64_applies(const ConvMfmaArgs &a) {
    if (!a.wf2 || a.ks != 3 || a.stride != 2 || a.pad != 1 || a.Cout != 64 || a.Cin != 64 || a.splits != 1 || (a.H & 1) || (a.W & 1)) return false;
    if (a.Ho * 2 != a.H || a.Wo * 2 != a.W || a.Wo != 56 || a.Ho != 56) return false;
    if (a.mode != EPI_BN && a.mode != EPI_BN_ADD_BN) return false;
    if (a.mode == EPI_BN_ADD_BN && !a.sc) return false;
    static const bool off = (frt_tuning_env("FRT_CONV_S2") && frt_tuning_env("FRT_CONV_S2")[0] == '0') ||
                            (frt_tuning_env("FRT_CONV_S2C64") && frt_tuning_env("FRT_CONV_S2C64")[0] == '0');
    return !off;
}

#ifdef FRT_ABLATE
// timing build, FRT_S2_STAMPS=1: every launch leaves its stamps in the next slot of a device ring; averages per (NT, Cin) printed at exit
struct S2Stamps {
    unsigned long long *dev = nullptr;
    int n = 0;
    static constexpr int CAP = 2048;
    int kind[CAP];
    ~S2Stamps() {
        if (!dev || !n) return;
        const int m = std::min(n, CAP);
        std::vector<unsigned long long> h((size_t)m * 16);
        if (hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
        std::vector<int> kinds;
        for (int i = 0; i < m; ++i)
            if (std::find(kinds.begin(), kinds.end(), kind[i]) == kinds.end()) kinds.push_back(kind[i]);
        for (int k : kinds) {
            double d[2][6] = {};
            int cnt = 0;
            for (int i = m / 2; i < m; ++i) {  // second half of the run (warm)
                if (kind[i] != k) continue;
                ++cnt;
                for (int b = 0; b < 2; ++b)
                    for (int j = 1; j < 6; ++j) d[b][j] += (double)(h[(size_t)i * 16 + b * 8 + j] - h[(size_t)i * 16 + b * 8 + j - 1]) * 0.01;
            }
            if (!cnt) continue;
            for (int b = 0; b < 2; ++b)
                fprintf(stderr, "[s2 stamps] NT %d Cin %d B %d %s workgroup, %d launches: descriptors + DMA / weight issue %.2f | planes landed %.2f | K loop %.2f | shortcut conv %.2f | epilogue %.2f us\n",
                        k & 15, (k >> 4) & 1023, k >> 14, b ? "last" : "first", cnt, d[b][1] / cnt, d[b][2] / cnt, d[b][3] / cnt, d[b][4] / cnt, d[b][5] / cnt);
        }
    }
};
static S2Stamps g_s2_stamps;
#endif

template <int NT, int PP, bool SEP = false, bool SCF = false>
void launch_s2_t(const ConvMfmaArgs &a, int R, int n_img, hipStream_t s) {
    constexpr size_t lds = (size_t)4 * PP * 4096;
    static_assert(lds <= 160 * 1024 && lds >= 4 * 32 * 36 * 4, "LDS budget / epilogue scratch");
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_s2_kernel<NT, PP, SEP, SCF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int strips = ((a.B + n_img - 1) / n_img) * (a.Ho / R);
    const int linear = (n_img == 1 && R * (a.Wo + 1) <= NT * 32) ? 1 : 0;
#ifdef FRT_ABLATE
    static const bool want_stamps = frt_tuning_env("FRT_S2_STAMPS") != nullptr;
    if (want_stamps && !SEP) {
        if (!g_s2_stamps.dev && hipMalloc(reinterpret_cast<void **>(&g_s2_stamps.dev), S2Stamps::CAP * 16 * 8) != hipSuccess) g_s2_stamps.dev = nullptr;
        if (g_s2_stamps.dev && g_s2_stamps.n < S2Stamps::CAP) {
            ConvMfmaArgs b = a;
            b.outf = reinterpret_cast<float *>(g_s2_stamps.dev + (size_t)g_s2_stamps.n * 16);
            g_s2_stamps.kind[g_s2_stamps.n++] = NT | (a.Cin << 4) | (a.B << 14);
            hipLaunchKernelGGL((conv_s2_kernel<NT, PP, SEP, SCF>), dim3(strips * (a.Cout / 128)), dim3(256), lds, s, b, R, n_img, linear);
            return;
        }
    }
#endif
    hipLaunchKernelGGL((conv_s2_kernel<NT, PP, SEP, SCF>), dim3(strips * (a.Cout / 128)), dim3(256), lds, s, a, R, n_img, linear);
}

// strip geometry; false: not eligible (the im2col kernel takes the layer)
bool s2_geometry(const ConvMfmaArgs &a, int &R, int &n_img, int &nt, int &pp) {
    if (!a.wf2 || a.ks != 3 || a.stride != 2 || a.pad != 1 || a.Cout % 128 || a.Cin % 64 || a.splits != 1 || a.H != a.W || (a.H & 1)) return false;
    if (a.Ho * 2 != a.H || a.Wo * 2 != a.W) return false;
    if (a.mode != EPI_BN && a.mode != EPI_BN_ADD_BN && a.mode != EPI_BN_SE) return false;
    const bool scf = a.mode == EPI_BN_ADD_BN && a.scx;  // fused 1x1 stride-2 shortcut conv instead of a shortcut tensor
    if (scf && !(a.wscf && a.psc0 && a.psc1 && (a.Csc == 64 || a.Csc == 128 || a.Csc == 256))) return false;
    if (!scf && (a.mode == EPI_BN_ADD_BN || a.mode == EPI_BN_SE) && !(a.sc && a.sc_stride == 1 && a.sc_h == a.Ho && a.sc_w == a.Wo)) return false;
    static const bool off = frt_tuning_env("FRT_CONV_S2") && frt_tuning_env("FRT_CONV_S2")[0] == '0';
    if (off) return false;
    const int Wq = a.Wo + 1;
    if (a.Ho * a.Wo <= 64) {  // 14 -> 7: whole images per strip, compact slots; two images (98 pixels, 4 tiles) for full batches
        R = a.Ho;
        n_img = (a.B >= 96 && 2 * a.Ho * a.Wo <= 128) ? 2 : 1;
        nt = n_img * a.Ho * a.Wo <= 64 ? 2 : 4;
    } else {
        n_img = 1;
        R = 0;
        for (int d = a.Ho; d >= 1; --d) {
            if (a.Ho % d || d * Wq > 224) continue;
            R = d;
            break;
        }
        if (!R) return false;
        nt = R * Wq <= 128 ? 4 : 7;
        if (R * a.Wo * 10 < nt * 32 * 7) return false;  // more than 30 % dead pixel slots
    }
    pp = (n_img * (R + 1) * Wq * 9 + 255) / 256;
    return pp <= 9;
}

}  // namespace

// IR-SE: can the launch of `a` (described as EPI_BN_ADD_BN + se_* scratch) run the SE tail in its epilogue?  (row-range strips of one image)
bool conv_s2_se_fused(const ConvMfmaArgs &a) {
    int R, n_img, nt, pp;
    if (s2c64_applies(a) || !s2_geometry(a, R, n_img, nt, pp)) return false;
    if (nt == 7) return n_img == 1 && a.Ho / R <= SE_SPLIT;
    if (nt == 4 && pp <= 5) return n_img == 2 ? R == a.Ho : (n_img == 1 && a.Ho / R <= SE_SPLIT);
    return false;
}

bool conv_s2_applies(const ConvMfmaArgs &a) {
    int R, n_img, nt, pp;
    return s2c64_applies(a) || s2_geometry(a, R, n_img, nt, pp);
}

const char *conv_s2_label(const ConvMfmaArgs &a) {
    int R, n_img, nt, pp;
    if (s2c64_applies(a)) return "conv_s2c64_kernel";
    if (!s2_geometry(a, R, n_img, nt, pp)) return nullptr;
    const bool scf = a.mode == EPI_BN_ADD_BN && a.scx;
    if (a.mode == EPI_BN_SE) return nt == 7 ? "conv_s2_kernel<7, 9, true, false>" : "conv_s2_kernel<4, 5, true, false>";
    if (nt == 7) return scf ? "conv_s2_kernel<7, 9, false, true>" : "conv_s2_kernel<7, 9, false, false>";
    if (nt == 4) return pp <= 5 ? (scf ? "conv_s2_kernel<4, 5, false, true>" : "conv_s2_kernel<4, 5, false, false>")
                                : (scf ? "conv_s2_kernel<4, 9, false, true>" : "conv_s2_kernel<4, 9, false, false>");
    return scf ? "conv_s2_kernel<2, 5, false, true>" : "conv_s2_kernel<2, 5, false, false>";
}

bool launch_conv_s2(const ConvMfmaArgs &a0, hipStream_t s) {
    int R, n_img, nt, pp;
    if (s2c64_applies(a0)) {
        ConvMfmaArgs a = a0;
        a.wf = a0.wf2;  // tap order for this kernel (the host packs wf2 in tap order when Cout == 64, frt_api.cpp)
        static bool attr_done[FRT_MAX_DEVICES] = {};
        if (frt_first_use_on_device(attr_done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_s2c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
        const int n_rows = a.B * a.Ho;
        hipLaunchKernelGGL(conv_s2c64_kernel, dim3(n_rows < 768 ? n_rows : 768), dim3(128), C64_LDS, s, a, n_rows);
        return true;
    }
    if (!s2_geometry(a0, R, n_img, nt, pp)) return false;
    ConvMfmaArgs a = a0;
    a.wf = a0.wf2;  // the kernel streams the stride-2 step order
    const bool scf = a.mode == EPI_BN_ADD_BN && a.scx;
    if (nt == 7 && a.mode == EPI_BN_SE) launch_s2_t<7, 9, true>(a, R, n_img, s);  // (the caller checked conv_s2_se_fused)
    else if (nt == 7 && scf) launch_s2_t<7, 9, false, true>(a, R, n_img, s);
    else if (nt == 7) launch_s2_t<7, 9>(a, R, n_img, s);
    else if (nt == 4 && pp <= 5 && a.mode == EPI_BN_SE) launch_s2_t<4, 5, true>(a, R, n_img, s);
    else if (nt == 4 && pp <= 5 && scf) launch_s2_t<4, 5, false, true>(a, R, n_img, s);
    else if (nt == 4 && pp <= 5) launch_s2_t<4, 5>(a, R, n_img, s);
    else if (nt == 4 && scf) launch_s2_t<4, 9, false, true>(a, R, n_img, s);
    else if (nt == 4) launch_s2_t<4, 9>(a, R, n_img, s);
    else if (pp <= 5 && scf) launch_s2_t<2, 5, false, true>(a, R, n_img, s);
    else if (pp <= 5) launch_s2_t<2, 5>(a, R, n_img, s);
    else return false;
    return true;
}
