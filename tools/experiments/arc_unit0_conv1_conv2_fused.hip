// PARKED EXPERIMENT (round 6; not part of libfrt.so).  The first IR unit as one kernel, built, bit-identical to the two-launch path (all 26 tests of
// tests/test_gpu_embedder.py green with it wired in from 48 faces on), and SLOWER: 329 - 332 us against 176.6 + 102.1 = 279 us (conv64_kernel at 112x112 +
// conv_s2c64_kernel) on the same box.  Timing ablations (us per launch, 128 faces, 256 persistent workgroups x 24.5 tiles): complete 329; conv1 without its
// MFMAs 221; without its B-fragment reads either 131; without conv2's loop 287; without the next patch's DMA 245; without conv1 altogether 92; skeleton
// (barriers, waits, epilogue stores) 45.  conv1 alone is ~ 237 us here: every MFMA needs a fresh 1 KB B fragment from LDS (the A fragment is reused over the
// wave's five pixel tiles, a B fragment by nobody - the second cout block of the same pixels lives in another wave, and one wave cannot hold both blocks'
// weights: 288 registers on top of conv2's 144), so four waves ask the LDS for 128 bytes per clock - its whole read bandwidth - for as long as the MFMAs
// take, and with one wave per SIMD the read phase, the MFMA phase, the T write and the patch DMA add up instead of overlapping.  What it would take: both
// cout blocks of a pixel tile in one wave (weights from LDS or from a second register file's worth of AGPRs), i.e. a 32-channel K split of the weights.
// To build it again: add kernels_arc_unit0 to csrc/Makefile, declare Unit0Args / launch_unit0 in frt_kernels.h as at the top of this kernel's launcher, and
// call launch_unit0 for units[0] in frt_embedder::forward (Unit0Args{Z, w1f, w2f2, prelu, s2, b2, sn, bn, Y, Ynext, Znext, zeros, F}).
//
//   struct Unit0Args { const half_t *x, *wf1, *wf2; const float *slope, *p0, *p1, *p2, *p3; const half_t *sc; half_t *out0, *out1; const half_t *zeros; int B; };
//
// ArcFace IR-50 / IR-SE-50, the FIRST unit (model_irse.py:48-90 with in = depth = 64, stride 2, 112x112 -> 56x56) in ONE kernel (round 6).
//
// As two launches the unit is the slowest stretch of the recogniser: conv1 (3x3, stride 1, 64 -> 64 at 112x112 = 118 GFLOP per 128 faces) on
// conv64_kernel takes 203 us - its 2-row strips read every input row twice (410 MB) and write the 205 MB tensor T - and conv2 (3x3, stride 2)
// on conv_s2c64_kernel reads T back and takes 104 - 112 us: 315 us for 148 GFLOP, bound by the 820 MB that T costs.  Here T never leaves the CU:
//   * a work item is an 8x8 tile of the 56x56 OUTPUT; it needs T at 17x17 positions and the unit's (BatchNorm'd) input at 19x19.  The input
//     patch (361 pixels x 144-byte rows = 52 KB) arrives by LDS-DMA into one of two buffers - the NEXT tile's patch is requested before this
//     tile's first MFMA - so HBM sees the input 1.41 times and nothing else but the shortcut and the two 56x56 outputs (~ 400 MB per pass);
//   * conv1 runs over the 289 T positions as 10 pixel tiles (compact enumeration: slot -> (ty, tx) = (slot / 17, slot % 17)) x 2 cout blocks:
//     wave w owns cout block w & 1 and the five tiles (w >> 1) + 2 i; its 36 weight fragments live in 144 registers for the kernel's life
//     (persistent workgroups), a B fragment is one ds_read_b128 at (base[tile] + tap offset + kk * 32);
//   * PReLU, rounding to fp16 and the zero padding of T (positions outside the 112x112 map) happen in registers; T goes to LDS in pixel rows
//     of 144 bytes (41.6 KB) - the same rounding point as the tensor T of the two-launch path;
//   * conv2 reads T with stride-2 pixel addressing (two-way bank conflicts, 36 of the 216 MFMAs of a wave): wave w = cout block w & 1,
//     pixel tile w >> 1 (4 output rows x 8 columns); its 36 weight fragments are the second 144 registers (one wave per SIMD: 512 registers);
//   * epilogue as conv_s2c64_kernel's: transpose through a wave-private fp32 tile (in the patch buffer conv1 has finished with), folded
//     BatchNorm, + shortcut (the input layer's raw output at the even positions), the NEXT unit's leading BatchNorm as second output.
// Both K loops walk (kh, kw, kk) in the order of conv64_kernel / conv_s2c64_kernel with the same fragments, so the outputs are theirs bit for
// bit (tests/test_gpu_embedder.py: the 128-face pass against passes of 64 / 32 / 8 faces that take other kernels).
// Work per 64 outputs: 720 + 144 MFMAs against 576 + 144 unfused (the 17x17 halo: + 13 % of conv1, + 10 % dead slots of its tenth tile).
#include <cstdlib>
#include <type_traits>

#include "frt_kernels.h"

namespace {

constexpr int U0_PW = 19, U0_TW = 17;            // input patch / T region edge
constexpr int U0_ROWB = 144;                      // bytes per pixel row in LDS (128 data + 16 pad: conflict-free ds_read_b128)
constexpr int U0_NDMA = 13;                       // LDS-DMA instructions per thread per patch: 361 pixels x 9 chunks = 3 249 <= 13 x 256
constexpr int U0_PATCH_B = U0_NDMA * 4096;        // 53 248
constexpr int U0_T_B = 41984;                     // 289 x 144 = 41 616, rounded
constexpr int U0_LDS = 2 * U0_PATCH_B + U0_T_B + 1280;   // 149 504: + conv2's four channel-parameter vectors (read per tile: no registers to spare)
constexpr int U0_EROW = 36;

template <int I, int N, class F>
__device__ __forceinline__ void u0_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        u0_for<I + 1, N>(f);
    }
}

template <bool TWO, int ABL = 0>  // ABL (measurement builds; wrong results): 1 no conv1 MFMAs, 2 no conv1 B reads either, 4 no conv2 loop, 8 no patch DMA after the first, 16 no T write.  TWO: mode EPI_BN_ADD_BN with the second (BatchNorm'd) output; otherwise EPI_BN_ADD_BN without it or EPI_BN (IR-SE: no shortcut)
__global__ __launch_bounds__(256, 1) void conv_unit0_kernel(Unit0Args p, int n_tiles) {
    constexpr int H = 112, W = 112, Ho = 56;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tbuf = smem + 2 * U0_PATCH_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int cb = wave & 1, tw = wave >> 1;

    // ---- weights: this wave's cout block, both convs, fragment order [block][tap][kk][lane][8]
    half8 w1[9][4], w2[9][4];
    {
        const half_t *f1 = p.wf1 + (long)cb * (9 * 4 * 512) + lane * 8, *f2 = p.wf2 + (long)cb * (9 * 4 * 512) + lane * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                w1[t][kk] = *reinterpret_cast<const half8 *>(f1 + (t * 4 + kk) * 512);
                w2[t][kk] = *reinterpret_cast<const half8 *>(f2 + (t * 4 + kk) * 512);
            }
    }

    // ---- conv1 geometry: T slot s = 32 tile + r -> (ty, tx); patch pixel of tap (0, 0) = ty * 19 + tx
    int b1[5], tpos[5];
    unsigned top = 0, left = 0, live = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int s = 32 * (tw + 2 * i) + r;
        const bool ok = s < U0_TW * U0_TW;
        const int ss = ok ? s : 0, ty = ss / U0_TW, tx = ss - ty * U0_TW;
        b1[i] = (ty * U0_PW + tx) * U0_ROWB + hi * 16;
        tpos[i] = ss * U0_ROWB;
        top |= (ty == 0 ? 1u : 0u) << i;
        left |= (tx == 0 ? 1u : 0u) << i;
        live |= (ok ? 1u : 0u) << i;
    }
    // ---- conv2 geometry: output slot r of tile tw -> (oy, ox) = (4 tw + r / 8, r % 8); T pixel of tap (0, 0) = (2 oy) * 17 + 2 ox
    const int b2 = ((2 * (4 * tw + (r >> 3))) * U0_TW + 2 * (r & 7)) * U0_ROWB + hi * 16;
    // ---- patch DMA: chunk g = (q * 4 + wave) * 64 + lane -> patch pixel g / 9, 16-byte piece g % 9 (the ninth piece is padding: zeros)
    int dyx[U0_NDMA];    // (piece << 16) | (py << 8) | px; -1: always zeros (the padding piece, chunks behind the patch)
#pragma unroll
    for (int q = 0; q < U0_NDMA; ++q) {
        const int g = (q * 4 + wave) * 64 + lane;
        const int pix = g / 9, pos = g - pix * 9;
        const int py = pix / U0_PW, px = pix - py * U0_PW;
        const bool ok = pos < 8 && pix < U0_PW * U0_PW;
        dyx[q] = ok ? (pos << 16) | (py << 8) | px : -1;
    }
    auto issue_patch = [&](int tile, int buf) {
        const int b = tile / 49, t = tile - b * 49, ty = t / 7, tx = t - ty * 7;
        const int y0 = 16 * ty - 2, x0 = 16 * tx - 2;  // input pixel of patch pixel (0, 0)
        const half_t *base = p.x + ((long)b * H * W + (long)y0 * W + x0) * 64;  // may point outside the image: only used with valid pixels' offsets
        char *dst = smem + buf * U0_PATCH_B + wave * 1024;
#pragma unroll
        for (int q = 0; q < U0_NDMA; ++q) {
            const int py = (dyx[q] >> 8) & 255, px = dyx[q] & 255, iy = y0 + py, ix = x0 + px;
            const bool ok = dyx[q] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const half_t *src = ok ? base + ((py * W + px) * 64 + ((dyx[q] >> 16) << 3)) : p.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(dst + q * 4096), 16,
                                             0, 0);
        }
    };

    // epilogue parameters of this lane's channel octet (after the transpose): cch = cb * 32 + (lane & 3) * 8
    const int chunk = lane & 3, cch = cb * 32 + chunk * 8;
    float *prm = reinterpret_cast<float *>(smem + 2 * U0_PATCH_B + U0_T_B);  // [5][64]: conv2's p0 .. p3, conv1's PReLU slopes
    if (tid < 64) {
        prm[256 + tid] = p.slope[tid];
        prm[tid] = p.p0[tid];
        prm[64 + tid] = p.p1[tid];
        prm[128 + tid] = TWO ? p.p2[tid] : 0.f;
        prm[192 + tid] = TWO ? p.p3[tid] : 0.f;
    }

    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    issue_patch(tile, 0);
    int cur = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / 49, t49 = tile - b * 49, tyi = t49 / 7, txi = t49 - tyi * 7;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the current patch (and the previous tile's stores)
        __syncthreads();                                   // everybody's; and everybody is done with T and with the other patch buffer
        const int next = tile + (int)gridDim.x;
        if (next < n_tiles && !(ABL & 8)) issue_patch(next, cur ^ 1);
        const char *pb = smem + cur * U0_PATCH_B;

        // ---------------------------------------------------------------- conv1: 5 pixel tiles x 36 steps
        floatx16 acc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        half8 bf[2][5];
        auto rd1 = [&](auto sc, half8 (&dst)[5]) {
            constexpr int S = decltype(sc)::value, TAP = S >> 2, KK = S & 3;
            constexpr int off = ((TAP / 3) * U0_PW + TAP % 3) * U0_ROWB + KK * 32;
#pragma unroll
            for (int i = 0; i < 5; ++i) dst[i] = *reinterpret_cast<const half8 *>(pb + b1[i] + off);
        };
        if (!(ABL & 2)) rd1(std::integral_constant<int, 0>{}, bf[0]);
        u0_for<0, 36>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            if constexpr (S + 1 < 36)
                if (!(ABL & 2)) rd1(std::integral_constant<int, S + 1>{}, bf[(S + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (ABL & 1) asm volatile("" ::"v"(w1[S >> 2][S & 3]), "v"(bf[S & 1][i]));
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[S >> 2][S & 3], bf[S & 1][i], acc[i], 0, 0, 0);
            }
        });
        // PReLU -> fp16 -> T (zero outside the 112x112 map: the tile's first row / column of T positions when the tile touches the border)
        {
            const bool ztop = tyi == 0, zleft = txi == 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const bool zero = (ztop && ((top >> i) & 1u)) || (zleft && ((left >> i) & 1u));
                if (((live >> i) & 1u) && !(ABL & 16)) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const floatx4 sl = *reinterpret_cast<const floatx4 *>(prm + 256 + cb * 32 + 8 * g + 4 * hi);  // this lane's rows (e & 3) + 8 g + 4 hi
                        half4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[i][4 * g + e];
                            const float a = v > 0.f ? v : v * sl[e];
                            o[e] = zero ? (half_t)0.f : (half_t)a;
                        }
                        *reinterpret_cast<half4 *>(tbuf + tpos[i] + (cb * 32 + 8 * g + 4 * hi) * 2) = o;
                    }
                }
            }
        }
        __syncthreads();  // T is complete; the current patch buffer is free (its epilogue tiles live there)

        // ---------------------------------------------------------------- conv2: one pixel tile x 36 steps
        floatx16 a2;
#pragma unroll
        for (int e = 0; e < 16; ++e) a2[e] = 0.f;
        // shortcut of this lane's two output pixels, requested now (lands under the MFMAs)
        const int oyb = 8 * tyi + 4 * tw, oxb = 8 * txi;
        half8 sc8[2];
        if (p.sc) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int px = (lane >> 2) + 16 * it;
                const long m = ((long)b * Ho + oyb + (px >> 3)) * Ho + oxb + (px & 7);
                sc8[it] = *reinterpret_cast<const half8 *>(p.sc + m * 64 + cch);
            }
        }
        half8 cf[2];
        auto rd2 = [&](auto sc) -> half8 {
            constexpr int S = decltype(sc)::value, TAP = S >> 2, KK = S & 3;
            constexpr int off = ((TAP / 3) * U0_TW + TAP % 3) * U0_ROWB + KK * 32;
            return *reinterpret_cast<const half8 *>(tbuf + b2 + off);
        };
        cf[0] = rd2(std::integral_constant<int, 0>{});
        if (!(ABL & 4)) u0_for<0, 36>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            if constexpr (S + 1 < 36) cf[(S + 1) & 1] = rd2(std::integral_constant<int, S + 1>{});
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[S >> 2][S & 3], cf[S & 1], a2, 0, 0, 0);
        });
        // epilogue: 32 couts x 32 pixels of this wave through its fp32 tile
        float *ep = reinterpret_cast<float *>(smem + cur * U0_PATCH_B) + wave * (32 * U0_EROW);
        float q0[8], q1[8], q2[8], q3[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const floatx4 a0 = *reinterpret_cast<const floatx4 *>(prm + cch + 4 * h), a1 = *reinterpret_cast<const floatx4 *>(prm + 64 + cch + 4 * h);
            const floatx4 a2q = *reinterpret_cast<const floatx4 *>(prm + 128 + cch + 4 * h), a3 = *reinterpret_cast<const floatx4 *>(prm + 192 + cch + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                q0[4 * h + e] = a0[e];
                q1[4 * h + e] = a1[e];
                q2[4 * h + e] = a2q[e];
                q3[4 * h + e] = a3[e];
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 v = {a2[4 * g], a2[4 * g + 1], a2[4 * g + 2], a2[4 * g + 3]};
            *reinterpret_cast<floatx4 *>(ep + r * U0_EROW + 8 * g + 4 * hi) = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int px = (lane >> 2) + 16 * it;
            const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + px * U0_EROW + chunk * 8);
            const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + px * U0_EROW + chunk * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * q0[e] + q1[e];
            if (p.sc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)sc8[it][e];
            }
            const long m = ((long)b * Ho + oyb + (px >> 3)) * Ho + oxb + (px & 7);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
            *reinterpret_cast<half8 *>(p.out0 + m * 64 + cch) = o;
            if (TWO) {
                half8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[e] * q2[e] + q3[e]);
                *reinterpret_cast<half8 *>(p.out1 + m * 64 + cch) = z;
            }
        }
        cur ^= 1;
    }
}

}  // namespace

bool launch_unit0(const Unit0Args &a, hipStream_t s) {
    static const bool off = frt_tuning_env("FRT_UNIT0") && frt_tuning_env("FRT_UNIT0")[0] == '0';
    if (off || !a.wf1 || !a.wf2 || a.B < 1) return false;
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_unit0_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, U0_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_unit0_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, U0_LDS);
    }
    const int n_tiles = a.B * 49;
    const int grid = n_tiles < 256 ? n_tiles : 256;
#ifdef FRT_ABLATE
    static const int abl = frt_tuning_env("FRT_UNIT0_ABLATE") ? atoi(frt_tuning_env("FRT_UNIT0_ABLATE")) : 0;
#define U0L(A) if (abl == A) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_unit0_kernel<true, A>), hipFuncAttributeMaxDynamicSharedMemorySize, U0_LDS); hipLaunchKernelGGL((conv_unit0_kernel<true, A>), dim3(grid), dim3(256), U0_LDS, s, a, n_tiles); return true; }
    U0L(1) U0L(3) U0L(4) U0L(8) U0L(16) U0L(7) U0L(23) U0L(31)
#undef U0L
#endif
    if (a.out1) hipLaunchKernelGGL(conv_unit0_kernel<true>, dim3(grid), dim3(256), U0_LDS, s, a, n_tiles);
    else hipLaunchKernelGGL(conv_unit0_kernel<false>, dim3(grid), dim3(256), U0_LDS, s, a, n_tiles);
    return true;
}
