// ArcFace IR-50 stage 1: conv3x3 64 -> 64, stride 1, pad 1 at 112x112 / 56x56 (model_irse.py:58-66 inside units 0-2), fp16 NHWC.
//
// These five layers are 17 % of the network's FLOPs but took 22 % of its time: with only 64 output channels a B (pixel) fragment
// feeds two MFMAs, the generic strip kernel needs a 69 KB patch per strip (one workgroup per CU, 4 waves) and - having a single
// K chunk - runs patch load, MFMA loop and epilogue strictly one after the other: PMC showed the matrix pipe 12.5 % busy and
// the LDS 12 % busy, i.e. a latency-bound kernel.  This kernel is built for exactly this shape:
//   * persistent 2-wave workgroups (wave = one 32-cout block); the wave's whole weight slice [32 cout][576 k] lives in 144
//     VGPRs for the lifetime of the kernel - no weight traffic at all after the first microsecond;
//   * a work item is a strip of 2 image rows x 56 columns (112 pixels = 3.5 MFMA pixel tiles, 4 accumulators): its halo patch
//     (4 x 58 pixels x 144-byte rows) is 34 KB, double-buffered in LDS -> two workgroups per CU; the NEXT strip's patch is
//     DMA'd global -> LDS (global_load_lds, 17 instructions per wave) at the start of a strip, so loads, MFMAs and the
//     previous strip's stores overlap and staging costs no VGPRs;
//   * per (tap, 16-channel step) one A fragment from registers and four ds_read_b128 B fragments, prefetched four steps ahead
//     through a 5-deep register ring; every B address is one per-tile base register plus an immediate;
//   * epilogue through a wave-private fp32 LDS tile: the lane-owns-a-pixel accumulator becomes 64-byte NHWC half-rows, the
//     per-channel parameters a lane needs (8 channels) sit in registers; same arithmetic and rounding points as the generic
//     epilogue (fp32 BN / PReLU / shortcut add, one rounding to fp16 per output).
#include <cstdlib>

#include "frt_kernels.h"

namespace {

constexpr int SW = 56;                 // strip width (pixels)
constexpr int PW = SW + 2;             // patch width
constexpr int PROWB = 144;             // bytes per patch pixel (128 data + 16 pad: conflict-free ds_read_b128)
constexpr int NDMA = (4 * PW * 9 + 127) / 128;  // LDS-DMA instructions per wave per patch (2088 16-byte slots / 2 waves / 64 lanes) = 17
constexpr int PATCH_B = NDMA * 2 * 1024;        // 34816: whole DMA slots
constexpr int RING = 5;                // B-fragment register ring: fragments are requested RING-1 steps (128 clk each) ahead
constexpr int EROW = 36;               // floats per pixel row of the epilogue tile (32 + 4 pad)

template <int MODE, int abl = 0>  // abl (measurement only): 1 no B reads, 2 no MFMA, 4 no epilogue
__global__ __launch_bounds__(128) void conv64_kernel(ConvMfmaArgs p, int n_strips) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *patch = smem;                                           // [2][PATCH_B]
    float *etile = reinterpret_cast<float *>(smem + 2 * PATCH_B); // [2 waves][32][EROW]
    const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int H = p.H, W = p.W;
    const int strips_x = W / SW, strips_per_img = (H / 2) * strips_x;

    // ---- persistent strip walk: XCD-contiguous full rounds, remainder dealt round-robin over the XCDs
    const int nwg = gridDim.x;
    const int bq = nwg >> 3, brem = nwg & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
    const int k_full = n_strips / nwg, rem_strips = n_strips - k_full * nwg;
    auto strip_of = [&](int k) {
        if (k < k_full) return wid + k * nwg;
        return (k == k_full && (int)blockIdx.x < rem_strips) ? k_full * nwg + (int)blockIdx.x : n_strips;
    };

    // ---- weights: lane (cout cb*32 + r, half hi) holds k = tap*64 + kk*16 + 8*hi .. +7 for every (tap, kk)
    half8 wreg[9][4];
    {
        const half_t *wrow = p.w + (long)(cb * 32 + r) * 576 + 8 * hi;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wreg[tap][kk] = *reinterpret_cast<const half8 *>(wrow + tap * 64 + kk * 16);
    }
    // ---- epilogue parameters of this lane's 8 channels (after the transpose a lane always handles the same channel octet)
    const int ec0 = cb * 32 + (lane & 3) * 8;
    float q0[8], q1[8], q2[8], q3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        q0[e] = p.p0[ec0 + e];
        q1[e] = MODE == EPI_PRELU ? 0.f : p.p1[ec0 + e];
        q2[e] = (MODE == EPI_BN_ADD_BN && p.out1) ? p.p2[ec0 + e] : 0.f;
        q3[e] = (MODE == EPI_BN_ADD_BN && p.out1) ? p.p3[ec0 + e] : 0.f;
    }

    // ---- B fragment bases: tile t, lane pixel q = 32 t + r (q >= 112: dead slot, clamped to pixel 0)
    int bbase[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = 32 * t + r;
        const int qq = q < 2 * SW ? q : 0;
        const int row = qq / SW, col = qq - row * SW;
        bbase[t] = (row * PW + col) * PROWB + hi * 16;
    }

    // ---- patch staging by LDS-DMA (global_load_lds: no VGPR round trip, no ds_write, completion counted by vmcnt).  The DMA
    //      writes wave-base + lane*16, so the patch image is slot-linear: slot = pixel*9 + part (part 8 = the 16-byte pad, fed
    //      from the zero buffer like every out-of-image pixel).  Wave w issues DMA q for slots (2q + w)*64 + lane.  Everything
    //      that does not depend on the strip is precomputed: rel = offset inside the strip's window, need = which borders the
    //      slot's pixel touches (bit0 top halo row, bit1 bottom halo row, bit2 left halo column, bit3 right, bit4 always zero).
    unsigned prel[NDMA];
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
        const int slot = (2 * q + cb) * 64 + lane;
        const int pix = slot / 9, part = slot - pix * 9;
        const int pr = pix / PW, pc = pix - pr * PW;
        const bool dead = part == 8 || pix >= 4 * PW;
        const unsigned need = dead ? 16u : ((pr == 0 ? 1u : 0u) | (pr == 3 ? 2u : 0u) | (pc == 0 ? 4u : 0u) | (pc == PW - 1 ? 8u : 0u));
        const unsigned rel = dead ? 0u : (unsigned)((pr * W + pc) * 64 + part * 8);  // halves, relative to (y0, x0)
        prel[q] = rel | (need << 27);
    }
    auto issue_patch = [&](int strip, char *dst) {
        const int b = strip / strips_per_img, rem = strip - b * strips_per_img;
        const int sy = rem / strips_x, sx = rem - sy * strips_x;
        const int y0 = sy * 2 - 1, x0 = sx * SW - 1;
        const unsigned edge = 16u | (y0 < 0 ? 1u : 0u) | (y0 + 3 >= H ? 2u : 0u) | (x0 < 0 ? 4u : 0u) | (x0 + PW - 1 >= W ? 8u : 0u);
        const half_t *win = p.x + ((long)b * H * W + (long)y0 * W + x0) * 64;  // may point before the image: only used with rel of valid pixels
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const bool live = ((prel[q] >> 27) & edge) == 0;
            const half_t *src = live ? win + (prel[q] & 0x7ffffffu) : p.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + (2 * q + cb) * 1024), 16, 0, 0);
        }
    };

    int k = 0;
    int strip = strip_of(0);
    if (strip >= n_strips) return;
    issue_patch(strip, patch);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;

    float *et = etile + cb * 32 * EROW;
    for (;;) {
        const int next = strip_of(k + 1);
        const bool has_next = next < n_strips;
        if (has_next) issue_patch(next, patch + (cur ^ 1) * PATCH_B);  // that buffer's readers retired at the last barrier
        // shortcut values of THIS strip, requested now so that they land behind the MFMA loop (loaded at their use in the epilogue
        // they cost one exposed HBM round trip per pixel tile: 81 vs 52 us against the PReLU variant)
        half8 scv[4][2];
        if (MODE == EPI_BN_ADD_BN) {
            const int b = strip / strips_per_img, rem = strip - b * strips_per_img;
            const int sy = rem / strips_x, sx = rem - sy * strips_x;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int q = 32 * t + (lane >> 2) + 16 * i;
                    const int qq = q < 2 * SW ? q : 0;  // dead slots: clamped, never used
                    const int row = qq / SW, col = qq - row * SW;
                    const long m = (long)b * H * W + (long)(sy * 2 + row) * W + sx * SW + col;
                    scv[t][i] = *reinterpret_cast<const half8 *>(p.sc + m * 64 + ec0);
                }
        }

        floatx16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

        const char *pb = patch + cur * PATCH_B;
        // 36 steps (tap, kk); B fragments of step s+2 are requested before the MFMAs of step s
        half8 bf[RING][4];
        auto read_b = [&](int s, half8 (&dst)[4]) {
            const int tap = s >> 2, kk = s & 3;
            const int kh = tap / 3, kw = tap - kh * 3;
            const int off = (kh * PW + kw) * PROWB + kk * 32;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = *reinterpret_cast<const half8 *>(pb + bbase[t] + off);
        };
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
            if (!(abl & 1)) read_b(s, bf[s]);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + RING - 1 < 36 && !(abl & 1)) read_b(s + RING - 1, bf[(s + RING - 1) % RING]);
            __builtin_amdgcn_sched_barrier(0);
            if (!(abl & 2)) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s >> 2][s & 3], bf[s % RING][t], acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][s & 15] += (float)bf[s % RING][t][0] * (float)wreg[s >> 2][s & 3][0];
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // the next patch has had the whole MFMA loop to land; waiting here (not after the epilogue) keeps this strip's
        // output stores out of the wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- epilogue of this strip
        if (!(abl & 4)) {
            const int b = strip / strips_per_img, rem = strip - b * strips_per_img;
            const int sy = rem / strips_x, sx = rem - sy * strips_x;
            const long img_base = (long)b * H * W;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // lane (r, hi) owns pixel r of the tile and channels (e&3) + 8*(e>>2) + 4*hi of this wave's 32
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                    *reinterpret_cast<floatx4 *>(et + r * EROW + 8 * g + 4 * hi) = v;
                }
                // read back as 128 octets (32 pixels x 4), two per lane: lane -> pixel (lane >> 2) + 16 i, channel octet lane & 3
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int px = (lane >> 2) + 16 * i;
                    const floatx4 v0 = *reinterpret_cast<const floatx4 *>(et + px * EROW + (lane & 3) * 8);
                    const floatx4 v1 = *reinterpret_cast<const floatx4 *>(et + px * EROW + (lane & 3) * 8 + 4);
                    const int q = 32 * t + px;
                    if (q >= 2 * SW) continue;
                    const int row = q / SW, col = q - row * SW;
                    const long m = img_base + (long)(sy * 2 + row) * W + sx * SW + col;
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if (MODE == EPI_PRELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * q0[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * q0[e] + q1[e];
                    }
                    if (MODE == EPI_BN_ADD_BN) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)scv[t][i][e];
                    }
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                    *reinterpret_cast<half8 *>(p.out0 + m * 64 + ec0) = o;
                    if (MODE == EPI_BN_ADD_BN && p.out1) {
                        half8 z;
#pragma unroll
                        for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[e] * q2[e] + q3[e]);
                        *reinterpret_cast<half8 *>(p.out1 + m * 64 + ec0) = z;
                    }
                }
            }
        }
        if (!has_next) break;
        __syncthreads();
        cur ^= 1;
        strip = next;
        ++k;
    }
}

// Round 4, built, measured and parked in tools/experiments/conv64_in_unit0_input_layer_fused.hip: this kernel computing its own input
// patch from the 3-channel crop (the input layer - 4 MFMAs, 16 gathers and a 64-value fp32 PReLU / BN epilogue per 32 patch pixels - inside
// the strip loop, so that the 205 MB tensor z is never written or read).  Bit-identical to the two-kernel path (goldens green) and
// SLOWER: 379 us at 128 faces against 78 (arc_input_mfma_kernel) + 169 (this kernel) = 247 us (profiles/r04/r04k_unit0_fused_in.txt).  The
// strip loop runs one wave per SIMD; the patch build is ~ 3 000 VALU / LDS / store instructions per strip and wave in the same in-order
// stream as the 144 MFMAs (7 261 instructions in the kernel, 800 of them AGPR moves at 498 registers), where the stand-alone input
// kernel spreads the same work over sixteen waves per CU and is bound by its 256 MB of HBM writes.  410 MB of traffic saved, 132 us lost.

// Round 3, measured and removed: a variant in which a wave owns BOTH 32-cout blocks (288 weight registers) and half of the strip's
// pixel tiles, so that every B fragment read from LDS feeds two MFMAs (half the LDS traffic per MFMA, the bound named above).  Correct,
// no spills in the PReLU / BN forms - and slower (profiles/r03/r03b_embed_ab.txt, one box, rocprofv3 averages at 128 faces): the three PReLU
// layers 92.2 -> 108.3 us per launch (the 56x56 ones 48.8 -> 57.8), the shortcut-add form 57.4 -> 79.6 us.  At 464 - 512 registers per
// lane the A operands of half the MFMAs come out of the accumulator file, and the kernel can no longer share a SIMD with another
// wave.  The kernel above stays.

}  // namespace

// Cin == Cout == 64, 3x3, stride 1, pad 1, H even, W a multiple of 56.  false: not this shape (use the generic kernels).
bool conv64_applies(const ConvMfmaArgs &a) {
    if (a.Cin != 64 || a.Cout != 64 || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if ((a.H & 1) || a.W % SW || a.mode == EPI_PARTIAL) return false;
    if (a.mode == EPI_BN_ADD_BN && !(a.sc && a.sc_stride == 1 && a.sc_h == a.H && a.sc_w == a.W)) return false;
    static const bool off = frt_tuning_env("FRT_CONV64") && frt_tuning_env("FRT_CONV64")[0] == '0';
    return !off;
}

bool launch_conv64(const ConvMfmaArgs &a, hipStream_t s) {
    if (!conv64_applies(a)) return false;
    if (a.Cin != 64 || a.Cout != 64 || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if ((a.H & 1) || a.W % SW || a.mode == EPI_PARTIAL) return false;
    if (a.mode == EPI_BN_ADD_BN && !(a.sc && a.sc_stride == 1 && a.sc_h == a.H && a.sc_w == a.W)) return false;
    static const bool off = frt_tuning_env("FRT_CONV64") && frt_tuning_env("FRT_CONV64")[0] == '0';
    if (off) return false;
    const int n_strips = a.B * (a.H / 2) * (a.W / SW);
    int grid = 512;
    if (grid > n_strips) grid = n_strips;
    const size_t lds = 2 * PATCH_B + 2 * 32 * EROW * sizeof(float);
#ifdef FRT_ABLATE
    static const int abl = frt_tuning_env("FRT_C64_ABLATE") ? atoi(frt_tuning_env("FRT_C64_ABLATE")) : 0;  // measurement only: 1 no B reads, 2 no MFMA, 4 no epilogue
#endif
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done)) {  // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_PRELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_BN_ADD_BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
#ifdef FRT_ABLATE
    if (abl && a.mode == EPI_PRELU) {
        static bool ad = false;
        if (!ad) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_PRELU, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_PRELU, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_PRELU, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_PRELU, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_PRELU, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            ad = true;
        }
        if (abl == 1) hipLaunchKernelGGL((conv64_kernel<EPI_PRELU, 1>), dim3(grid), dim3(128), lds, s, a, n_strips);
        else if (abl == 2) hipLaunchKernelGGL((conv64_kernel<EPI_PRELU, 2>), dim3(grid), dim3(128), lds, s, a, n_strips);
        else if (abl == 4) hipLaunchKernelGGL((conv64_kernel<EPI_PRELU, 4>), dim3(grid), dim3(128), lds, s, a, n_strips);
        else if (abl == 5) hipLaunchKernelGGL((conv64_kernel<EPI_PRELU, 5>), dim3(grid), dim3(128), lds, s, a, n_strips);
        else hipLaunchKernelGGL((conv64_kernel<EPI_PRELU, 7>), dim3(grid), dim3(128), lds, s, a, n_strips);
        return true;
    }
#endif
    switch (a.mode) {
        case EPI_PRELU: hipLaunchKernelGGL((conv64_kernel<EPI_PRELU>), dim3(grid), dim3(128), lds, s, a, n_strips); break;
        case EPI_BN: hipLaunchKernelGGL((conv64_kernel<EPI_BN>), dim3(grid), dim3(128), lds, s, a, n_strips); break;
        default: hipLaunchKernelGGL((conv64_kernel<EPI_BN_ADD_BN>), dim3(grid), dim3(128), lds, s, a, n_strips); break;
    }
    return true;
}
