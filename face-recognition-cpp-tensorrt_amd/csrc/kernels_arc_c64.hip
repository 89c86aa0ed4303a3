// ArcFace IR-50 stage 1: conv3x3 64 -> 64, stride 1, pad 1 at 112x112 / 56x56 (model_irse.py:58-66 inside units 0-2), fp16 NHWC.
//
// These five layers are 17 % of the network's FLOPs but took 22 % of its time: with only 64 output channels a B (pixel) fragment
// feeds two MFMAs, the generic strip kernel needs a 69 KB patch per strip (one workgroup per CU, 4 waves) and - having a single
// K chunk - runs patch load, MFMA loop and epilogue strictly one after the other: PMC showed the matrix pipe 12.5 % busy and
// the LDS 12 % busy, i.e. a latency-bound kernel.  This kernel is built for exactly this shape:
//   * persistent 2-wave workgroups (wave = one 32-cout block); the wave's whole weight slice [32 cout][576 k] lives in 144
//     VGPRs for the lifetime of the kernel - no weight traffic at all after the first microsecond;
//   * a work item is a strip of 2 image rows x 56 columns (112 pixels = 3.5 MFMA pixel tiles, 4 accumulators): its halo patch
//     (4 x 58 pixels x 144-byte rows) is 33 KB, double-buffered in LDS -> two workgroups per CU; the NEXT strip's patch is
//     fetched global -> registers at the start of a strip and written to the other buffer at its end, so loads, MFMAs and the
//     previous strip's stores overlap;
//   * per (tap, 16-channel step) one A fragment from registers and four ds_read_b128 B fragments, prefetched two steps ahead
//     through a 3-deep register ring; every B address is one per-tile base register plus an immediate;
//   * epilogue through a wave-private fp32 LDS tile: the lane-owns-a-pixel accumulator becomes 64-byte NHWC half-rows, the
//     per-channel parameters a lane needs (8 channels) sit in registers; same arithmetic and rounding points as the generic
//     epilogue (fp32 BN / PReLU / shortcut add, one rounding to fp16 per output).
#include <cstdlib>

#include "frt_kernels.h"

namespace {

constexpr int SW = 56;                 // strip width (pixels)
constexpr int PW = SW + 2;             // patch width
constexpr int PROWB = 144;             // bytes per patch pixel (128 data + 16 pad: conflict-free ds_read_b128)
constexpr int PATCH_B = 4 * PW * PROWB;  // 33408
constexpr int NSEG = 4 * PW * 8;       // 16-byte segments per patch = 1856
constexpr int SPT = (NSEG + 127) / 128;  // per thread = 15
constexpr int EROW = 36;               // floats per pixel row of the epilogue tile (32 + 4 pad)

template <int MODE>
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

    // ---- patch staging: segment j = tid + 128 i  ->  patch pixel j / 8, 16-byte part j % 8
    half8 pst[SPT];
    unsigned pok = 0;
    auto fetch_patch = [&](int strip) {
        const int b = strip / strips_per_img, rem = strip - b * strips_per_img;
        const int sy = rem / strips_x, sx = rem - sy * strips_x;
        const int y0 = sy * 2 - 1, x0 = sx * SW - 1;
        const half_t *img = p.x + (long)b * H * W * 64;
        pok = 0;
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int j = tid + 128 * i;
            const int pix = j >> 3, part = j & 7;
            const int pr = pix / PW, pc = pix - pr * PW;
            const int iy = y0 + pr, ix = x0 + pc;
            const bool ok = j < NSEG && iy >= 0 && iy < H && ix >= 0 && ix < W;
            // unconditional load from a clamped address; zeroed at store time (no load behind a branch, no early wait)
            pst[i] = *reinterpret_cast<const half8 *>(img + ((long)(ok ? iy : 0) * W + (ok ? ix : 0)) * 64 + part * 8);
            pok |= ok ? (1u << i) : 0u;
        }
    };
    auto store_patch = [&](char *dst) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int j = tid + 128 * i;
            const int pix = j >> 3, part = j & 7;
            half8 v = pst[i];
            if (!((pok >> i) & 1u)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
            }
            if (j < NSEG) *reinterpret_cast<half8 *>(dst + pix * PROWB + part * 16) = v;
        }
    };

    int k = 0;
    int strip = strip_of(0);
    if (strip >= n_strips) return;
    fetch_patch(strip);
    store_patch(patch);
    __syncthreads();
    int cur = 0;

    float *et = etile + cb * 32 * EROW;
    for (;;) {
        const int next = strip_of(k + 1);
        const bool has_next = next < n_strips;
        if (has_next) fetch_patch(next);

        floatx16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

        const char *pb = patch + cur * PATCH_B;
        // 36 steps (tap, kk); B fragments of step s+2 are requested before the MFMAs of step s
        half8 bf[3][4];
        auto read_b = [&](int s, half8 (&dst)[4]) {
            const int tap = s >> 2, kk = s & 3;
            const int kh = tap / 3, kw = tap - kh * 3;
            const int off = (kh * PW + kw) * PROWB + kk * 32;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = *reinterpret_cast<const half8 *>(pb + bbase[t] + off);
        };
        read_b(0, bf[0]);
        read_b(1, bf[1]);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 2 < 36) read_b(s + 2, bf[(s + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s >> 2][s & 3], bf[s % 3][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue of this strip
        {
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
                        const half8 s8 = *reinterpret_cast<const half8 *>(p.sc + m * 64 + ec0);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)s8[e];
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
        store_patch(patch + (cur ^ 1) * PATCH_B);
        __syncthreads();
        cur ^= 1;
        strip = next;
        ++k;
    }
}

}  // namespace

// Cin == Cout == 64, 3x3, stride 1, pad 1, H even, W a multiple of 56.  false: not this shape (use the generic kernels).
bool conv64_applies(const ConvMfmaArgs &a) {
    if (a.Cin != 64 || a.Cout != 64 || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if ((a.H & 1) || a.W % SW || a.mode == EPI_PARTIAL) return false;
    if (a.mode == EPI_BN_ADD_BN && !(a.sc && a.sc_stride == 1 && a.sc_h == a.H && a.sc_w == a.W)) return false;
    static const bool off = getenv("FRT_CONV64") && getenv("FRT_CONV64")[0] == '0';
    return !off;
}

bool launch_conv64(const ConvMfmaArgs &a, hipStream_t s) {
    if (!conv64_applies(a)) return false;
    if (a.Cin != 64 || a.Cout != 64 || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if ((a.H & 1) || a.W % SW || a.mode == EPI_PARTIAL) return false;
    if (a.mode == EPI_BN_ADD_BN && !(a.sc && a.sc_stride == 1 && a.sc_h == a.H && a.sc_w == a.W)) return false;
    static const bool off = getenv("FRT_CONV64") && getenv("FRT_CONV64")[0] == '0';
    if (off) return false;
    const int n_strips = a.B * (a.H / 2) * (a.W / SW);
    int grid = 512;
    if (grid > n_strips) grid = n_strips;
    const size_t lds = 2 * PATCH_B + 2 * 32 * EROW * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {  // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_PRELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_kernel<EPI_BN_ADD_BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    switch (a.mode) {
        case EPI_PRELU: hipLaunchKernelGGL((conv64_kernel<EPI_PRELU>), dim3(grid), dim3(128), lds, s, a, n_strips); break;
        case EPI_BN: hipLaunchKernelGGL((conv64_kernel<EPI_BN>), dim3(grid), dim3(128), lds, s, a, n_strips); break;
        default: hipLaunchKernelGGL((conv64_kernel<EPI_BN_ADD_BN>), dim3(grid), dim3(128), lds, s, a, n_strips); break;
    }
    return true;
}
