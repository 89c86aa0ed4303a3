// RetinaFace dense 3x3 convs with 64 input channels (FPN merges, fused SSH 64->48) on the fp16 matrix cores at fp32 accuracy - round 5:
// the wave's whole weight slice lives in REGISTERS.
//
// Arithmetic spec: /root/reference/conversion/retina/models/net.py:9-17,40-66,88-96 (BN folded on the host); the fp16 hi/lo split scheme
// (three v_mfma_f32_32x32x16_f16 per 16 input channels, fp32-class accuracy) is kernels_det_conv3h.hip's, and so is every accumulator's
// product order (chunk, tap, hi*hi, hi*lo, lo*hi): the outputs are bit-identical to that kernel's (tools/ubench/det_conv3h_bench.hip).
//
// Why: kernels_det_conv3h.hip walks (tile, 16-channel chunk) steps and restages 46 KB of weights + a patch chunk per step through
// registers -> LDS, one step ahead.  A step is 54 MFMAs per wave = 0.7 us of matrix work; the loads it waits for take 2 - 3 us from
// HBM / MALL, and no schedule of the tap loop changes that (round 5: the hand-pipelined tap loop moved merge1 from 86.4 to 84.8 us;
// ablations in profiles/r05b_conv3h_ablations.txt).  Here the recipe of the recogniser's conv64_kernel (kernels_arc_c64.hip) is applied:
//   * a wave = one 32-cout block x half of a 16 x 8 pixel tile (two 32-pixel blocks); its weights for ALL 9 taps x 64 input channels,
//     hi and lo parts, are 288 registers loaded once per pyramid level - no weight traffic, no weight staging, no per-chunk barrier;
//   * the tile's halo patch holds all 64 channels ([10 x 18 positions][4 chunks][hi16 | lo16], 272-byte rows: conflict-free ds_read_b128),
//     double-buffered (2 x 48 KB): the NEXT tile's raw fp32 values are requested at the top of a tile (48 registers in flight per lane)
//     and have the tile's whole K loop - 216 MFMAs per wave, ~ 3 us - to arrive; they are split and stored after the loop; ONE barrier per tile;
//   * per (chunk, tap) step four ds_read_b128 B fragments feed six MFMAs; fragments are requested one step ahead (two register sets).
// One workgroup per CU (one 512-register wave per SIMD), persistent over the tiles of up to three pyramid levels.
#include <cstdlib>

#include "frt_kernels.h"

namespace {

constexpr int TS = 16, TH = 8;            // tile: 16 columns x 8 rows of output pixels
constexpr int PS = TS + 2;                // patch width
constexpr int NPOS = (TH + 2) * PS;       // 180 halo positions
constexpr int PROWB = 272;                // bytes per patch position: 4 chunks x (16 hi + 16 lo halves) + 16 pad
constexpr int PATCH_B = NPOS * PROWB;     // 48 960
constexpr int QITEMS = NPOS * 16;         // (position, channel quad) items of a patch
constexpr int PPT = (QITEMS + 255) / 256; // 12 per thread (the last round partial)

struct ConvW {
    Conv3Args p[3];
    int tiles_x[3], tiles_y[3], base[4];
};
struct TileW {
    int lv, b, oy0, ox0;
};
__device__ __forceinline__ TileW tile_w(const ConvW &mm, int t) {
    TileW g;
    g.lv = t >= mm.base[2] ? 2 : (t >= mm.base[1] ? 1 : 0);
    const int tx_n = mm.tiles_x[g.lv], per = tx_n * mm.tiles_y[g.lv];
    const int local = t - mm.base[g.lv];
    g.b = local / per;
    const int rem = local - g.b * per;
    const int tyi = rem / tx_n;
    g.oy0 = tyi * TH;
    g.ox0 = (rem - tyi * tx_n) * TS;
    return g;
}

#ifndef FRT_C3W_ABL
#define FRT_C3W_ABL 0  // timing ablations (harness only; wrong results): 1 no K loop, 2 no patch staging, 8 no output stores
#endif

__global__ __launch_bounds__(256) void conv3x3_splitw_kernel(ConvW mm) {
    extern __shared__ __attribute__((aligned(16))) char smem3w[];  // [2][PATCH_B]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hi = lane >> 5;
    const int cb = wave & 1, ph = wave >> 1;  // cout block, pixel half (tile rows 4 ph .. 4 ph + 3)

    const int nwg = gridDim.x;
    const int bq = nwg >> 3, brem = nwg & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
    const int total = mm.base[3];
    const int k_full = total / nwg, rem_tiles = total - k_full * nwg;
    auto tile_of = [&](int k) {
        if (k < k_full) return wid + k * nwg;
        return (k == k_full && (int)blockIdx.x < rem_tiles) ? k_full * nwg + (int)blockIdx.x : total;
    };

    // ---- patch staging: (position, channel quad) items -> registers (raw fp32); split + stored after the K loop
    floatx4 pst[PPT];
    unsigned pok = 0;
    auto fetch_patch = [&](int t) {
        const TileW g = tile_w(mm, t);
        const Conv3Args &a = mm.p[g.lv];
        const long HW = (long)a.H * a.W;
        const float *inb = a.in + (long)g.b * a.Cin * HW;
        pok = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int it = tid + i * 256;
            const bool live = it < QITEMS;
            const int q = live ? it / NPOS : 0, pos = it - (it / NPOS) * NPOS;   // pos fastest: consecutive lanes read consecutive pixels of a row
            const int py = pos / PS, px = pos - py * PS;
            const int iy = g.oy0 - 1 + py, ix = g.ox0 - 1 + px;
            const bool ok = live && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const float *src = inb + (long)(4 * q) * HW + (ok ? iy * a.W + ix : 0);  // clamped, unconditional
            pst[i][0] = src[0];
            pst[i][1] = src[HW];
            pst[i][2] = src[2 * HW];
            pst[i][3] = src[3 * HW];
            pok |= ok ? (1u << i) : 0u;
        }
    };
    auto store_patch = [&](char *dst) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int it = tid + i * 256;
            const int q = it / NPOS, pos = it - q * NPOS;
            const bool ok = (pok >> i) & 1u;
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = ok ? pst[i][e] : 0.f;
                const half_t xh = (half_t)x;
                h[e] = xh;
                l[e] = (half_t)(x - (float)xh);
            }
            if (it < QITEMS) {  // channels 4 q .. 4 q + 3 = chunk q >> 2, slot q & 3
                char *row = dst + pos * PROWB + (q >> 2) * 64 + (q & 3) * 8;
                *reinterpret_cast<half4 *>(row) = h;
                *reinterpret_cast<half4 *>(row + 32) = l;
            }
        }
    };

    // ---- weights: host-packed [chunk][tap][64 rows][hi16 | lo16] halves; lane (r, hi) of cout block cb keeps halves 8 hi .. 8 hi + 7 of the hi
    //      and of the lo part of row 32 cb + r for every (chunk, tap): 4 x 9 x 2 x 4 = 288 registers
    half8 wa[4][9][2];
    auto load_weights = [&](int lv) {
        const half_t *src = mm.p[lv].wh + (long)(cb * 32 + r) * 32 + 8 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                wa[c][t][0] = *reinterpret_cast<const half8 *>(src + (long)(c * 9 + t) * 64 * 32);
                wa[c][t][1] = *reinterpret_cast<const half8 *>(src + (long)(c * 9 + t) * 64 * 32 + 16);
            }
    };
    auto lv_of = [&](int t) { return t >= mm.base[2] ? 2 : (t >= mm.base[1] ? 1 : 0); };

    int k = 0;
    int t0 = tile_of(0);
    if (t0 >= total) return;
    int wlv = lv_of(t0);
    load_weights(wlv);
    fetch_patch(t0);
    store_patch(smem3w);
    __syncthreads();
    int cur = 0;

    // lane geometry: pixel block pb of this wave covers tile rows 2 (2 ph + pb) + (r >> 4), column r & 15
    int bbase[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) bbase[pb] = ((2 * (2 * ph + pb) + (r >> 4)) * PS + (r & 15)) * PROWB + 16 * hi;

    for (;;) {
        const int t1 = tile_of(k + 1);
        const bool v1 = t1 < total;
        if (v1 && !(FRT_C3W_ABL & 2)) fetch_patch(t1);

        floatx16 acc[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pb][e] = 0.f;

        if (!(FRT_C3W_ABL & 1)) {
            const char *pbuf = smem3w + cur * PATCH_B;
            // fragments of one (chunk, tap) step: [pb] b_hi, b_lo - two sets, step s + 1 lands while step s multiplies
            half8 fr[2][4];
            auto read_frag = [&](int s, int i, half8 &dst) {  // i = 2 pb + (0: hi, 1: lo)
                const int c = s / 9, tap = s - 9 * c, kh = tap / 3, kw = tap - 3 * kh;
                dst = *reinterpret_cast<const half8 *>(pbuf + bbase[i >> 1] + (kh * PS + kw) * PROWB + c * 64 + 32 * (i & 1));
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) read_frag(0, i, fr[0][i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                const int c = s / 9, tap = s - 9 * c;
                half8(&f)[4] = fr[s & 1];
                half8(&n)[4] = fr[(s + 1) & 1];
                // MFMA m: product m / 2 (hi*hi, hi*lo, lo*hi), pixel block m % 2 - an accumulator's next MFMA is two issue slots away.  Behind the
                // first four, one fragment of step s + 1 each, in the order that step needs them (b_hi(0), b_hi(1), b_lo(0), b_lo(1)).
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const int prod = m >> 1, pb = m & 1;
                    acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(prod == 2 ? wa[c][tap][1] : wa[c][tap][0], prod == 1 ? f[2 * pb + 1] : f[2 * pb], acc[pb], 0, 0, 0);
                    constexpr int ord[4] = {0, 2, 1, 3};
                    if (s < 35 && m < 4) read_frag(s + 1, ord[m], n[ord[m]]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- tile finished: lane (r, hi) owns pixel r of its blocks and channels cb*32 + (e&3) + 8*(e>>2) + 4*hi
        {
            const TileW g = tile_w(mm, t0);
            const Conv3Args &a = mm.p[g.lv];
            const long HoWo = (long)a.Ho * a.Wo;
            float bias[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                bias[e] = a.b[co < a.Cout ? co : 0];
            }
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int oy = g.oy0 + 2 * (2 * ph + pb) + (r >> 4), ox = g.ox0 + (r & 15);
                const bool inside = oy < a.Ho && ox < a.Wo;
                const long pix = inside ? (long)oy * a.Wo + ox : 0;
                float *o1 = a.out + ((long)g.b * a.out_ctotal + a.out_coff) * HoWo + pix;
                float *o2 = a.out2 ? a.out2 + ((long)g.b * a.out2_ctotal + a.out2_coff - a.split) * HoWo + pix : o1;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    float v = acc[pb][e] + bias[e];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (inside && co < a.Cout && (!(FRT_C3W_ABL & 8) || v == 12345.678f)) (co < a.split ? o1 : o2)[co * HoWo] = v;
                }
            }
        }
        if (!v1) break;
        if (!(FRT_C3W_ABL & 2)) store_patch(smem3w + (cur ^ 1) * PATCH_B);  // the other buffer: its readers passed the last barrier
        const int nlv = lv_of(t1);
        if (nlv != wlv) {  // the next tile belongs to another pyramid level: its weights (happens at most twice per workgroup)
            wlv = nlv;
            load_weights(wlv);
        }
        __syncthreads();
        cur ^= 1;
        t0 = t1;
        ++k;
    }
}

}  // namespace

// Up to 3 same-shaped stride-1 problems with Cin == 64 and 32 < Cout <= 64 in one launch.  false: shape not covered / split weights absent.
bool launch_conv3x3_splitw(const Conv3Args *a, int n, hipStream_t s) {
    static const bool off = frt_tuning_env("FRT_DET_SPLITW") && frt_tuning_env("FRT_DET_SPLITW")[0] == '0';
    if (off || n < 1 || n > 3) return false;
    ConvW mm;
    int base = 0;
    for (int i = 0; i < 3; ++i) {
        const Conv3Args &p = a[i < n ? i : 0];
        if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo || p.Cin != 64 || p.Cout != a[0].Cout || p.Cout > 64 || p.Cout <= 32 || !p.wh) return false;
        mm.p[i] = p;
        if (!mm.p[i].out2) mm.p[i].split = p.Cout;
        mm.tiles_x[i] = (p.Wo + TS - 1) / TS;
        mm.tiles_y[i] = (p.Ho + TH - 1) / TH;
        mm.base[i] = base;
        if (i < n) base += p.B * mm.tiles_x[i] * mm.tiles_y[i];
    }
    for (int i = n; i < 4; ++i) mm.base[i] = base;
    if (base < 1) return true;
    const size_t lds = 2 * (size_t)PATCH_B;  // 97.9 KB: one workgroup per CU
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_splitw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int grid = 256;
    if (grid > base) grid = base;
    hipLaunchKernelGGL(conv3x3_splitw_kernel, dim3(grid), dim3(256), lds, s, mm);
    return true;
}
