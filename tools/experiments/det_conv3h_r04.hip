// RetinaFace dense 3x3 convs with 64 input channels (FPN merges, fused SSH 64->48) on the fp16 matrix cores at fp32 accuracy.
//
// Arithmetic spec: /root/reference/conversion/retina/models/net.py:9-17,40-66,88-96 (BN folded on the host).
//
// The fp32 MFMA kernel (kernels_det_conv3.hip) is bound by the fp32 matrix rate (157 TF, ~50 % reached: 181 us for the 80x80 merge).
// Here every fp32 value x is split into two fp16 numbers, hi = fp16(x) and lo = fp16(x - hi) (x = hi + lo up to 2^-22 |x|), and
//      a * b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (the dropped lo*lo term is 2^-22 relative)
// is accumulated by three v_mfma_f32_32x32x16_f16 per 16 input channels - every fp16 x fp16 product is exact in fp32, the
// accumulation is fp32, so the result carries fp32-class error (~1e-6 relative per layer, inside the detector's tolerances)
// at 96 matrix-pipe clocks per 16 channels instead of 512.  Weights are split on the host, activations when the halo patch is
// staged into LDS (the patch costs the same LDS bytes as fp32: 2 x 2 B).
//
//   * persistent workgroups, 4 waves; tile = 16 columns x 8*PBW rows; a wave owns PBW 32-pixel blocks x all 64 output channels.
//     PBW = 1 (8x16 tiles, 75 KB of LDS, two workgroups per CU) measured 27 / 82 / 95 us on merge2 / merge1 / fused SSH against
//     59 / 116 / 133 us for PBW = 2 (16x16 tiles, 98 KB, one workgroup per CU) and 54 / 181 / 223 us for the fp32 MFMA kernel:
//     a second resident workgroup hides the per-step latencies better than the 2x operand reuse of the big tile pays;
//   * step = (tile, 16-channel chunk): all 9 taps of the chunk's weights [9][64][hi16|lo16] sit in LDS (46 KB, single buffer:
//     the next chunk's copy waits in registers during the step), the halo patch chunk [(8*PBW+2)x18][hi16|lo16] is
//     double-buffered; both are fetched one step ahead; two barriers per step (54*PBW MFMAs per wave);
//   * LDS rows are 80 B (64 + 16 pad): conflict-free ds_read_b128 for both operands;
//   * same epilogue conventions as the fp32 kernel (bias, ReLU, channel-split second output, up to 3 pyramid levels per launch).
#include <cstdlib>

#include "frt_kernels.h"

namespace {

constexpr int TS = 16;                    // tile width (output pixels); tile height = 8 * PBW (PBW pixel blocks per wave)
constexpr int PS = TS + 2;                // patch width
constexpr int ROWH = 40;                  // halves per LDS row: 16 hi + 16 lo + 8 pad (80 B)
constexpr int WCH_H = 9 * 64 * ROWH;      // halves of one weight chunk in LDS
constexpr int WUNITS = 9 * 64 * 4;        // 16-byte units of a weight chunk (hi 2 + lo 2 per row)
constexpr int WPT = WUNITS / 256;         // 9

struct Conv3H {
    Conv3Args p[3];
    int tiles_x[3], tiles_y[3], base[4];
};

struct TileG {
    int lv, b, oy0, ox0;
};
template <int PBW>
__device__ __forceinline__ TileG tile_g(const Conv3H &mm, int t) {
    TileG g;
    g.lv = t >= mm.base[2] ? 2 : (t >= mm.base[1] ? 1 : 0);
    const int tx_n = mm.tiles_x[g.lv], per = tx_n * mm.tiles_y[g.lv];
    const int local = t - mm.base[g.lv];
    g.b = local / per;
    const int rem = local - g.b * per;
    const int tyi = rem / tx_n;
    g.oy0 = tyi * (8 * PBW);
    g.ox0 = (rem - tyi * tx_n) * TS;
    return g;
}

template <int PBW>
__global__ __launch_bounds__(256) void conv3x3_split_kernel(Conv3H mm) {
    constexpr int TH = 8 * PBW;               // tile height
    constexpr int NPOS = (TH + 2) * PS;       // halo positions (180 / 324)
    constexpr int PATCH_H = NPOS * ROWH;      // halves per patch buffer
    constexpr int PITEMS = NPOS * 4;          // (position, channel quad) items of a patch chunk
    constexpr int PPT = (PITEMS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem3h[];
    half_t *pbuf = reinterpret_cast<half_t *>(smem3h);        // [2][NPOS][ROWH]
    half_t *wbuf = pbuf + 2 * PATCH_H;                        // [9][64][ROWH]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hi = lane >> 5;

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
    constexpr int NCH = 4;  // 64 input channels = 4 chunks of 16

    // ---- staging: patch chunk (position, channel quad) items -> registers (raw fp32), split + stored later
    floatx4 pst[PPT];
    unsigned pok = 0;
    auto fetch_patch = [&](int t, int c) {
        const TileG g = tile_g<PBW>(mm, t);
        const Conv3Args &a = mm.p[g.lv];
        const long HW = (long)a.H * a.W;
        const float *inb = a.in + ((long)g.b * a.Cin + c * 16) * HW;
        pok = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int it = tid + i * 256;
            const int q = it / NPOS, pos = it - q * NPOS;   // pos fastest: consecutive lanes read consecutive pixels of a row
            const int py = pos / PS, px = pos - py * PS;
            const int iy = g.oy0 - 1 + py, ix = g.ox0 - 1 + px;
            const bool ok = it < PITEMS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const float *src = inb + (long)(4 * (it < PITEMS ? q : 0)) * HW + (ok ? iy * a.W + ix : 0);  // clamped, unconditional
            pst[i][0] = src[0];
            pst[i][1] = src[HW];
            pst[i][2] = src[2 * HW];
            pst[i][3] = src[3 * HW];
            pok |= ok ? (1u << i) : 0u;
        }
    };
    auto store_patch = [&](half_t *dst) {
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
            if (it < PITEMS) {
                *reinterpret_cast<half4 *>(dst + pos * ROWH + 4 * q) = h;
                *reinterpret_cast<half4 *>(dst + pos * ROWH + 16 + 4 * q) = l;
            }
        }
    };
    // weights: host-packed [chunk][9][64][32 halves = hi16|lo16]; 2304 16-byte units per chunk, 9 per thread
    half8 wst[WPT];
    auto fetch_weights = [&](int lv, int c) {
        const half_t *src = mm.p[lv].wh + (long)c * (9 * 64 * 32);
#pragma unroll
        for (int i = 0; i < WPT; ++i) wst[i] = *reinterpret_cast<const half8 *>(src + (long)(tid + i * 256) * 8);
    };
    auto store_weights = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int u = tid + i * 256;
            const int row = u >> 2, part = u & 3;  // row = tap*64 + cout; part: hi0, hi1, lo0, lo1 (8 halves each)
            *reinterpret_cast<half8 *>(wbuf + row * ROWH + part * 8) = wst[i];
        }
    };

    struct Step {
        int k, t, c, pc;
    };
    auto advance = [&](Step s) {
        s.pc ^= 1;
        if (++s.c == NCH) {
            s.c = 0;
            s.t = tile_of(++s.k);
        }
        return s;
    };
    auto lv_of = [&](int t) { return t >= mm.base[2] ? 2 : (t >= mm.base[1] ? 1 : 0); };

    Step s0{0, tile_of(0), 0, 0};
    if (s0.t >= total) return;
    Step s1 = advance(s0);

    fetch_patch(s0.t, 0);
    fetch_weights(lv_of(s0.t), 0);
    store_patch(pbuf);
    store_weights();
    __syncthreads();

    // lane geometry: pixel block pb of this wave covers tile rows 2*(PBW*wave + pb) + (r >> 4), column r & 15
    int bbase[PBW];
#pragma unroll
    for (int pb = 0; pb < PBW; ++pb) bbase[pb] = ((2 * (PBW * wave + pb) + (r >> 4)) * PS + (r & 15)) * ROWH + 8 * hi;
    const int abase = r * ROWH + 8 * hi;

    floatx16 acc[PBW][2];  // [pixel block][cout block]
#pragma unroll
    for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pb][cb][e] = 0.f;

    for (;;) {
        const bool v1 = s1.t < total;
        if (v1) {
            fetch_patch(s1.t, s1.c);
            fetch_weights(lv_of(s1.t), s1.c);
        }
        const half_t *pb_ = pbuf + s0.pc * PATCH_H;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            const int poff = (kh * PS + kw) * ROWH;
            half8 ah[2], al[2], bh[PBW], bl[PBW];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                ah[cb] = *reinterpret_cast<const half8 *>(wbuf + (tap * 64 + cb * 32) * ROWH + abase);
                al[cb] = *reinterpret_cast<const half8 *>(wbuf + (tap * 64 + cb * 32) * ROWH + abase + 16);
            }
#pragma unroll
            for (int pb = 0; pb < PBW; ++pb) {
                bh[pb] = *reinterpret_cast<const half8 *>(pb_ + bbase[pb] + poff);
                bl[pb] = *reinterpret_cast<const half8 *>(pb_ + bbase[pb] + poff + 16);
            }
#pragma unroll
            for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    acc[pb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bh[pb], acc[pb][cb], 0, 0, 0);
                    acc[pb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bl[pb], acc[pb][cb], 0, 0, 0);
                    acc[pb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], bh[pb], acc[pb][cb], 0, 0, 0);
                }
        }
        if (s0.c == NCH - 1) {
            // ---- tile finished: lane (r, hi) owns pixel r of its block and channels cb*32 + (e&3) + 8*(e>>2) + 4*hi
            const TileG g = tile_g<PBW>(mm, s0.t);
            const Conv3Args &a = mm.p[g.lv];
            const long HoWo = (long)a.Ho * a.Wo;
#pragma unroll
            for (int pb = 0; pb < PBW; ++pb) {
                const int oy = g.oy0 + 2 * (PBW * wave + pb) + (r >> 4), ox = g.ox0 + (r & 15);
                const bool inside = oy < a.Ho && ox < a.Wo;
                const long pix = inside ? (long)oy * a.Wo + ox : 0;
                float *o1 = a.out + ((long)g.b * a.out_ctotal + a.out_coff) * HoWo + pix;
                float *o2 = a.out2 ? a.out2 + ((long)g.b * a.out2_ctotal + a.out2_coff - a.split) * HoWo + pix : o1;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                        float v = acc[pb][cb][e] + a.b[co < a.Cout ? co : 0];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (inside && co < a.Cout) (co < a.split ? o1 : o2)[co * HoWo] = v;
                        acc[pb][cb][e] = 0.f;
                    }
            }
        }
        if (!v1) break;
        store_patch(pbuf + s1.pc * PATCH_H);  // the other patch buffer: its readers finished a step ago
        __syncthreads();                      // everybody is done with this step's weights
        store_weights();
        __syncthreads();
        s0 = s1;
        s1 = advance(s1);
    }
}

}  // namespace

// Up to 3 same-shaped stride-1 problems with Cin == 64 in one launch.  false: shape not covered / split weights absent.
bool launch_conv3x3_split(const Conv3Args *a, int n, hipStream_t s) {
    static const bool off = frt_tuning_env("FRT_DET_SPLIT") && frt_tuning_env("FRT_DET_SPLIT")[0] == '0';
    if (off || n < 1 || n > 3) return false;
    static const int pbw = frt_tuning_env("FRT_DET_SPLIT_PBW") ? atoi(frt_tuning_env("FRT_DET_SPLIT_PBW")) : 1;
    const int th = 8 * pbw;
    Conv3H mm;
    int base = 0;
    for (int i = 0; i < 3; ++i) {
        const Conv3Args &p = a[i < n ? i : 0];
        if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo || p.Cin != 64 || p.Cout != a[0].Cout || p.Cout > 64 || p.Cout < 16 || !p.wh) return false;
        mm.p[i] = p;
        if (!mm.p[i].out2) mm.p[i].split = p.Cout;
        mm.tiles_x[i] = (p.Wo + TS - 1) / TS;
        mm.tiles_y[i] = (p.Ho + th - 1) / th;
        mm.base[i] = base;
        if (i < n) base += p.B * mm.tiles_x[i] * mm.tiles_y[i];
    }
    for (int i = n; i < 4; ++i) mm.base[i] = base;
    const size_t lds = (size_t)(2 * (th + 2) * PS * ROWH + WCH_H) * sizeof(half_t);  // 75 KB (two workgroups per CU) / 98 KB
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_split_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    }
    int grid = pbw == 1 ? 512 : 256;
    if (grid > base) grid = base;
    if (pbw == 1) hipLaunchKernelGGL(conv3x3_split_kernel<1>, dim3(grid), dim3(256), lds, s, mm);
    else hipLaunchKernelGGL(conv3x3_split_kernel<2>, dim3(grid), dim3(256), lds, s, mm);
    return true;
}
