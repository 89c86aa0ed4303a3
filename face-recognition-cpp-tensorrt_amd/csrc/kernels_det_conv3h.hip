// RetinaFace dense 3x3 convs with 64 or 16 input channels (FPN merges, fused SSH 64->48, the SSH 16->32 / 16->16 convs) on the fp16
// matrix cores at fp32 accuracy.
//
// Arithmetic spec: /root/reference/conversion/retina/models/net.py:9-17,40-66,88-96 (BN folded on the host).
//
// Every fp32 value x is split into two fp16 numbers, hi = fp16(x) and lo = fp16(x - hi) (x = hi + lo up to 2^-22 |x|), and
//      a * b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (the dropped lo*lo term is 2^-22 relative)
// is accumulated by three v_mfma_f32_32x32x16_f16 per 16 input channels - every fp16 x fp16 product is exact in fp32, the
// accumulation is fp32, so the result carries fp32-class error (~1e-6 relative per layer, inside the detector's tolerances)
// at 96 matrix-pipe clocks per 16 channels instead of the 512 of the true-fp32 MFMA (kernels_det_conv3.hip).  Weights are split on the
// host, activations when the halo patch is staged into LDS (the patch costs the same LDS bytes as fp32: 2 x 2 B).
//
//   * persistent workgroups, 4 waves; tile = 16 columns x 8 rows; a wave owns one 32-pixel block (two tile rows) x all output channels
//     (NCB blocks of 32); 52 - 75 KB of LDS: two workgroups per CU, the second one's staging phase under the first one's MFMAs;
//   * step = (tile, 16-channel chunk): all 9 taps of the chunk's weights [9][32 NCB][hi16|lo16] sit in LDS (single buffer: the next
//     chunk's copy waits in registers during the step), the halo patch chunk [10 x 18][hi16|lo16] is double-buffered; both are fetched
//     one step ahead.  With ONE chunk (16 input channels) the weights stay in LDS for as long as the pyramid level does not change;
//   * LDS rows are 80 B (64 + 16 pad): conflict-free ds_read_b128 for both operands;
//   * round 5: the tap loop is software-pipelined by hand.  Left to the compiler every tap was "six ds_read_b128, wait for all of them,
//     six MFMAs" with the three MFMAs of an accumulator back to back - two exposed LDS round trips per tap, 18 per step, and a step took
//     ~ 15 k cycles for 1.7 k cycles of matrix work per wave (merge1 98 us).  Now the fragments of tap t + 1 are requested one by one behind
//     the MFMAs of tap t (two register sets, sched_barrier after every MFMA + read pair) and the MFMAs of the two cout blocks alternate, so
//     that an accumulator's next MFMA is two issue slots away.  Each accumulator still sees its products in the order (chunk, tap,
//     hi*hi, hi*lo, lo*hi): the results are bit-identical to the round-4 kernel (tools/ubench/det_conv3h_bench.hip compares them);
//   * same epilogue conventions as the fp32 kernel (bias, ReLU, channel-split second output, up to 3 pyramid levels per launch).
#include <cstdlib>

#include "frt_kernels.h"

namespace {

#ifndef FRT_C3H_ABL
#define FRT_C3H_ABL 0   // timing ablations of tools/ubench/det_conv3h_bench.hip (wrong results by design): 1 no tap loop, 2 no patch staging,
#endif                  // 4 no weight staging, 8 no output stores, 16 no barriers
constexpr int TS = 16;                    // tile width (output pixels); tile height 8
constexpr int TH = 8;
constexpr int PS = TS + 2;                // patch width
constexpr int ROWH = 40;                  // halves per LDS row: 16 hi + 16 lo + 8 pad (80 B)
constexpr int NPOS = (TH + 2) * PS;       // halo positions (180)
constexpr int PATCH_H = NPOS * ROWH;      // halves per patch buffer
constexpr int PITEMS = NPOS * 4;          // (position, channel quad) items of a patch chunk
constexpr int PPT = (PITEMS + 255) / 256; // 3

struct Conv3H {
    Conv3Args p[3];
    int tiles_x[3], tiles_y[3], base[4];
};

struct TileG {
    int lv, b, oy0, ox0;
};
__device__ __forceinline__ TileG tile_g(const Conv3H &mm, int t) {
    TileG g;
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

// NCH = input channels / 16 (4: the 64-channel convs, 1: the 16-channel SSH convs); NCB = 32-wide output-channel blocks (Cout <= 32 NCB)
template <int NCH, int NCB>
__global__ __launch_bounds__(256) void conv3x3_split_kernel(Conv3H mm) {
    constexpr int WROWS = 9 * 32 * NCB;       // weight rows of a chunk in LDS
    constexpr int WUNITS = WROWS * 4;         // 16-byte units (hi 2 + lo 2 per row)
    constexpr int WPT = (WUNITS + 255) / 256; // 9 / 5 per thread
    extern __shared__ __attribute__((aligned(16))) char smem3h[];
    half_t *pbuf = reinterpret_cast<half_t *>(smem3h);        // [2][NPOS][ROWH]
    half_t *wbuf = pbuf + 2 * PATCH_H;                        // [9][32 NCB][ROWH]
    __shared__ float sbias[3][64];                            // the levels' biases (read in every tile's epilogue: from global memory that was an L2 round trip per tile)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 192) {
        const int lv = tid >> 6, co = tid & 63;
        sbias[lv][co] = co < mm.p[lv].Cout ? mm.p[lv].b[co] : 0.f;
    }
    const int r = lane & 31, hi = lane >> 5;
    const int co0 = blockIdx.y * (32 * NCB);  // grid.y > 1: the output channels split over workgroups (a few frames per call: twice the workgroups, half the weights each)

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

    // ---- staging: patch chunk (position, channel quad) items -> registers (raw fp32), split + stored later
    floatx4 pst[PPT];
    unsigned pok = 0;
    auto fetch_patch = [&](int t, int c) {
        const TileG g = tile_g(mm, t);
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
    // weights: host-packed [chunk][9][64][32 halves = hi16|lo16] (rows >= Cout are zero); the kernel stages rows [0, 32 NCB) of every tap
    half8 wst[WPT];
    auto fetch_weights = [&](int lv, int c) {
        const half_t *src = mm.p[lv].wh + (long)c * (9 * 64 * 32) + co0 * 32;  // (this workgroup's first output channel: a uniform offset, rows are 32 halves)
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            int u = tid + i * 256;
            if (u >= WUNITS) u = 0;  // (only the last, partial round of the NCB = 1 shape: a clamped, unused load)
            const int row = u >> 2, part = u & 3, tap = row / (32 * NCB), co = row - tap * (32 * NCB);
            wst[i] = *reinterpret_cast<const half8 *>(src + ((long)(tap * 64 + co) * 4 + part) * 8);
        }
    };
    auto store_weights = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int u = tid + i * 256;
            const int row = u >> 2, part = u & 3;  // row = tap*(32 NCB) + cout; part: hi0, hi1, lo0, lo1 (8 halves each)
            if (u < WUNITS) *reinterpret_cast<half8 *>(wbuf + row * ROWH + part * 8) = wst[i];
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

    // lane geometry: this wave's pixel block covers tile rows 2*wave + (r >> 4), column r & 15
    const int bbase = ((2 * wave + (r >> 4)) * PS + (r & 15)) * ROWH + 8 * hi;
    const int abase = r * ROWH + 8 * hi;

    floatx16 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;

    // fragments of one tap: [cb] a_hi, a_lo; b_hi, b_lo - two sets, tap t + 1 lands while tap t multiplies
    constexpr int NF = 2 * NCB + 2;
    half8 fr[2][NF];

    for (;;) {
        const bool v1 = s1.t < total;
        // (one chunk per tile: the weights in LDS are the next step's too unless the pyramid level changes)
        const bool new_w = NCH > 1 || (v1 && lv_of(s1.t) != lv_of(s0.t));
        if (v1) {
            if (!(FRT_C3H_ABL & 2)) fetch_patch(s1.t, s1.c);
            if (new_w && !(FRT_C3H_ABL & 4)) fetch_weights(lv_of(s1.t), s1.c);
        }
        const half_t *pb_ = pbuf + s0.pc * PATCH_H;
        auto read_frag = [&](int tap, int i, half8 &dst) {   // fragment i of a tap: 0 .. 2 NCB - 1 weights (cb, hi | lo), then b_hi, b_lo
            const int kh = tap / 3, kw = tap - kh * 3;
            if (i < 2 * NCB)
                dst = *reinterpret_cast<const half8 *>(wbuf + (tap * 32 * NCB + (i >> 1) * 32) * ROWH + abase + 16 * (i & 1));
            else
                dst = *reinterpret_cast<const half8 *>(pb_ + bbase + (kh * PS + kw) * ROWH + 16 * (i - 2 * NCB));
        };
        if (!(FRT_C3H_ABL & 1)) {
#pragma unroll
        for (int i = 0; i < NF; ++i) read_frag(0, i, fr[0][i]);
        __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tap = 0; tap < ((FRT_C3H_ABL & 1) ? 0 : 9); ++tap) {
            half8(&f)[NF] = fr[tap & 1];
            half8(&n)[NF] = fr[(tap + 1) & 1];
            const half8 bh = f[2 * NCB], bl = f[2 * NCB + 1];
            // MFMA m of the tap: product (m / NCB): hi*hi, hi*lo, lo*hi; cout block m % NCB.  Behind each one, one fragment of tap + 1.
            // Fragment i of the NEXT set may only be overwritten once this tap's MFMAs that read slot i of THIS set ... are a different
            // register set: no hazard; the set being refilled was consumed by tap - 1.
#pragma unroll
            for (int m = 0; m < 3 * NCB; ++m) {
                const int prod = m / NCB, cb = m - prod * NCB;
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(prod == 2 ? f[2 * cb + 1] : f[2 * cb], prod == 1 ? bl : bh, acc[cb], 0, 0, 0);
                // (requested in the order the next tap's MFMAs need them: a_hi(0), b_hi, a_hi(1), b_lo, a_lo(0), a_lo(1))
                constexpr int ord2[6] = {0, 4, 2, 5, 1, 3}, ord1[4] = {0, 2, 3, 1};
                if (tap < 8 && m < NF) {
                    const int i = NCB == 2 ? ord2[m] : ord1[m];
                    read_frag(tap + 1, i, n[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tap < 8 && NCB == 1) {
                read_frag(tap + 1, 1, n[1]);  // (NCB = 1: four fragments, three MFMAs)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (s0.c == NCH - 1) {
            // ---- tile finished: lane (r, hi) owns pixel r of its block and channels cb*32 + (e&3) + 8*(e>>2) + 4*hi
            const TileG g = tile_g(mm, s0.t);
            const Conv3Args &a = mm.p[g.lv];
            const long HoWo = (long)a.Ho * a.Wo;
            const int oy = g.oy0 + 2 * wave + (r >> 4), ox = g.ox0 + (r & 15);
            const bool inside = oy < a.Ho && ox < a.Wo;
            const long pix = inside ? (long)oy * a.Wo + ox : 0;
            // (this workgroup's first output channel co0 enters through uniform quantities only: bases and limits)
            float *o1 = a.out + ((long)g.b * a.out_ctotal + a.out_coff + co0) * HoWo + pix;
            float *o2 = a.out2 ? a.out2 + ((long)g.b * a.out2_ctotal + a.out2_coff - a.split + co0) * HoWo + pix : o1;
            const float *sb = &sbias[g.lv][co0];
            const int cout_l = a.Cout - co0, split_l = a.split - co0;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    float v = acc[cb][e] + sb[co];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (inside && co < cout_l && (!(FRT_C3H_ABL & 8) || v == 12345.678f)) (co < split_l ? o1 : o2)[co * HoWo] = v;
                    acc[cb][e] = 0.f;
                }
        }
        if (!v1) break;
        if (!(FRT_C3H_ABL & 2)) store_patch(pbuf + s1.pc * PATCH_H);  // the other patch buffer: its readers finished a step ago
        if (new_w) {
            if (!(FRT_C3H_ABL & 16)) __syncthreads();                  // everybody is done with this step's weights
            if (!(FRT_C3H_ABL & 4)) store_weights();
        }
        if (!(FRT_C3H_ABL & 16)) __syncthreads();
        s0 = s1;
        s1 = advance(s1);
    }
}

template <int NCH, int NCB>
void launch_split(const Conv3H &mm, int total, hipStream_t s, int cout_groups = 1) {
    const size_t lds = (size_t)(2 * PATCH_H + 9 * 32 * NCB * ROWH) * sizeof(half_t);  // 74.9 KB / 51.8 KB: two workgroups per CU
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_split_kernel<NCH, NCB>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    int grid = NCB == 1 ? 768 : 512;  // (51.8 KB of LDS and 132 - 144 registers: three workgroups per CU; 74.9 KB: two)
    if (grid > total) grid = total;
    hipLaunchKernelGGL((conv3x3_split_kernel<NCH, NCB>), dim3(grid, cout_groups), dim3(256), lds, s, mm);
}

}  // namespace

// Up to 3 same-shaped stride-1 problems with Cin == 64 or 16 in one launch.  false: shape not covered / split weights absent.
bool launch_conv3x3_split(const Conv3Args *a, int n, hipStream_t s) {
    static const bool off = frt_tuning_env("FRT_DET_SPLIT") && frt_tuning_env("FRT_DET_SPLIT")[0] == '0';
    if (off || n < 1 || n > 3) return false;
    Conv3H mm;
    int base = 0;
    for (int i = 0; i < 3; ++i) {
        const Conv3Args &p = a[i < n ? i : 0];
        if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo || (p.Cin != 64 && p.Cin != 16) || p.Cin != a[0].Cin || p.Cout != a[0].Cout || p.Cout > 64 || p.Cout < 16 || !p.wh)
            return false;
        mm.p[i] = p;
        if (!mm.p[i].out2) mm.p[i].split = p.Cout;
        mm.tiles_x[i] = (p.Wo + TS - 1) / TS;
        mm.tiles_y[i] = (p.Ho + TH - 1) / TH;
        mm.base[i] = base;
        if (i < n) base += p.B * mm.tiles_x[i] * mm.tiles_y[i];
    }
    for (int i = n; i < 4; ++i) mm.base[i] = base;
    if (base < 1) return true;
    const bool one = a[0].Cout <= 32;
    if (a[0].Cin == 64) {
        // a few frames per call: fewer tiles than CUs - the 64 output channels as two workgroups of 32 (same accumulation order per channel:
        // bit-identical; every workgroup stages half the weights and the whole patch)
        static const int small_tiles = frt_tuning_env("FRT_C3H_SPLIT_TILES") ? atoi(frt_tuning_env("FRT_C3H_SPLIT_TILES")) : 384;
        if (one) launch_split<4, 1>(mm, base, s);
        else if (base <= small_tiles) launch_split<4, 1>(mm, base, s, 2);
        else launch_split<4, 2>(mm, base, s);
    } else {
        if (one) launch_split<1, 1>(mm, base, s);
        else launch_split<1, 2>(mm, base, s);
    }
    return true;
}
