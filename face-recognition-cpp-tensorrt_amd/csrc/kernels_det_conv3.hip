// RetinaFace dense 3x3 convs (stride 1, pad 1: FPN merges, SSH branches) on the fp32 matrix cores.
//
// Arithmetic spec: /root/reference/conversion/retina/models/net.py:9-17 (conv_bn / conv_bn_no_relu), :40-66 (SSH), :88-96 (FPN
// merges).  BN folded on the host; true-fp32 v_mfma_f32_32x32x2_f32, so results stay inside the fp32 tolerances of
// tests/test_gpu_detector.py.
//
// Structure (each point is a measured failure of the previous version):
//   * persistent workgroups walk a contiguous range of 8x16-pixel tiles; the work is a flat sequence of steps
//     (tile, KC-channel chunk, tap).  A first version staged the whole fp32 halo patch, then computed: patch loads (65 us),
//     MFMAs (105 us) and the rest (80 us) simply added up to 250 us on the 64->64 80x80 merge because both resident
//     workgroups of a CU were always in the same phase.  Now the halo patch of the NEXT chunk (or next tile) is fetched
//     global -> registers during the 9 taps of the current one and written to the other LDS buffer at the end.
//   * every wave streaming its own weights L2 -> registers was L2-bound (3.4 TB/s of weight traffic): weights go through a
//     triple-buffered LDS tile shared by the 4 waves, fetched two taps ahead from a host-packed [tap][chunk][Cout][KC] copy,
//     so that a tap's LDS operand reads can be issued one tap early, behind the previous tap's MFMAs (PMC: 67 % of the wave
//     time was s_waitcnt when the reads sat in front of their own MFMAs).
//   * LDS layouts are channel-fastest ([position][KC] and [cout][KC], rows padded by 16 B): a lane's 16 (KC=32) or 8 (KC=16)
//     k-values are contiguous, so operands arrive as ds_read_b128 (12 reads per tap instead of 48 ds_read_b32).  k-step j of a
//     chunk multiplies channels j (lanes 0-31) and j + KC/2 (lanes 32-63) - any pairing is valid as long as A and B agree.
//   * optional second output (channel split): SSH conv3X3 (64->32) and conv5X5_1 (64->16) read the same input and are one
//     64->48 launch writing to two tensors.
//   * up to 3 pyramid levels per launch (flat tile index).
#include <cstdlib>

#include "frt_kernels.h"

namespace {

struct Conv3Mfma {
    Conv3Args p[3];
    int tiles_x[3], tiles_y[3], base[4];  // base[l] = first flat tile of level l; base[3] = total
};

struct TileGeom {
    int lv, b, oy0, ox0;
};

__device__ __forceinline__ TileGeom tile_geom(const Conv3Mfma &mm, int t) {
    TileGeom g;
    g.lv = t >= mm.base[2] ? 2 : (t >= mm.base[1] ? 1 : 0);
    const int tx_n = mm.tiles_x[g.lv], per = tx_n * mm.tiles_y[g.lv];
    const int local = t - mm.base[g.lv];
    g.b = local / per;
    const int rem = local - g.b * per;
    const int tyi = rem / tx_n;
    g.oy0 = tyi * 8;
    g.ox0 = (rem - tyi * tx_n) * 16;
    return g;
}

template <int CB, int KC>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(Conv3Mfma mm, int tiles_per_wg) {
    constexpr int KH = KC / 2;            // channels per lane half = k-steps per tap-chunk
    constexpr int PST = KC + 4;           // floats per patch position (row + 16 B pad: conflict-free ds_read_b128)
    constexpr int NPOS = 180;             // 10 x 18 halo positions
    constexpr int PITEMS = NPOS * (KC / 4);
    constexpr int PPT = (PITEMS + 255) / 256;
    constexpr int WUNITS = CB * 32 * (KC / 4);  // float4 units of one weight chunk
    constexpr int WPT = (WUNITS + 255) / 256;
    constexpr int NV = KH / 4;            // float4 reads per operand per lane

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *pbuf = smem;                                   // [2][NPOS][PST]
    float *wbuf = smem + 2 * NPOS * PST;                  // [3][CB*32][PST]

    // ---- this workgroup's tiles: wid, wid + nwg, wid + 2 nwg, ...  (wid is XCD-contiguous, so every XCD's L2 sees runs of
    //      nwg/8 neighbouring tiles; the strided walk keeps the per-workgroup tile counts within one of each other)
    const int nwg = gridDim.x;
    const int bq = nwg >> 3, brem = nwg & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
    const int total = mm.base[3];
    const int t_lo = wid, t_hi = total;
    if (t_lo >= t_hi) return;

    const int Cin = mm.p[0].Cin, ncc = Cin / KC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hi = lane >> 5;
    const int ty = wave * 2 + (r >> 4), tx = r & 15;

    // ---- staging roles
    floatx4 pst[PPT];
    unsigned pst_ok = 0;  // bit i: item i of pst is inside the image (the zeroing happens at store time, see below)
    auto fetch_patch = [&](int t, int cc) {
        pst_ok = 0;
        const TileGeom g = tile_geom(mm, t);
        const Conv3Args &a = mm.p[g.lv];
        const long HW = (long)a.H * a.W;
        const float *inb = a.in + ((long)g.b * a.Cin + cc * KC) * HW;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int it = threadIdx.x + i * 256;
            const int q = it / NPOS, pos = it - q * NPOS;       // q: channel quad, pos fastest -> coalesced rows
            const int py = pos / 18, px = pos - py * 18;
            const int iy = g.oy0 - 1 + py, ix = g.ox0 - 1 + px;
            const bool ok = it < PITEMS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            // loads are UNCONDITIONAL from a clamped in-bounds address, zeroed afterwards: `ok ? load : 0` compiles to
            // exec-masked branches with a vmcnt wait per group, which serialised the 24 gathers (45 us per tile, measured)
            const int qq = it < PITEMS ? q : 0;
            const float *src = inb + (long)(4 * qq) * HW + (ok ? iy * a.W + ix : 0);
            // ... and the select is deferred to store_patch: a select here would need the data, i.e. a vmcnt(0) wait right
            // after the loads, in front of the tap's MFMAs (the fetch sits in a branch, the compiler cannot sink it)
            pst[i][0] = src[0];
            pst[i][1] = src[HW];
            pst[i][2] = src[2 * HW];
            pst[i][3] = src[3 * HW];
            pst_ok |= ok ? (1u << i) : 0u;
        }
    };
    auto store_patch = [&](float *dst) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int it = threadIdx.x + i * 256;
            const int q = it / NPOS, pos = it - q * NPOS;
            const bool ok = (pst_ok >> i) & 1u;
            floatx4 v;
            v[0] = ok ? pst[i][0] : 0.f;
            v[1] = ok ? pst[i][1] : 0.f;
            v[2] = ok ? pst[i][2] : 0.f;
            v[3] = ok ? pst[i][3] : 0.f;
            if (it < PITEMS) *reinterpret_cast<floatx4 *>(dst + pos * PST + 4 * q) = v;
        }
    };
    floatx4 wst[WPT];
    auto fetch_weights = [&](int lv, int cc, int tap) {
        const float *src = mm.p[lv].wm + ((long)(tap * ncc + cc) * (CB * 32)) * KC;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int u = threadIdx.x + i * 256;
            wst[i] = *reinterpret_cast<const floatx4 *>(src + (long)(u < WUNITS ? u : 0) * 4);  // unconditional (see fetch_patch)
        }
    };
    auto store_weights = [&](float *dst) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int u = threadIdx.x + i * 256;
            const int row = u / (KC / 4), c4 = u - row * (KC / 4);
            if (u < WUNITS) *reinterpret_cast<floatx4 *>(dst + row * PST + 4 * c4) = wst[i];
        }
    };

    // ---- the work of this workgroup as a flat sequence of steps (round k -> tile, chunk cc, tap); pc = patch buffer of the chunk
    struct Step {
        int k, t, cc, tap, pc;
    };
    const int k_full = total / nwg, rem_tiles = total - k_full * nwg;
    auto tile_of = [&](int k) {
        // full rounds walk XCD-contiguous ids; the last partial round is dealt out in blockIdx order, i.e. round-robin over the
        // XCDs (dealt by wid, all remainder tiles land on XCD 0 and its CUs finish a whole tile after everybody else)
        if (k < k_full) return wid + k * nwg;
        return (k == k_full && (int)blockIdx.x < rem_tiles) ? k_full * nwg + (int)blockIdx.x : total;
    };
    auto advance = [&](Step s) {
        if (++s.tap == 9) {
            s.tap = 0;
            s.pc ^= 1;
            if (++s.cc == ncc) {
                s.cc = 0;
                s.t = tile_of(++s.k);
            }
        }
        return s;
    };
    auto lv_of = [&](int t) { return t >= mm.base[2] ? 2 : (t >= mm.base[1] ? 1 : 0); };
    const int lane_pos = (ty * 18 + tx) * PST + hi * KH;
    auto read_operands = [&](const Step &st, int wslot, floatx4 (&bv)[NV], floatx4 (&av)[CB][NV]) {
        const int kh = st.tap / 3, kw = st.tap - kh * 3;
        const float *bp = pbuf + st.pc * NPOS * PST + lane_pos + (kh * 18 + kw) * PST;
        const float *ap = wbuf + wslot * (CB * 32) * PST + r * PST + hi * KH;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            bv[v] = *reinterpret_cast<const floatx4 *>(bp + 4 * v);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) av[cb][v] = *reinterpret_cast<const floatx4 *>(ap + cb * 32 * PST + 4 * v);
        }
    };

    Step s0{0, tile_of(0), 0, 0, 0};
    if (s0.t >= total) return;
    Step s1 = advance(s0), s2 = advance(s1);

    // ---- prologue: patch of the first chunk, weights of steps 0 and 1
    fetch_patch(s0.t, 0);
    fetch_weights(lv_of(s0.t), 0, 0);
    store_patch(pbuf);
    store_weights(wbuf);
    fetch_weights(lv_of(s1.t), s1.cc, s1.tap);  // a tile has >= 9 steps: s1 is always valid
    store_weights(wbuf + (CB * 32) * PST);
    __syncthreads();

    floatx16 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
    floatx4 bvA[NV], avA[CB][NV], bvB[NV], avB[CB][NV];
    read_operands(s0, 0, bvA, avA);
    int w0 = 0;            // weight slot of s0 (s1: w0+1, s2: w0+2, mod 3)
    bool has_np = false;   // a next patch chunk is in flight (fetched at tap 0, stored at tap 7)

    // One step.  Operands of the CURRENT step are already in registers; this step's LDS reads fetch the NEXT step's operands and
    // its global loads the weights of the step after that, both behind the 16*CB MFMAs.  Returns true after the last step.
    auto step = [&](floatx4 (&bv)[NV], floatx4 (&av)[CB][NV], floatx4 (&bvn)[NV], floatx4 (&avn)[CB][NV]) -> bool {
        const bool v1 = s1.t < total, v2 = s2.t < total;
        if (s0.tap == 0) {
            Step c = s0;
            c.tap = 8;
            c = advance(c);
            has_np = c.t < total;
            if (has_np) fetch_patch(c.t, c.cc);
        }
        // The common part is ONE basic block with a pinned order - loads | MFMA group 0 | LDS reads of the next operands | MFMA
        // groups | LDS writes of the weights | last MFMA group - so that everything that is not an MFMA issues in the shadow
        // of this wave's own MFMAs.  (With the non-MFMA work behind all MFMAs, the two waves of a SIMD - round-robin on the
        // matrix pipe - fall into lock-step and do their non-MFMA work at the same time: pipe 65 % busy, measured.)  Invalid
        // next steps are clamped to this step's own coordinates: the loads stay in bounds, the results are never used.
        const Step f2 = v2 ? s2 : s0, f1 = v1 ? s1 : s0;
        const int w1 = w0 == 2 ? 0 : w0 + 1, w2 = w1 == 2 ? 0 : w1 + 1;
        fetch_weights(lv_of(f2.t), f2.cc, f2.tap);
        auto mfma_group = [&](int v) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][v][j], bv[v][j], acc[cb], 0, 0, 0);
        };
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0);
        __builtin_amdgcn_sched_barrier(0);
        read_operands(f1, v1 ? w1 : w0, bvn, avn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 1; v < NV - 1; ++v) mfma_group(v);
        __builtin_amdgcn_sched_barrier(0);
        store_weights(wbuf + w2 * (CB * 32) * PST);
        __builtin_amdgcn_sched_barrier(0);
        if (NV > 1) mfma_group(NV - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (s0.tap == 7 && has_np) store_patch(pbuf + (s0.pc ^ 1) * NPOS * PST);
        if (s0.tap == 8 && s0.cc == ncc - 1) {
            // ---- tile finished: lane (r, hi) owns pixel r and channels cb*32 + (e&3) + 8*(e>>2) + 4*hi
            const TileGeom g = tile_geom(mm, s0.t);
            const Conv3Args &a = mm.p[g.lv];
            const int oy = g.oy0 + ty, ox = g.ox0 + tx;
            const bool inside = oy < a.Ho && ox < a.Wo;
            const long HoWo = (long)a.Ho * a.Wo;
            const long pix = inside ? (long)oy * a.Wo + ox : 0;
            float *o1 = a.out + ((long)g.b * a.out_ctotal + a.out_coff) * HoWo + pix;
            float *o2 = a.out2 ? a.out2 + ((long)g.b * a.out2_ctotal + a.out2_coff - a.split) * HoWo + pix : o1;
            float bq[CB][16];  // (in front of the first store: load / wait / store chains otherwise - the output may alias the biases for all the compiler knows)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    bq[cb][e] = a.b[co < a.Cout ? co : 0];
                }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    float v = acc[cb][e] + bq[cb][e];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (inside && co < a.Cout) (co < a.split ? o1 : o2)[co * HoWo] = v;
                    acc[cb][e] = 0.f;
                }
        }
        if (!v1) return true;
        __syncthreads();
        s0 = s1;
        s1 = s2;
        s2 = advance(s2);
        w0 = w1;
        return false;
    };
    for (;;) {
        if (step(bvA, avA, bvB, avB)) break;
        if (step(bvB, avB, bvA, avA)) break;
    }
}

}  // namespace

// Up to 3 same-shaped (Cin, Cout) stride-1 problems (the pyramid levels) in one launch.  false: shape not covered.
bool launch_conv3x3_mfma(const Conv3Args *a, int n, hipStream_t s) {
    if (n < 1 || n > 3) return false;
    Conv3Mfma mm;
    int base = 0;
    for (int i = 0; i < 3; ++i) {
        const Conv3Args &p = a[i < n ? i : 0];
        if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo || p.Cin != a[0].Cin || p.Cout != a[0].Cout || !p.wm) return false;
        mm.p[i] = p;
        if (!mm.p[i].out2) mm.p[i].split = p.Cout;
        mm.tiles_x[i] = (p.Wo + 15) / 16;
        mm.tiles_y[i] = (p.Ho + 7) / 8;
        mm.base[i] = base;
        if (i < n) base += p.B * mm.tiles_x[i] * mm.tiles_y[i];
    }
    for (int i = n; i < 4; ++i) mm.base[i] = base;  // unused levels start past the end: never selected
    const int cin = a[0].Cin, cout = a[0].Cout;
    const int kc = cin == 16 ? 16 : 32;
    // 16 -> 16 convs: 8 MFMAs per step cannot hide a step's fixed costs (measured 51 us against 31 us for the scalar kernel)
    // (With the few tiles of ONE frame it is the other way round - 10 us against 23, profiles/r03/r03q_det_b1_switches.txt - but switching by
    //  batch size would give a frame other last bits in its scores alone than in a batch: tests/test_gpu_detector.py holds the detector to
    //  bit-identical outputs whatever the batch.)
    if (cin == 16 && !frt_tuning_env("FRT_C3_FORCE16")) return false;
    if (cout > 64 || cout < 16 || cin % kc || a[0].wm_kc != kc) return false;
    const int cb = cout > 32 ? 2 : 1;
    if (a[0].wm_cpad != cb * 32) return false;
    static const int wg_per_cu = [] {
        const char *e = frt_tuning_env("FRT_C3_WG_PER_CU");
        return e ? atoi(e) : 2;
    }();
    int grid = 256 * wg_per_cu;
    if (grid > base) grid = base;
    const int tiles_per_wg = (base + grid - 1) / grid;
    const size_t lds = (size_t)(2 * 180 + 3 * cb * 32) * (kc + 4) * sizeof(float);
    if (kc == 16) {
        if (cb == 2) hipLaunchKernelGGL((conv3x3_mfma_kernel<2, 16>), dim3(grid), dim3(256), lds, s, mm, tiles_per_wg);
        else hipLaunchKernelGGL((conv3x3_mfma_kernel<1, 16>), dim3(grid), dim3(256), lds, s, mm, tiles_per_wg);
    } else {
        if (cb == 2) hipLaunchKernelGGL((conv3x3_mfma_kernel<2, 32>), dim3(grid), dim3(256), lds, s, mm, tiles_per_wg);
        else hipLaunchKernelGGL((conv3x3_mfma_kernel<1, 32>), dim3(grid), dim3(256), lds, s, mm, tiles_per_wg);
    }
    return true;
}
