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
//     double-buffered LDS tile shared by the 4 waves, fetched one tap ahead from a host-packed [tap][chunk][Cout][KC] copy.
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
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(Conv3Mfma mm, int tiles_per_wg) {
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
    float *wbuf = smem + 2 * NPOS * PST;                  // [2][CB*32][PST]

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
    auto fetch_patch = [&](int t, int cc) {
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
            const float v0 = src[0], v1 = src[HW], v2 = src[2 * HW], v3 = src[3 * HW];
            pst[i][0] = ok ? v0 : 0.f;
            pst[i][1] = ok ? v1 : 0.f;
            pst[i][2] = ok ? v2 : 0.f;
            pst[i][3] = ok ? v3 : 0.f;
        }
    };
    auto store_patch = [&](float *dst) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int it = threadIdx.x + i * 256;
            const int q = it / NPOS, pos = it - q * NPOS;
            if (it < PITEMS) *reinterpret_cast<floatx4 *>(dst + pos * PST + 4 * q) = pst[i];
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

    // ---- prologue: first patch chunk and first weight chunk
    {
        const TileGeom g0 = tile_geom(mm, t_lo);
        fetch_patch(t_lo, 0);
        fetch_weights(g0.lv, 0, 0);
        store_patch(pbuf);
        store_weights(wbuf);
    }
    __syncthreads();

    int pcur = 0, wcur = 0;
    for (int t = t_lo; t < t_hi; t += nwg) {
        const TileGeom g = tile_geom(mm, t);
        const Conv3Args &a = mm.p[g.lv];
        floatx16 acc[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;

        for (int cc = 0; cc < ncc; ++cc) {
            // what comes after this (tile, chunk)?
            const bool last_cc = cc + 1 == ncc;
            const bool has_next = !last_cc || t + nwg < t_hi;
            const int nt = last_cc ? t + nwg : t, ncc_i = last_cc ? 0 : cc + 1;
            const int nlv = has_next ? tile_geom(mm, nt).lv : g.lv;
            const float *pb = pbuf + pcur * NPOS * PST + (ty * 18 + tx) * PST + hi * KH;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                if (tap == 0 && has_next) fetch_patch(nt, ncc_i);
                const bool wnext = tap < 8 || has_next;
                if (wnext) {
                    if (tap < 8) fetch_weights(g.lv, cc, tap + 1);
                    else fetch_weights(nlv, ncc_i, 0);
                }
                const int kh = tap / 3, kw = tap - kh * 3;
                const float *bp = pb + (kh * 18 + kw) * PST;
                const float *ap = wbuf + wcur * (CB * 32) * PST + r * PST + hi * KH;
                floatx4 bv[NV], av[CB][NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    bv[v] = *reinterpret_cast<const floatx4 *>(bp + 4 * v);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) av[cb][v] = *reinterpret_cast<const floatx4 *>(ap + cb * 32 * PST + 4 * v);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][v][j], bv[v][j], acc[cb], 0, 0, 0);
                if (wnext) store_weights(wbuf + (wcur ^ 1) * (CB * 32) * PST);
                if (tap == 8 && has_next) store_patch(pbuf + (pcur ^ 1) * NPOS * PST);
                __syncthreads();
                wcur ^= 1;
            }
            pcur ^= 1;
        }

        // ---- epilogue: lane (r, hi) owns pixel r and channels cb*32 + (e&3) + 8*(e>>2) + 4*hi
        const int oy = g.oy0 + ty, ox = g.ox0 + tx;
        if (oy < a.Ho && ox < a.Wo) {
            const long HoWo = (long)a.Ho * a.Wo;
            const long pix = (long)oy * a.Wo + ox;
            float *o1 = a.out + ((long)g.b * a.out_ctotal + a.out_coff) * HoWo + pix;
            float *o2 = a.out2 ? a.out2 + ((long)g.b * a.out2_ctotal + a.out2_coff - a.split) * HoWo + pix : nullptr;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (co < a.Cout) {
                        float v = acc[cb][e] + a.b[co];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (co < a.split) o1[co * HoWo] = v;
                        else o2[co * HoWo] = v;
                    }
                }
        }
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
    if (cout > 64 || cout < 16 || cin % kc || a[0].wm_kc != kc) return false;
    const int cb = cout > 32 ? 2 : 1;
    if (a[0].wm_cpad != cb * 32) return false;
    static const int wg_per_cu = [] {
        const char *e = getenv("FRT_C3_WG_PER_CU");
        return e ? atoi(e) : 2;
    }();
    int grid = 256 * wg_per_cu;
    if (grid > base) grid = base;
    const int tiles_per_wg = (base + grid - 1) / grid;
    const size_t lds = (size_t)(2 * 180 + 2 * cb * 32) * (kc + 4) * sizeof(float);
    if (kc == 16) {
        if (cb == 2) hipLaunchKernelGGL((conv3x3_mfma_kernel<2, 16>), dim3(grid), dim3(256), lds, s, mm, tiles_per_wg);
        else hipLaunchKernelGGL((conv3x3_mfma_kernel<1, 16>), dim3(grid), dim3(256), lds, s, mm, tiles_per_wg);
    } else {
        if (cb == 2) hipLaunchKernelGGL((conv3x3_mfma_kernel<2, 32>), dim3(grid), dim3(256), lds, s, mm, tiles_per_wg);
        else hipLaunchKernelGGL((conv3x3_mfma_kernel<1, 32>), dim3(grid), dim3(256), lds, s, mm, tiles_per_wg);
    }
    return true;
}
