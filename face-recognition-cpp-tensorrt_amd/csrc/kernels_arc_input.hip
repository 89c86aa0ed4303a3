// ArcFace input layer on the matrix cores: conv3x3 3->64 (pad 1) + BN + PReLU, plus unit 0's leading BN as a second output.
//
// Arithmetic spec: /root/reference/conversion/arcface/model_irse.py:139-141 (input_layer) and :60-61 (the BatchNorm2d that opens
// bottleneck_IR's res_layer).  fp16 operands, fp32 accumulation - the precision of every other conv of the recogniser.
//
// The scalar version (arc_input_kernel, kept for other shapes) is LDS-bound: 54 ds_read_b128 of weights per thread for 216 FMAs
// (306 us at 128 faces against a ~55 us HBM floor).  Here one wave turns 32 consecutive pixels into a [64 cout x 32 px] tile with
// FOUR v_mfma_f32_32x32x16_f16:
//   * K = 27 taps, padded to 32.  Slot 27 carries a constant 1.0 whose weight is the folded BN bias, slots 28-31 are zero; the
//     BN scale is folded into the fp16 weights.  The accumulator leaves the MFMA already batch-normalised.
//   * B operand (im2col): lane (pixel r, half hi) gathers its 16 taps straight from the planar fp32 crop (lanes = consecutive
//     pixels: 128-byte segments), per-lane tap offsets and border masks are computed once and kept in registers.
//   * A operand: the whole weight matrix is 4 x half8 per lane, loaded once per (persistent) wave.
//   * epilogue: PReLU, second BN, fp16; the lane-owns-a-pixel accumulator layout is turned into NHWC rows through a private
//     LDS tile per wave (ds_write_b64 / ds_read_b128, no barrier: LDS is in-order per wave), so z leaves as 4 KB contiguous
//     per wave-tile in 16-byte stores.  y (only read as unit 0's MaxPool(1,2) shortcut) is written at even positions only.
#include "frt_kernels.h"

namespace {

constexpr int IH = 112, IW = 112, IHW = IH * IW;
constexpr int TRS = 72;  // halves per pixel row of the transpose tile (64 + 8 pad: conflict-free 16-byte row reads)

__global__ __launch_bounds__(256) void arc_input_mfma_kernel(ArcInputArgs a, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float prm[3 * 64];              // slope, s1, b1
    __shared__ __attribute__((aligned(16))) half_t tr[4][2][32 * TRS];      // per wave: z tile, y tile
    if (threadIdx.x < 64) {
        prm[threadIdx.x] = a.slope[threadIdx.x];
        prm[64 + threadIdx.x] = a.s1[threadIdx.x];
        prm[128 + threadIdx.x] = a.b1[threadIdx.x];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hi = lane >> 5;

    // weights: wh [64 cout][32 k] fp16; lane (cout r of block cb, half hi) holds k = ks*16 + 8*hi .. +7
    half8 wa[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wa[cb][ks] = *reinterpret_cast<const half8 *>(a.wh + (cb * 32 + r) * 32 + ks * 16 + 8 * hi);

    // per-lane tap table: slot s of k-step ks is tap k = ks*16 + 8*hi + s.  Byte offsets (32-bit; the launcher bounds the tensor below 2 GB).
    int koff[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = (s >> 3) * 16 + 8 * hi + (s & 7);
        const int ci = k / 9, rem = k - ci * 9, kh = rem / 3, kw = rem - kh * 3;
        koff[s] = k < 27 ? (ci * IHW + (kh - 1) * IW + (kw - 1)) * 4 : 0;
    }
    const float one_hi = hi ? 1.f : 0.f;  // slot 11 of the upper half-wave is k = 27: the constant 1.0 that carries the folded BN bias
    const int max_off = a.F * 3 * IHW * 4 - 4;

    half_t *trz = tr[wave][0], *try_ = tr[wave][1];
    const int total_waves = gridDim.x * 4;
    // The tile loop is software-pipelined one tile deep: the 16 tap loads of tile t + stride are issued before tile t's MFMAs / epilogue / stores, so a
    // wave always has a tile's loads in flight (one tile at a time made the launch latency-structured: 24 tiles per wave x ~ 3.4 us).
    // Border handling without per-slot vector compares: a tile's four edge classes are wave masks in scalar registers (4 v_cmp per tile); which classes
    // kill slot s is known at compile time per half-wave (k = ks*16 + s for lanes 0-31, + 8 for lanes 32-63), so a slot's kill mask is a few SCALAR
    // and / or of those and one v_cndmask applies it.  Loads are unconditional from an address clamped into the tensor (v_med3): a dead tap reads some
    // neighbouring element and is replaced afterwards.
    struct Edge { unsigned long long t, b, l, r; };
    float xv[16];
    Edge edge = {0, 0, 0, 0};
    auto fetch = [&](int tile, float (&dst)[16], Edge &e_out) {
        const unsigned g = (unsigned)tile * 32u + (unsigned)r;  // (32-bit; divisions by constants)
        const unsigned f = g / (unsigned)IHW, pix = g - f * (unsigned)IHW;
        const unsigned oh = pix / (unsigned)IW, ow = pix - oh * (unsigned)IW;
        e_out.t = __builtin_amdgcn_ballot_w64(oh == 0);
        e_out.b = __builtin_amdgcn_ballot_w64(oh == IH - 1);
        e_out.l = __builtin_amdgcn_ballot_w64(ow == 0);
        e_out.r = __builtin_amdgcn_ballot_w64(ow == IW - 1);
        const int base = (int)((f * 3u * (unsigned)IHW + pix) * 4u);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const unsigned off = min((unsigned)(base + koff[s]), (unsigned)max_off);  // (a negative offset wraps to a large one: one v_min_u32 clamps both ends)
            dst[s] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.x) + off);
        }
    };
    auto kill_of = [](int k, const Edge &e) -> unsigned long long {  // (k is a literal after unrolling)
        if (k >= 27) return ~0ull;
        const int rem = k % 9, kh = rem / 3, kw = rem % 3;
        return (kh == 0 ? e.t : 0ull) | (kh == 2 ? e.b : 0ull) | (kw == 0 ? e.l : 0ull) | (kw == 2 ? e.r : 0ull);
    };
    int tile = blockIdx.x * 4 + wave;
    if (tile < n_tiles) fetch(tile, xv, edge);
    for (; tile < n_tiles; tile += total_waves) {
        float xn[16];
        Edge edge_n = {0, 0, 0, 0};
        const int tnext = tile + total_waves;
        if (tnext < n_tiles) fetch(tnext, xn, edge_n);
        half8 bf[2];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k0 = (s >> 3) * 16 + (s & 7);
            const unsigned long long kill = (kill_of(k0, edge) & 0x00000000ffffffffull) | (kill_of(k0 + 8, edge) & 0xffffffff00000000ull);
            const float dead = s == 11 ? one_hi : 0.f;
            const float v = __builtin_amdgcn_inverse_ballot_w64(kill) ? dead : xv[s];
            bf[s >> 3][s & 7] = (half_t)v;
        }
        floatx16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cb][0], bf[0], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cb][1], bf[1], acc[cb], 0, 0, 0);
        }
        // lane (r, hi) owns pixel r and channels cb*32 + 8*q + 4*hi + (0..3), q = 0..3.  (The per-channel parameters below are loop-invariant LDS reads the
        // compiler keeps in 96 registers; re-reading them per tile frees a wave per SIMD and measures SLOWER: 86 - 88 us against 71.)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = cb * 32 + 8 * q + 4 * hi;
                const floatx4 sl = *reinterpret_cast<const floatx4 *>(prm + c0);
                const floatx4 s1 = *reinterpret_cast<const floatx4 *>(prm + 64 + c0);
                const floatx4 b1 = *reinterpret_cast<const floatx4 *>(prm + 128 + c0);
                half4 y4, z4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[cb][4 * q + j];
                    v = v > 0.f ? v : v * sl[j];
                    y4[j] = (half_t)v;
                    z4[j] = (half_t)(v * s1[j] + b1[j]);
                }
                *reinterpret_cast<half4 *>(trz + r * TRS + c0) = z4;
                *reinterpret_cast<half4 *>(try_ + r * TRS + c0) = y4;
            }
        // row-major read-back: 256 16-byte segments (32 pixels x 8) per tensor, 4 per lane
        const unsigned p0 = (unsigned)tile * 32u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int seg = lane + 64 * i, px = seg >> 3, sg = seg & 7;
            const half8 zv = *reinterpret_cast<const half8 *>(trz + px * TRS + sg * 8);
            *reinterpret_cast<half8 *>(a.z + (size_t)(p0 + px) * 64 + sg * 8) = zv;
            const unsigned gp = p0 + px;
            const unsigned ff = gp / (unsigned)IHW, pp = gp - ff * (unsigned)IHW;
            const unsigned yh = pp / (unsigned)IW, yw = pp - yh * (unsigned)IW;
            if (((yh | yw) & 1) == 0) {
                const half8 yv = *reinterpret_cast<const half8 *>(try_ + px * TRS + sg * 8);
                *reinterpret_cast<half8 *>(a.y + (size_t)((ff * (IH / 2) + (yh >> 1)) * (IW / 2) + (yw >> 1)) * 64 + sg * 8) = yv;
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) xv[s] = xn[s];
        edge = edge_n;
    }
}

}  // namespace

// false: shape not covered (the caller uses the scalar kernel)
bool launch_arc_input_mfma(const ArcInputArgs &a, hipStream_t s) {
    if (a.H != IH || a.W != IW || !a.wh || ((long)a.F * IHW) % 32 || (long)a.F * 3 * IHW * 4 >= (1L << 31)) return false;
    const int n_tiles = (int)((long)a.F * IHW / 32);
    int grid = 512;  // persistent: 2 workgroups per CU
    if (grid * 4 > n_tiles) grid = (n_tiles + 3) / 4;
    hipLaunchKernelGGL(arc_input_mfma_kernel, dim3(grid), dim3(256), 0, s, a, n_tiles);
    return true;
}
