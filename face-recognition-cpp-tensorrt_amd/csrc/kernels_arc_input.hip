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

    // per-lane tap table: slot s of k-step ks is tap k = ks*16 + 8*hi + s
    int koff[16];
    unsigned need[16];  // bit0 needs row above, bit1 row below, bit2 column left, bit3 column right, bit4 = constant one, bit5 = zero
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = (s >> 3) * 16 + 8 * hi + (s & 7);
        const int ci = k / 9, rem = k - ci * 9, kh = rem / 3, kw = rem - kh * 3;
        koff[s] = k < 27 ? ci * IHW + (kh - 1) * IW + (kw - 1) : 0;
        need[s] = k < 27 ? ((kh == 0 ? 1u : 0u) | (kh == 2 ? 2u : 0u) | (kw == 0 ? 4u : 0u) | (kw == 2 ? 8u : 0u)) : (k == 27 ? 16u : 32u);
    }

    half_t *trz = tr[wave][0], *try_ = tr[wave][1];
    const int total_waves = gridDim.x * 4;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += total_waves) {
        const long g = (long)tile * 32 + r;
        const int f = (int)(g / IHW), pix = (int)(g - (long)f * IHW);
        const int oh = pix / IW, ow = pix - oh * IW;
        const unsigned edge = (oh == 0 ? 1u : 0u) | (oh == IH - 1 ? 2u : 0u) | (ow == 0 ? 4u : 0u) | (ow == IW - 1 ? 8u : 0u);
        const float *xb = a.x + (long)f * 3 * IHW + pix;
        float xv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bool live = (need[s] & (edge | 48u)) == 0;  // unconditional load from a clamped address, masked below
            xv[s] = xb[live ? koff[s] : 0];
        }
        half8 bf[2];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bool live = (need[s] & (edge | 48u)) == 0;
            const float v = live ? xv[s] : ((need[s] & 16u) ? 1.f : 0.f);
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
        // lane (r, hi) owns pixel r and channels cb*32 + 8*q + 4*hi + (0..3), q = 0..3
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
        const long p0 = (long)tile * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int seg = lane + 64 * i, px = seg >> 3, sg = seg & 7;
            const half8 zv = *reinterpret_cast<const half8 *>(trz + px * TRS + sg * 8);
            *reinterpret_cast<half8 *>(a.z + (p0 + px) * 64 + sg * 8) = zv;
            const long gp = p0 + px;
            const int ff = (int)(gp / IHW), pp = (int)(gp - (long)ff * IHW);
            const int yh = pp / IW, yw = pp - yh * IW;
            if (((yh | yw) & 1) == 0) {
                const half8 yv = *reinterpret_cast<const half8 *>(try_ + px * TRS + sg * 8);
                *reinterpret_cast<half8 *>(a.y + (((long)ff * (IH / 2) + (yh >> 1)) * (IW / 2) + (yw >> 1)) * 64 + sg * 8) = yv;
            }
        }
    }
}

}  // namespace

// false: shape not covered (the caller uses the scalar kernel)
bool launch_arc_input_mfma(const ArcInputArgs &a, hipStream_t s) {
    if (a.H != IH || a.W != IW || !a.wh || ((long)a.F * IHW) % 32) return false;
    const int n_tiles = (int)((long)a.F * IHW / 32);
    int grid = 512;  // persistent: 2 workgroups per CU
    if (grid * 4 > n_tiles) grid = (n_tiles + 3) / 4;
    hipLaunchKernelGGL(arc_input_mfma_kernel, dim3(grid), dim3(256), 0, s, a, n_tiles);
    return true;
}
