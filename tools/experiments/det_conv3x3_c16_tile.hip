// PARKED EXPERIMENT (round 4, not in the build): bit-identical to conv3x3_kernel<16>, and SLOWER - 75.6 + 49.5 us against 2 x 44.3 us at 32 frames,
// 45 + 26 against 2 x 23 us at ONE frame: a lone wave takes 45 us.  Its 4 608 weights per wave are streamed through the scalar cache in chunks
// of sixteen that each carry only eight v_pk_fma_f32 - the scalar-load latency is exposed 288 times per wave, and 4 200 waves x 18 KB is not
// what the scalar cache is built for.  (det_stem_kernel gets away with the same scheme because a chunk there serves a whole pass of positions.)
// What it would take: the weights in LDS and two or four pixels per lane, so that a broadcast ds_read_b128 carries 16+ fmas.
//
// RetinaFace SSH: the dense 3x3 convs with 16 input channels (conv5X5_2 | conv7X7_2 as one 16 -> 32 launch, conv7x7_3 16 -> 16; net.py:55-66 of
// the reference) on all three pyramid levels, round 4.
//
// The scalar kernel (conv3x3_kernel<16>, kernels_det.hip) is thread = pixel with 144 masked global loads and ~ 250 instructions per input
// channel around its 72 packed fmas: 44 us per launch at 32 frames for 34 MB of tensors, 23 us at 4 frames.  Here a wave owns an 8x8 output
// tile: the 10x10 halo'd input region goes through LDS once (zeros outside the map, so no masks), a lane reads the nine taps of an input channel
// with nine ds_read_b32, and what remains is the arithmetic - 16 x 9 x COUT / 2 v_pk_fma_f32 whose weights are scalar pairs streamed in
// chunks of sixteen (kernels_det_stem.hip's chunk pipeline).
// Same chain per output as the scalar kernel (input channel outer, tap inner): bit-identical.
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "frt_kernels.h"

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));

struct Conv3Tile {
    Conv3Args p[3];
    int tiles_x[3], tiles_y[3], base[4];
};

template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T *uni(const T *p) {
    return reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <class P>
__device__ __forceinline__ void ld8(const P w, int off, floatx2 (&dst)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) dst[c] = floatx2{w[off + 2 * c], w[off + 2 * c + 1]};
}

template <int COUT>
__global__ __launch_bounds__(64) void conv3x3_c16_tile_kernel(Conv3Tile mm) {
    constexpr int SUB = COUT / 16;  // chunks of sixteen output channels per (input channel, tap)
    __shared__ float xin[16][104];  // [channel][10x10 positions]
    const int lane = threadIdx.x;
    const int t = blockIdx.x;
    const int lv = t >= mm.base[2] ? 2 : (t >= mm.base[1] ? 1 : 0);
    const Conv3Args &a = mm.p[lv];
    const int tx_n = mm.tiles_x[lv], per = tx_n * mm.tiles_y[lv];
    const int local = t - mm.base[lv];
    const int b = local / per, rem = local - b * per, ty = rem / tx_n, tx = rem - ty * tx_n;
    const int Y0 = ty * 8, X0 = tx * 8, HW = a.H * a.W;
    const float *inb = a.in + (long)b * 16 * HW;

    // ---- the 10x10 input region -> LDS (zeros outside the map)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int idx = lane + 64 * ps;
        if (idx < 100) {
            const int ry = idx / 10, rx = idx - ry * 10;
            const int gy = Y0 - 1 + ry, gx = X0 - 1 + rx;
            const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const float *src = inb + (ok ? gy * a.W + gx : 0);
            float v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = src[(long)c * HW];
#pragma unroll
            for (int c = 0; c < 16; ++c) xin[c][idx] = ok ? v[c] : 0.f;
        }
    }
    __syncthreads();

    // ---- lane = output pixel; input channel outer, tap inner (the scalar kernel's order per output); a chunk = sixteen output channels of one
    //      (input channel, tap), its scalar weights fetched one chunk ahead
    const int py = lane >> 3, px = lane & 7;
    floatx2 acc[COUT / 2];
#pragma unroll
    for (int c = 0; c < COUT / 2; ++c) acc[c] = floatx2{0.f, 0.f};
    const auto w0 = uni(a.w);  // [16][9][COUT]
#pragma unroll 1
    for (int ci = 0; ci < 16; ++ci) {
        int z;
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));  // (keeps the scalar weight loads inside the loop: see kernels_det_stem.hip)
        const auto w = w0 + (z + ci * 9 * COUT);
        float x[9];
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) x[tt] = xin[ci][(py + tt / 3) * 10 + px + tt % 3];
        floatx2 wc[2][8];
        ld8(w, 0, wc[0]);
        static_for<0, 9 * SUB>([&](auto kc) {
            constexpr int k = decltype(kc)::value, tt = k / SUB, sub = k % SUB;
            if constexpr (k + 1 < 9 * SUB) ld8(w, (k + 1) * 16, wc[(k + 1) & 1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[sub * 8 + c] = __builtin_elementwise_fma(floatx2{x[tt], x[tt]}, wc[k & 1][c], acc[sub * 8 + c]);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    const int oy = Y0 + py, ox = X0 + px;
    if (oy >= a.Ho || ox >= a.Wo) return;
    const int HoWo = a.Ho * a.Wo, p = oy * a.Wo + ox;
    float bv[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) bv[c] = a.b[c];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        float v = acc[c >> 1][c & 1] + bv[c];
        if (a.relu) v = fmaxf(v, 0.f);
        // output channels at or beyond `split` belong to the second output tensor (two convs that share their input, run as one)
        float *ob = (a.out2 && c >= a.split) ? a.out2 + ((long)b * a.out2_ctotal + a.out2_coff + c - a.split) * HoWo + p
                                             : a.out + ((long)b * a.out_ctotal + a.out_coff + c) * HoWo + p;
        *ob = v;
    }
}

}  // namespace

// true: launched.  16 input channels, 16 or 32 output channels, stride 1, up to three levels per launch
bool launch_conv3x3_c16_tile(const Conv3Args *a, int n, hipStream_t s) {
    static const bool on = !(frt_tuning_env("FRT_DET_C16_TILE") && frt_tuning_env("FRT_DET_C16_TILE")[0] == '0');
    if (!on || n < 1 || n > 3) return false;
    const int cout = a[0].Cout;
    if (cout != 16 && cout != 32) return false;
    Conv3Tile mm;
    int base = 0;
    for (int i = 0; i < 3; ++i) {
        const Conv3Args &q = a[i < n ? i : 0];
        if (q.Cin != 16 || q.Cout != cout || q.stride != 1 || q.H != q.Ho || q.W != q.Wo) return false;
        if (q.out2 && (q.split <= 0 || q.split >= cout)) return false;
        mm.p[i] = q;
        mm.tiles_x[i] = (q.Wo + 7) / 8;
        mm.tiles_y[i] = (q.Ho + 7) / 8;
        mm.base[i] = base;
        if (i < n) base += q.B * mm.tiles_x[i] * mm.tiles_y[i];
    }
    mm.base[3] = base;
    for (int i = n; i < 3; ++i) mm.base[i] = base;  // (unused levels: empty ranges)
    if (cout == 32) hipLaunchKernelGGL((conv3x3_c16_tile_kernel<32>), dim3((unsigned)base), dim3(64), 0, s, mm);
    else hipLaunchKernelGGL((conv3x3_c16_tile_kernel<16>), dim3((unsigned)base), dim3(64), 0, s, mm);
    return true;
}
