// ArcFace IR-50 / IR-SE-50 at MEDIUM batches (about 10 - 40 faces per pass): the 3x3 stride-1 convolutions with 256 / 512 input channels
// (14x14 and 7x7 maps: 31 of the 48 convs of a pass, model_irse.py:48-66), fp16 NHWC, fp32 accumulation on v_mfma_f32_32x32x16_f16.
//
// This is the shape of BASELINE configs[3] as written (one 32-frame batch over 8 GPUs = 4 frames = 16 faces per rank), of K = 1 (32 faces
// per step) and of a coalesced multi-threaded server (frt_coalescer_*).  Neither older kernel family is at home there:
//   * the strip kernels (kernels_arc.hip) give a workgroup 128 output channels of a strip, i.e. 590 KB of weights (256 -> 256) or 1.18 MB
//     (512 -> 512) through ONE CU's vector memory path (~ 80 - 130 GB/s) for 2 us of MFMAs: 13 us per 14x14 layer and 21 us per 7x7 layer
//     at 16 faces (profiles/r04/r04c_layers_16.txt), 26 + 4 such launches per pass;
//   * conv_small_kernel (kernels_arc_small.hip) makes the unit 32 couts x 32 pixels but gathers every pixel 9 times from L2: 294 KB of
//     operands per unit, fine for 25 units per block and too much for 100.
// What bounds a launch here is the operand bytes a CU has to pull in, so the tile is chosen to minimise THEM at ~ one workgroup per CU:
// with Mc couts x Np pixels per workgroup and Mc * Np fixed by the chip (layer / 256 CUs), bytes = Mc * 9 * Cin * 2 + Np * Cin * 2 * halo is
// smallest at Mc = 32: ONE 32-cout block x a strip of 98 - 196 pixels.  The four waves then cannot split couts (there is one block); they
// split K: wave w owns the 64-channel chunks w, w + 4 (Cin = 256: one chunk each; 512: two) -
//   * its chunk of the strip's halo'd patch sits in a wave-private LDS region (LDS-DMA, global_load_lds_dwordx4; 144-byte pixel rows as in
//     conv_patch_kernel: conflict-free ds_read_b128 without a swizzle), no workgroup barrier in the K loop;
//   * ALL its weight fragments (36 KB per chunk, the strip kernels' fragment order: one contiguous KB per MFMA) are requested up front into
//     registers - one wave per SIMD owns 512 registers, 144 per chunk hold the weights - so the whole operand stream of a workgroup
//     (147 + 74 KB for 14x14x256, 295 + 94 KB for 7x7x512) is in flight at once instead of behind a 2-deep ring;
//   * the four partial accumulators meet in LDS and are added in wave order (deterministic, independent of the position in the batch),
//     pixel tile j by wave j % 4, which also runs that tile's epilogue (the strip kernel's: transpose through LDS, 16-byte stores).
// 14x14x256 at 16 faces: 256 workgroups x 221 KB.  Epilogues: PReLU; BN; BN + shortcut tensor + next BN (no SE tail: IR-SE takes the
// stand-alone tail behind this kernel).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "frt_kernels.h"

namespace {

constexpr int PROW = 144;  // bytes per patch pixel row (64 channels + one 16-byte pad slot)
constexpr int EROW = 36;   // floats per pixel row of the epilogue transpose tile

// NT pixel tiles per strip, CPW 64-channel chunks per wave (Cin = 256 * CPW), HW = map size (14 or 7), R = rows per strip.
// PIECES = DMA pieces per patch chunk: one piece = 7 patch rows = 63 sixteen-byte units (lane 63 sits out), so a lane's (row within the
// piece, 16-byte slot) never changes and a piece's source address is a dozen instructions (the first version divided the unit index out
// per piece: 2.3 us of address arithmetic in front of a 2 us K loop, profiles/r04/r04e_ks_stamps.txt).
template <int NT, int CPW, int HW, int R>
__global__ __launch_bounds__(256, 1) void conv_ks_kernel(ConvMfmaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = HW, W = HW, Wp = W + 2, Cin = 256 * CPW, nch = 4 * CPW;
    constexpr int NP = (R + 2) * Wp, PIECES = (NP + 6) / 7, PATCH_B = PIECES * 1008;
    constexpr int strips_per_img = H / R;
    constexpr int TPW = (NT + 3) / 4;  // pixel tiles a wave finishes
    static_assert(Wp == 16 || Wp == 9, "row decomposition below");
    static_assert((NP + 2) * PROW <= PATCH_B, "taps of the last slots stay inside the patch");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int strip = blockIdx.x, blk = blockIdx.y;
    const int img = strip / strips_per_img, row0 = (strip - img * strips_per_img) * R;
    char *patch = smem + wave * (CPW * PATCH_B);
#ifdef FRT_ABLATE
    // timing build: phase stamps (100 MHz constant clock) of wave 0 of the first and the last workgroup, 8 per workgroup, into p.outf
    unsigned long long *stamps = (p.outf && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && blockIdx.y == 0)
                                     ? reinterpret_cast<unsigned long long *>(p.outf) + (blockIdx.x ? 8 : 0) : nullptr;
#define KS_STAMP(i) do { if (stamps && lane == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KS_STAMP(i) do { } while (0)
#endif
    KS_STAMP(0);
    // ---- weights: every fragment of this wave's chunks, straight into registers.  The first three taps go out before the patch pieces,
    //      the rest behind them: "at most that many loads outstanding" then means the patch and the first taps' weights have landed (VMEM
    //      retires in order) and the K loop starts under the rest of the weight stream
    half8 areg[CPW][9][4];
    const half_t *wfrag = p.wf + (long)blk * nch * (9 * 4 * 512) + lane * 8;
    auto load_w = [&](auto kc, auto tc) {
        constexpr int k = decltype(kc)::value, tap = decltype(tc)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) areg[k][tap][kk] = *reinterpret_cast<const half8 *>(wfrag + ((long)((wave + 4 * k) * 9 + tap) * 4 + kk) * 512);
    };
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    KS_STAMP(1);
    // ---- patch DMA: piece q = patch rows 7q .. 7q + 6; lane = (row within the piece) * 9 + 16-byte slot (slot 8 = the row's pad)
    if (lane < 63) {
        const int lr = lane / 9, pos = lane - lr * 9;
        const half_t *xb = p.x + ((long)(img * H + row0 - 1) * W - 1) * Cin + wave * 64 + pos * 8;  // pixel (row0 - 1, -1), this wave's first chunk
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int prow = 7 * q + lr;
            const int pr = Wp == 16 ? prow >> 4 : (prow * 57) >> 9, pc = prow - pr * Wp;  // (prow / 9 exactly for prow < 500)
            const bool in = (pos < 8) & (prow < NP) & ((unsigned)(row0 + pr - 1) < (unsigned)H) & ((unsigned)(pc - 1) < (unsigned)W);
            const int off = (pr * W + pc) * Cin;
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                const half_t *src = in ? xb + off + k * 256 : p.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(patch + k * PATCH_B + q * 1008), 16, 0, 0);
            }
        }
    }
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 7>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
    if constexpr (CPW == 2) {
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 5>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 6>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 7>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{});
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW == 2 ? 60 : 24) : "memory");
    // The patch is wave-private: no barrier follows.  Round 4 suspected (kernels_det_wave.hip) that a ds_read issued right behind the vmcnt wait
    // of an LDS-DMA could still see the old LDS contents; round 5's probe (tools/ubench/lds_dma_raw.hip: 0 stale words in 4.2e10 with nothing in
    // between) says the issuing wave's covering vmcnt does order its own reads, and the round-4 symptom was a miscounted wait.  The sleep is
    // kept as margin (64 cycles once per workgroup), not as the mechanism.
    asm volatile("s_sleep 2" ::: "memory");
    KS_STAMP(2);

    // ---- epilogue operands of the tiles this wave finishes (they land under the K loop)
    const int chunk8 = lane & 3;
    const int cch = blk * 32 + chunk8 * 8;
    const long m0 = ((long)img * H + row0) * W;
    auto slot_pixel = [&](int sl, long &m) -> bool {  // pixel slot (enumerated over the padded row width) -> flattened output pixel
        const int rr = Wp == 16 ? sl >> 4 : (sl * 57) >> 9;
        const int cc = sl - rr * Wp;
        m = m0 + rr * W + cc;
        return rr < R && cc < W;
    };
    floatx4 q0[2], q1[2], q2[2], q3[2];
    q0[0] = *reinterpret_cast<const floatx4 *>(p.p0 + cch);
    q0[1] = *reinterpret_cast<const floatx4 *>(p.p0 + cch + 4);
    if (p.mode != EPI_PRELU) {
        q1[0] = *reinterpret_cast<const floatx4 *>(p.p1 + cch);
        q1[1] = *reinterpret_cast<const floatx4 *>(p.p1 + cch + 4);
    }
    const bool two_out = p.mode == EPI_BN_ADD_BN && p.out1;
    if (two_out) {
        q2[0] = *reinterpret_cast<const floatx4 *>(p.p2 + cch);
        q2[1] = *reinterpret_cast<const floatx4 *>(p.p2 + cch + 4);
        q3[0] = *reinterpret_cast<const floatx4 *>(p.p3 + cch);
        q3[1] = *reinterpret_cast<const floatx4 *>(p.p3 + cch + 4);
    }
    half8 sc8[TPW][2];
    if (p.mode == EPI_BN_ADD_BN) {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int j = wave + 4 * t;
                long m = 0;
                const bool ok = j < NT && slot_pixel(j * 32 + (lane >> 2) + 16 * it, m);
                sc8[t][it] = *reinterpret_cast<const half8 *>(p.sc + (ok ? m : m0) * p.Cout + cch);
            }
    }

    // ---- K loop: this wave's chunk(s) x 9 taps x 4 k-steps, every pixel tile of the strip
    int pbase[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int sl = j * 32 + r;
        pbase[j] = (sl < R * Wp ? sl : 0) * PROW + hi * 16;
    }
    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    // B fragments one (tap, k-step) ahead in a two-deep register ring; sched_barrier pins the steps (left alone the compiler hoists dozens
    // of fragment reads and spills at 7 tiles)
    half8 bf[2][NT];
    constexpr int STEPS = CPW * 36;
    auto frag_off = [&](int st) {  // LDS byte offset of step st = ((chunk k, tap), kk) relative to a slot's tap-(0,0) row
        const int k = st / 36, tap = (st / 4) % 9, kk = st & 3;
        return ((tap / 3) * Wp + (tap % 3)) * PROW + k * PATCH_B + kk * 32;
    };
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[0][j] = *reinterpret_cast<const half8 *>(patch + pbase[j] + frag_off(0));
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        if (st + 1 < STEPS) {
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[(st + 1) & 1][j] = *reinterpret_cast<const half8 *>(patch + pbase[j] + frag_off(st + 1));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[st / 36][(st / 4) % 9][st & 3], bf[st & 1][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- the four partial sums meet in LDS (over the patches, which nobody reads any more), added in wave order
    asm volatile("" ::"v"(acc[0][0]));
    KS_STAMP(3);
    __syncthreads();
    KS_STAMP(4);
    float *red = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<floatx4 *>(red + ((((wave * NT + j) * 4 + g) * 64) + lane) * 4) = floatx4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
    __syncthreads();
    KS_STAMP(5);
    constexpr int RED_B = 4 * NT * 4096, EP_B = 4 * 32 * EROW * 4;
    constexpr int MAIN_B = 4 * CPW * PATCH_B > RED_B ? 4 * CPW * PATCH_B : RED_B;
    constexpr int EP_OFF = RED_B + EP_B <= MAIN_B ? RED_B : MAIN_B;  // the transpose tiles sit behind the partial sums, over dead patches where they fit
    float *ep = reinterpret_cast<float *>(smem + EP_OFF) + wave * (32 * EROW);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int j = wave + 4 * t;
        if (j >= NT) break;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            floatx4 v = *reinterpret_cast<const floatx4 *>(red + ((((0 * NT + j) * 4 + g) * 64) + lane) * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const floatx4 *>(red + ((((w * NT + j) * 4 + g) * 64) + lane) * 4);
            *reinterpret_cast<floatx4 *>(ep + r * EROW + 8 * g + 4 * hi) = v;  // D[cout][pixel] -> pixel rows
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int px = (lane >> 2) + 16 * it;
            const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk8 * 8);
            const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk8 * 8 + 4);
            long m;
            if (!slot_pixel(j * 32 + px, m)) continue;
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.mode == EPI_PRELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * q0[e >> 2][e & 3];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * q0[e >> 2][e & 3] + q1[e >> 2][e & 3];
            }
            if (p.mode == EPI_BN_ADD_BN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)sc8[t][it][e];
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
            *reinterpret_cast<half8 *>(p.out0 + m * p.Cout + cch) = o;
            if (two_out) {
                half8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[e] * q2[e >> 2][e & 3] + q3[e >> 2][e & 3]);
                *reinterpret_cast<half8 *>(p.out1 + m * p.Cout + cch) = z;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KS_STAMP(6);
}

#ifdef FRT_ABLATE
// timing build, FRT_KS_STAMPS=1: every launch leaves its 16 stamps in the next slot of a device ring; printed at exit
struct KsStamps {
    unsigned long long *dev = nullptr;
    int n = 0;
    static constexpr int CAP = 4096;
    int kind[CAP];
    ~KsStamps() {
        if (!dev || !n) return;
        std::vector<unsigned long long> h((size_t)std::min(n, CAP) * 16);
        if (hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
        for (int k : {2, 4, 7}) {
            double d[2][7] = {};
            int cnt = 0;
            for (int i = std::min(n, CAP) / 2; i < std::min(n, CAP); ++i) {  // second half of the run (warm)
                if (kind[i] != k) continue;
                ++cnt;
                for (int b = 0; b < 2; ++b)
                    for (int j = 1; j < 7; ++j) d[b][j] += (double)(h[(size_t)i * 16 + b * 8 + j] - h[(size_t)i * 16 + b * 8 + j - 1]) * 0.01;
            }
            if (!cnt) continue;
            for (int b = 0; b < 2; ++b)
                fprintf(stderr, "[ks stamps] NT %d %s workgroup, %d launches: dma issue %.2f | weights issued + patch landed %.2f | K loop %.2f | barrier %.2f | partials written %.2f | sum + epilogue %.2f us\n",
                        k, b ? "last" : "first", cnt, d[b][1] / cnt, d[b][2] / cnt, d[b][3] / cnt, d[b][4] / cnt, d[b][5] / cnt, d[b][6] / cnt);
        }
    }
};
static KsStamps g_stamps;
#endif

template <int NT, int CPW, int HW, int R>
void launch_ks_t(const ConvMfmaArgs &a, hipStream_t s) {
    constexpr int PIECES = ((R + 2) * (HW + 2) + 6) / 7;
    constexpr int PATCH_B = PIECES * 1008, RED_B = 4 * NT * 4096, EP_B = 4 * 32 * EROW * 4;
    constexpr int MAIN_B = 4 * CPW * PATCH_B > RED_B ? 4 * CPW * PATCH_B : RED_B;
    constexpr int EP_OFF = RED_B + EP_B <= MAIN_B ? RED_B : MAIN_B;
    constexpr size_t lds = (size_t)(MAIN_B > EP_OFF + EP_B ? MAIN_B : EP_OFF + EP_B);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ks_kernel<NT, CPW, HW, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    constexpr int spi = HW / R;
#ifdef FRT_ABLATE
    static const bool want_stamps = frt_tuning_env("FRT_KS_STAMPS") != nullptr;
    if (want_stamps) {
        if (!g_stamps.dev && hipMalloc(reinterpret_cast<void **>(&g_stamps.dev), KsStamps::CAP * 16 * 8) != hipSuccess) g_stamps.dev = nullptr;
        if (g_stamps.dev && g_stamps.n < KsStamps::CAP) {
            ConvMfmaArgs b = a;
            b.outf = reinterpret_cast<float *>(g_stamps.dev + (size_t)g_stamps.n * 16);
            g_stamps.kind[g_stamps.n++] = NT;
            hipLaunchKernelGGL((conv_ks_kernel<NT, CPW, HW, R>), dim3(a.B * spi, a.Cout / 32), dim3(256), lds, s, b);
            return;
        }
    }
#endif
    hipLaunchKernelGGL((conv_ks_kernel<NT, CPW, HW, R>), dim3(a.B * spi, a.Cout / 32), dim3(256), lds, s, a);
}

// faces per pass this kernel takes (above: the strip kernels; conv_small_kernel is asked first and keeps the small batches)
int ks_max_faces(int H) {
    static const int e14 = frt_tuning_env("FRT_CONV_KS_MAX14") ? atoi(frt_tuning_env("FRT_CONV_KS_MAX14")) : 40;
    static const int e7 = frt_tuning_env("FRT_CONV_KS_MAX7") ? atoi(frt_tuning_env("FRT_CONV_KS_MAX7")) : 40;
    return H == 14 ? e14 : e7;
}

}  // namespace

bool conv_ks_applies(const ConvMfmaArgs &a) {
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || !a.wf || a.splits != 1 || a.Cout % 32 || a.H != a.W || a.Ho != a.H || a.Wo != a.W) return false;
    if (!((a.H == 14 && a.Cin == 256) || (a.H == 7 && a.Cin == 512))) return false;
    if (a.mode != EPI_PRELU && a.mode != EPI_BN && a.mode != EPI_BN_ADD_BN) return false;
    if (a.mode == EPI_BN_ADD_BN && (a.scx || !a.sc || a.sc_stride != 1 || a.sc_h != a.Ho || a.sc_w != a.Wo)) return false;
    return a.B >= 1 && a.B <= ks_max_faces(a.H);
}

bool launch_conv_ks(const ConvMfmaArgs &a, hipStream_t s) {
    if (!conv_ks_applies(a)) return false;
    if (a.H == 7) {
        launch_ks_t<2, 2, 7, 7>(a, s);   // whole 7x7 images: 9 x 9 patch rows -> 12 pieces per chunk, 63 slots in 2 tiles
    } else if (a.B * 2 * (a.Cout / 32) <= 320) {
        launch_ks_t<4, 1, 14, 7>(a, s);  // half images (7 rows): 9 x 16 patch rows -> 21 pieces, 112 slots in 4 tiles
    } else {
        launch_ks_t<7, 1, 14, 14>(a, s); // whole images: 16 x 16 patch rows -> 37 pieces, 224 slots in 7 tiles
    }
    return true;
}
