// PARKED EXPERIMENT (round 3) - not part of libfrt.so.  Built, parity-green (tests/test_gpu_embedder.py, tests/test_gpu_headline.py with this
// kernel serving the 34 dominant launches), measured slower than conv_patch_kernel: 40.9 -> 43.2 us per launch, pipelined step 3.274 -> 3.348 ms
// (profiles/r03/r03g_patch2_embed_ab.txt, profiles/r03/r03g_patch2_bench_ab.txt).  Kept as source for the record; to rebuild it, add it to
// csrc/Makefile's NAMES, declare conv_patch2_applies / launch_conv_patch2 in frt_kernels.h and call them from launch_conv_mfma's CV_P_255 case.
// A balanced second form (tiles 0-2 / 4-6 x both cout blocks + the middle tile shared, one block per pixel half: 7 MFMA slots per kk step on
// every wave, 4 B-fragment reads instead of 7) measured 40.8 -> 42.2 us per launch, step 3.185 -> 3.241 ms (profiles/r03/r03n_patch2_balanced_ab.txt)
// and was dropped with one parity test still failing: DESIGN 3.15.
//
// ArcFace IR-50: the dominant 3x3 stride-1 convolutions (Cout % 128 == 0, strips of up to 224 pixel slots: 26 x 256 -> 256 at 14x14,
// 6 x 128 -> 128 at 28x28, 128 -> 256 at 28x28, 256 -> 512 at 14x14; model_irse.py:58-66), fp16 NHWC, fp32 accumulation on
// v_mfma_f32_32x32x16_f16.  Round 3 re-tiling of conv_patch_kernel (kernels_arc.hip), same strip / patch / DMA design.
//
// What bounded the round-1/2 kernel: a wave owned ONE 32-cout block x 7 pixel tiles, so every 1 KB B (pixel) fragment it read from LDS
// fed a single MFMA, and the four waves of a workgroup read the SAME fragments (same pixels, different couts): 4 waves x 1 KB per
// 32-clock MFMA = 128 B/clk - exactly the CU's LDS bandwidth.  LDS time equalled MFMA time, the matrix pipe could be busy at most when
// the two overlapped perfectly (measured 34.5 % over the launch, 54 % inside the K loop), and the second recogniser pass's workgroup on
// the same CU competed for the same 128 B/clk.
//
// Here a wave owns TWO cout blocks (64 couts) and HALF of the strip's pixel tiles (waves 0-1: tiles 0-3, waves 2-3: tiles 4-6): a B
// fragment feeds two MFMAs, and only two waves read it: LDS bytes per MFMA halve (14 KB instead of 28 KB per kk step and workgroup).
// Price: the workgroup tile is 8 tile slots for 7 tiles (the waves with 4 tiles set the pace: 8 instead of 7 MFMA slots per kk step on
// the critical SIMDs, the other two idle a quarter of the time), and each 32-cout block's weight fragments are streamed by two waves
// instead of one (twice the L2 -> CU weight bytes).  To stay at 256 registers - two waves per SIMD, so that a workgroup of the other
// recogniser pass still shares the CU - the weight ring is kk-granular: the registers of (step t, kk) are refilled with (step t + 2, kk)
// the moment their MFMAs have issued (64 registers for two cout blocks instead of 96 for a three-step ring).  Nine taps per chunk is
// odd, so the ring parity alternates per chunk: the chunk loop is unrolled by two (Cin is 128 / 256 / 512: always an even chunk count).
#include <type_traits>

#include "frt_kernels.h"

namespace {

constexpr int PROW = 144;  // bytes per patch pixel row (128 data + 16 pad: conflict-free ds_read_b128)
constexpr int EROW = 36;   // floats per pixel row of the epilogue tile

template <int PPS>  // patch DMA pieces (4 KB per workgroup each) per patch buffer
__global__ __launch_bounds__(256, 2) void conv_patch2_kernel(ConvMfmaArgs p, int R) {
    constexpr int PATCH_B = PPS * 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int ch = wave & 1, ph = wave >> 1;  // cout half (64 couts), pixel half (tiles 4 ph ...)
    const int H = p.H, W = p.W, Wp = W + 2;
    const int NP = (R + 2) * Wp;
    const int strips_per_img = H / R;

    const int n_co_tiles = p.Cout >> 7;
    const int nblk = gridDim.x, bq = nblk >> 3, brem = nblk & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
    const int co_tile = lid % n_co_tiles, strip = lid / n_co_tiles;
    const int co_base = co_tile * 128, cow = ch * 64;
    const int img0 = strip / strips_per_img;
    const int row0 = (strip % strips_per_img) * R;
    const int n_chunks = p.Cin >> 6;
    char *patch = smem;

    // ---- patch DMA: piece q of this lane covers 16-byte chunk g = (q*4 + wave)*64 + lane of the patch image (pixel row g / 9, part
    //      g % 9; part 8 is the pad).  The source offsets are recomputed per burst (about 15 VALU operations per piece, once per chunk,
    //      in the shadow of the MFMAs) instead of living in ten registers: the kernel runs at the 256-register limit.
    const int g0 = wave * 64 + lane;
    const float inv_wp = 1.0f / (float)Wp;
    auto issue_patch = [&](int c) {
        const bool real = c < n_chunks;
        char *pl = patch + (c & 1) * PATCH_B + wave * 1024;
#pragma unroll
        for (int q = 0; q < PPS; ++q) {
            int g = g0 + q * 256;
            asm volatile("" : "+v"(g));          // keep the recomputation HERE: hoisted out of the chunk loop it would cost the ten registers again
            const int prow = (g * 7282) >> 16;  // g / 9 for g < 2816
            const int pos = g - prow * 9;
            const int pr = (int)(((float)prow + 0.5f) * inv_wp), pc = prow - pr * Wp;
            const int iy = row0 + pr - 1, ix = pc - 1;
            const bool ok = real && pos < 8 && prow < NP && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const half_t *src = ok ? p.x + (unsigned)(((img0 * H + iy) * W + ix) * p.Cin + pos * 8 + (c << 6)) : p.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(pl + q * 4096), 16, 0, 0);
        }
    };
    // fragment-ordered weights: [32-cout block][chunk][tap][kk][lane][8 halfs]; this wave's two blocks are `blk_stride` apart
    const long blk_stride = (long)n_chunks * (9 * 4 * 512);
    const half_t *wfrag = p.wf + (long)((co_base + cow) >> 5) * blk_stride + lane * 8;
    const half_t *wfrag1 = wfrag + blk_stride;

    // pixel slot -> flattened output pixel (linear enumeration over the padded row width, as in conv_patch_kernel)
    const long m0 = ((long)img0 * H + row0) * W;
    const long Mtot = (long)p.B * H * W;
    auto slot_pixel = [&](int sl, long &m) -> bool {
        const int rr = (int)(((float)sl + 0.5f) * inv_wp);
        const int cc = sl - rr * Wp;
        m = m0 + rr * W + cc;
        return rr < R && cc < W && m < Mtot;
    };

    auto body = [&](auto ntw_c) {
        constexpr int NTW = decltype(ntw_c)::value;  // pixel tiles of this wave: 4 (ph == 0) or 3 (ph == 1)
        const int tile0 = ph * 4;
        // B-fragment addresses: slot (tile0 + j) * 32 + r -> patch row of tap (0, 0); consecutive tiles are 32 patch rows = 4608 bytes
        // apart, so ONE base register serves all tiles and the tile / kk offsets are instruction immediates.  (Dead slots behind the
        // strip's last pixel read whatever lies there - at most row 255 of a 284-row buffer - and are dropped in the epilogue.)
        const int pbase0 = (tile0 * 32 + r) * PROW + hi * 16;
        constexpr int TSTEP = 32 * PROW;
        floatx16 acc[NTW][2];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][b][e] = 0.f;

        half8 areg[2][2][4];  // [ring parity][cout block][kk]
        auto load_w = [&](int c, int tap, int kk, auto par_c) {  // wave-uniform c, tap, kk; clamped at the tail (values unused there)
            constexpr int PAR = decltype(par_c)::value;
            const unsigned woff = c < n_chunks ? (unsigned)(c * 9 + tap) * 2048u : 0u;  // halfs; kk * 512 is an instruction immediate
            areg[PAR][0][kk] = *reinterpret_cast<const half8 *>(wfrag + woff + kk * 512);
            areg[PAR][1][kk] = *reinterpret_cast<const half8 *>(wfrag1 + woff + kk * 512);
        };
        // prologue: patch(0) completely, then the weight fragments of steps 0 and 1 (16 loads)
        issue_patch(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) load_w(0, 0, kk, std::integral_constant<int, 0>{});
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) load_w(0, 1, kk, std::integral_constant<int, 1>{});
        half8 bf[NTW];
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // this wave's patch(0) pieces have landed (younger: the 16 fragment loads)
        __builtin_amdgcn_s_barrier();                       // ... and everybody else's
#pragma unroll
        for (int j = 0; j < NTW; ++j) bf[j] = *reinterpret_cast<const half8 *>(patch + pbase0 + j * TSTEP);

        // one (chunk, tap) step; CPAR = parity of the chunk index (compile time: the ring slot of a step is (9 c + tap) & 1)
        auto step = [&](int c, auto tap_c, auto cpar_c) {
            constexpr int TAP = decltype(tap_c)::value, CPAR = decltype(cpar_c)::value;
            constexpr int NTAP = (TAP + 1) % 9;
            constexpr int PAR = (TAP + CPAR) & 1;
            const int pbuf = CPAR * PATCH_B;
            const char *bp = patch + pbase0 + ((TAP / 3) * Wp + (TAP % 3)) * PROW + pbuf;                                              // this tap
            const char *bpn = patch + pbase0 + ((NTAP / 3) * Wp + (NTAP % 3)) * PROW + (TAP == 8 ? (CPAR ^ 1) * PATCH_B : pbuf);      // the next one
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk == 3 && TAP == 8) {
                    // chunk boundary: from here on the refills read the NEXT chunk's patch buffer.  This wave's pieces of it were issued
                    // at tap 1 behind that step's kk = 0 fragment loads; younger since then: 6 (tap 1) + 6 x 8 (taps 2 - 7) + 6 (tap 8,
                    // kk 0 - 2) = 60 fragment loads.  Every wave's reads of the buffer about to be recycled have returned; then all meet.
                    asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[PAR][0][kk], bf[j], acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[PAR][1][kk], bf[j], acc[j][1], 0, 0, 0);
                    // refill the fragment just consumed with the one a kk-slot ahead (the next tap's first one behind kk = 3)
                    if (kk < 3) bf[j] = *reinterpret_cast<const half8 *>(bp + j * TSTEP + (kk + 1) * 32);
                    else bf[j] = *reinterpret_cast<const half8 *>(bpn + j * TSTEP);
                    if (j == NTW - 1) {
                        // the weight registers of (this step, kk) are free: fetch (step + 2, kk) into them
                        constexpr int T2 = TAP + 2;
                        load_w(T2 < 9 ? c : c + 1, T2 % 9, kk, std::integral_constant<int, PAR>{});
                        if (kk == 0 && TAP == 1) issue_patch(c + 1);  // the next chunk's patch, one burst (it has 7.75 steps to land)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        for (int c = 0; c < n_chunks; c += 2) {
            step(c, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            step(c, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            step(c, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
            step(c, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
            step(c, std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{});
            step(c, std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{});
            step(c, std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{});
            step(c, std::integral_constant<int, 7>{}, std::integral_constant<int, 0>{});
            step(c, std::integral_constant<int, 8>{}, std::integral_constant<int, 0>{});
            step(c + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            step(c + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            step(c + 1, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
            step(c + 1, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
            step(c + 1, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
            step(c + 1, std::integral_constant<int, 5>{}, std::integral_constant<int, 1>{});
            step(c + 1, std::integral_constant<int, 6>{}, std::integral_constant<int, 1>{});
            step(c + 1, std::integral_constant<int, 7>{}, std::integral_constant<int, 1>{});
            step(c + 1, std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's dummy DMAs still target LDS
        __syncthreads();

        // ------------------------------------------------------------------ epilogue: per wave 2 cout blocks x NTW pixel tiles through LDS
        float *ep = reinterpret_cast<float *>(smem) + wave * (32 * EROW);
        const int chunk = lane & 3;
        const bool two = p.mode == EPI_BN_ADD_BN && p.out1;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int cch = co_base + cow + b * 32 + chunk * 8;
            floatx4 q0[2], q1[2], q2[2], q3[2];
            q0[0] = *reinterpret_cast<const floatx4 *>(p.p0 + cch);
            q0[1] = *reinterpret_cast<const floatx4 *>(p.p0 + cch + 4);
            if (p.mode != EPI_PRELU) {
                q1[0] = *reinterpret_cast<const floatx4 *>(p.p1 + cch);
                q1[1] = *reinterpret_cast<const floatx4 *>(p.p1 + cch + 4);
            }
            if (two) {
                q2[0] = *reinterpret_cast<const floatx4 *>(p.p2 + cch);
                q2[1] = *reinterpret_cast<const floatx4 *>(p.p2 + cch + 4);
                q3[0] = *reinterpret_cast<const floatx4 *>(p.p3 + cch);
                q3[1] = *reinterpret_cast<const floatx4 *>(p.p3 + cch + 4);
            }
            // the shortcut has the output's geometry; its values are requested one pixel tile ahead (all tiles of both blocks at once do
            // not fit the 256-register budget next to the 128 accumulator registers)
            half8 sc8[2][2];
            auto load_sc = [&](int j, half8 (&dst)[2]) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    long m;
                    const bool ok = slot_pixel((tile0 + j) * 32 + (lane >> 2) + 16 * it, m);
                    dst[it] = *reinterpret_cast<const half8 *>(p.sc + (ok ? m : 0) * p.Cout + cch);
                }
            };
            if (p.mode == EPI_BN_ADD_BN) load_sc(0, sc8[0]);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (p.mode == EPI_BN_ADD_BN && j + 1 < NTW) load_sc(j + 1, sc8[(j + 1) & 1]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const floatx4 v = {acc[j][b][4 * g], acc[j][b][4 * g + 1], acc[j][b][4 * g + 2], acc[j][b][4 * g + 3]};
                    *reinterpret_cast<floatx4 *>(ep + r * EROW + 8 * g + 4 * hi) = v;
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int px = (lane >> 2) + 16 * it;
                    const floatx4 v0 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8);
                    const floatx4 v1 = *reinterpret_cast<const floatx4 *>(ep + px * EROW + chunk * 8 + 4);
                    long m;
                    if (!slot_pixel((tile0 + j) * 32 + px, m)) continue;
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
                        for (int e = 0; e < 8; ++e) v[e] += (float)sc8[j & 1][it][e];
                    }
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                    *reinterpret_cast<half8 *>(p.out0 + m * p.Cout + cch) = o;
                    if (two) {
                        half8 z;
#pragma unroll
                        for (int e = 0; e < 8; ++e) z[e] = (half_t)(v[e] * q2[e >> 2][e & 3] + q3[e >> 2][e & 3]);
                        *reinterpret_cast<half8 *>(p.out1 + m * p.Cout + cch) = z;
                    }
                }
            }
        }
    };
    if (ph == 0) body(std::integral_constant<int, 4>{});
    else body(std::integral_constant<int, 3>{});
}

}  // namespace

// Covers what conv_patch_kernel<10, 1, 5, false, 0, false, 7, 1, 3, false> covers: 3x3 / stride 1 / pad 1, Cout % 128 == 0, an even number
// of 64-channel chunks, one image per strip with R x (W + 2) <= 224 pixel slots and a patch of <= 10 DMA pieces; not the IR-SE epilogue.
bool conv_patch2_applies(const ConvMfmaArgs &a, int R, int n_img, int nt, int slots) {
    if (!a.wf || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.Cout % 128 || a.Cin % 128 || a.splits != 1 || a.H != a.W) return false;
    if (a.mode != EPI_PRELU && a.mode != EPI_BN && a.mode != EPI_BN_ADD_BN) return false;
    if (a.mode == EPI_BN_ADD_BN && !(a.sc && a.sc_stride == 1 && a.sc_h == a.Ho && a.sc_w == a.Wo)) return false;
    if (n_img != 1 || nt != 7 || slots > 10 || R * (a.W + 2) > 224 || R * (a.W + 2) <= 192 || a.H % R) return false;  // (7 tiles: 4 + 3 per pixel half)
    static const bool off = frt_tuning_env("FRT_CONV_PATCH2") && frt_tuning_env("FRT_CONV_PATCH2")[0] == '0';
    return !off;
}

void launch_conv_patch2(const ConvMfmaArgs &a, int R, hipStream_t s) {
    constexpr int PPS = 10;
    const size_t lds = (size_t)2 * PPS * 4096;
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_patch2_kernel<PPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int strips = a.B * (a.H / R);
    hipLaunchKernelGGL((conv_patch2_kernel<PPS>), dim3(strips * (a.Cout / 128)), dim3(256), lds, s, a, R);
}
