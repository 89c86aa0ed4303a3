// Detector conv_dw blocks (depthwise 3x3 + BN + ReLU -> pointwise 1x1 + BN + ReLU; net.py:14-24 of the reference's RetinaFace), round 4:
// one WAVE owns 64 consecutive output pixels of one image and ALL input channels.
//
// Why a second formulation next to dwpw_mfma_kernel (kernels_det_mfma.hip): that kernel maps a thread to (channel, 4 pixels), pays ~50
// VALU instructions per output value on addresses, border selects, the hi/lo split and 2-byte LDS stores, and goes through four
// barrier-separated phases per workgroup; profiles/r04/r04o_det_ablations.txt shows the 40x40 blocks issue-/latency-bound at 0.16 of HBM with
// 85 % of the launch left when all memory traffic is removed.  Here
//   * lane = pixel, the loop runs over channels: the depthwise weights of a channel are wave-uniform, i.e. SGPR operands of
//     v_pk_fma_f32 (two channels per instruction), and there is no per-element address arithmetic at all;
//   * the input rows a wave needs (the rows its 64 pixels touch + one above and below) come in by LDS-DMA (global_load_lds_dwordx4) in chunks
//     of 8 channels: LDS image [row][channel][16-byte slot] with one zero slot in front of every channel row, one DMA instruction per 64
//     (channel, slot) positions of a row.  The zero slots are never written (exec-masked lanes; the region is zeroed once) and rows outside
//     the image are fetched from a buffer of zeros, so every border case is a plain ds_read2_b32 with immediate offsets - no selects;
//   * the MFMA B operand never goes through LDS: after 16 channels a lane holds their 16 depthwise outputs for ITS pixel, as fp16 hi / lo
//     pairs; lanes 0-31 are pixel tile A, lanes 32-63 tile B, and v_permlane32_swap_b32 (gfx950) turns "channels 0-7 | channels 8-15 of my
//     pixel" into the two tiles' (pixel, k = 8*hi + j) fragments: 8 swaps per 16 channels;
//   * no barrier anywhere: the LDS region is wave-private and ordered by s_waitcnt vmcnt (+ a step of distance: see WaveGeom::younger);
//   * depthwise weights and tap registers live in inline asm with FIXED scalar registers: what the compiler does to in-flight asm results and to
//     lgkmcnt is told where it happened, below.
// Arithmetic is the old kernel's operation for operation (same fma chain per pixel, same RNE split, same MFMA order), so the output is
// bit-identical to dwpw_mfma_kernel<..., SPLIT = true> - checked by tools/dwpw_wave_check.py and by the detector tests, which run unchanged.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "frt_kernels.h"

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int xcd_tile(int nblocks) {  // block b runs on XCD b % 8: every XCD's L2 sees a contiguous range of tiles
    const int bq = nblocks >> 3, brem = nblocks & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    return (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
}

template <int CIN, int NCB, int H, int W, int STRIDE, int NBUF_>
struct WaveGeom {
    static constexpr int WO = W / STRIDE, HO = H / STRIDE;   // output map
    static constexpr int HW = H * W, HOWO = HO * WO;
    static constexpr int SPAN = (WO + 62) / WO + 1;          // output rows 64 consecutive pixels can touch
    static constexpr int ROWS = STRIDE * (SPAN - 1) + 3;     // input rows behind them
    static constexpr int SL = W / 4 + 1;                     // 16-byte slots per LDS row of one channel: one zero slot + the row
    static constexpr int CPC = 8;                            // channels per DMA chunk
    static constexpr int SPC = CPC / 4;                      // steps (of 4 channels) per chunk
    // LDS layout of a chunk: [input row][channel of the chunk][slot] - the two channels of a pair sit CHS bytes apart (the second offset of
    // ds_read2_b32 reaches 255 dwords), and one DMA instruction covers 64 consecutive (channel, slot) positions of ONE input row
    static constexpr int CHS = SL * 16;
    // (+ W % 32 dwords: CPC * SL * 4 dwords is a multiple of the 32 banks; with the pad a lane's bank is its pixel index + const, rows included)
    static constexpr int ROWB = CPC * CHS + (W % 32) * 4;
    static constexpr int CHUNKB = ROWS * ROWB;
    static constexpr int NDR = (CPC * SL + 63) / 64;         // DMA instructions per input row of a chunk
    static constexpr int DMAOPS = ROWS * NDR;
    static constexpr int NBUF = NBUF_, LA = NBUF_ - 1;       // chunks in flight ahead of the one being consumed
    static constexpr int NCHUNK = CIN / CPC, NG = CIN / 16, NSTEP = CIN / 4;
    static constexpr int NA = 2 * NCB;                       // weight-fragment loads per 16-channel group
    static constexpr int LDS_BYTES = NBUF * CHUNKB + 16;
    static_assert(SPC == 2 && CIN % 16 == 0 && CHS / 4 + 2 < 256 && LA >= 2, "shape not covered");
    // VMEM retires in order: "at most n operations outstanding" with n = the operations issued after the last DMA of chunk c means chunk c
    // has been delivered.  Issue order (a step = two channel pairs = 4 channels, two steps per chunk, four per 16-channel group):
    //   prologue: DMA(0 .. LA-1), A(0);   step s: [s even: DMA(s/2 + LA) at the top] ... [s % 4 == 3: A(s/4 + 1) at the end]
    // The wait for chunk c sits at the top of step 2c - 2, right behind that step's DMA issue - a whole step before the first read of the
    // chunk (issued in step 2c - 1).  Round 4 recorded wrong sums with the read right behind the wait and read that as "the counter drops before
    // the data is readable".  Round 5 re-examined it (advisor finding): tools/ubench/lds_dma_raw.hip hammers exactly the pattern - LDS-DMA into
    // the same bytes, s_waitcnt vmcnt(0) or a counted vmcnt(8) with younger DMAs in flight, ds_read2_b32 of other lanes' slots in the next
    // instruction, 4 096 one-wave workgroups x 2 000 iterations - and finds 0 stale words in 4.2e10 with nothing, s_nop, s_barrier or s_sleep in
    // between (profiles/r05a_lds_dma_raw.txt): the issuing wave's covering vmcnt DOES order its own ds_read behind the DMA, which is also what
    // /opt/skills/guides/MI355X_MICROARCH.md states (two-waves-per-SIMD item 7).  The round-4 failures are explained by the other defect found
    // at the same time and fixed with the compiler barriers in the prologue below: a plain fragment load hoisted above a DMA it was counted
    // behind made the COUNT one operation too generous.  The ordering this kernel relies on is therefore the architectural one (counted
    // vmcnt, in-order VMEM return); the step of distance and chunk 0's s_sleep stay as margin, not as the mechanism.
    static constexpr int younger(int cw) {
        const int at_step = cw == 0 ? -1 : 2 * cw - 2;  // the wait follows the DMA issue of this step (-1: the prologue)
        int n = 0;
        bool seen = false;
        for (int k = 0; k < LA; ++k) {
            if (seen) n += DMAOPS;
            if (k == cw) seen = true;
        }
        if (seen) n += NA;
        for (int st = 0; st <= at_step; ++st) {
            if ((st & 1) == 0 && st / 2 + LA < NCHUNK) {
                if (seen) n += DMAOPS;
                if (st / 2 + LA == cw) seen = true;
            }
            if (st < at_step && (st & 3) == 3 && st / 4 + 1 < NG && seen) n += NA;
        }
        return n;
    }
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
#ifdef WAVE_VM0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif

}

// a.wdp: depthwise weights of channel pairs [CIN/2][10][2] (taps 0-8, bias; the pair interleaved = the SGPR pairs of v_pk_fma_f32)
// a.wpf: pointwise weights, fp16 hi/lo split, in fragment order [CIN/16][COUT/32][hi|lo][64 lanes][8] (lane (r, hi): cout 32*cb + r, channels 16*g + 8*hi + j)
struct WaveArgs {  // (the few fields of DwPwArgs this kernel reads: scalar registers are what limits its weight prefetch)
    const float *in; float *out;
    const float *bp, *wdp;
    const half_t *wpf;
    const float *zeros;  // >= ((CPC - 1) * H * W + W) * 4 + 16 bytes of zeros: the source of input rows outside the image
#ifdef WAVE_STAMP
    long long *stamps;
#endif
};
#ifdef WAVE_STAMP
#define WSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif

template <int CIN, int COUT, int NCB, int H, int W, int STRIDE, int NBUF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(48))) void dwpw_wave_kernel(WaveArgs a) {
    using G = WaveGeom<CIN, NCB, H, W, STRIDE, NBUF>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
    constexpr int HW = G::HW, HoWo = G::HOWO;  // (compile-time: the DMA source offsets are immediates, not live scalar pairs)
    constexpr int tpi = (HoWo + 63) >> 6;      // tiles per image (the last one may be partial)
    const int lid = xcd_tile(gridDim.x);
    const int b = lid / tpi, p0 = (lid - b * tpi) * 64;
    const int y_first = p0 / G::WO;
    const int cob = (int)blockIdx.y * NCB;  // first 32-cout block of this wave

    WSTAMP(0);
    // ---- zero the wave's LDS region once (the pad slots stay zero: the DMA never writes them)
    for (int o = lane * 16; o < G::LDS_BYTES; o += 1024) *reinterpret_cast<floatx4 *>(lds + o) = floatx4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- DMA role: instruction i of an input row covers (channel k, slot s) = divmod(lane + 64 i, SL); slot 0 is the zero pad (masked off).
    //      Rows outside the image come from a buffer of zeros - same number of DMA instructions for every wave (the waits count them)
    unsigned voff[G::NDR];
    bool lval[G::NDR];
#pragma unroll
    for (int i = 0; i < G::NDR; ++i) {
        const int q = lane + 64 * i, k = q / G::SL, sl = q - k * G::SL;
        lval[i] = q < G::CPC * G::SL && sl >= 1;
        voff[i] = lval[i] ? (unsigned)((k * HW + (sl - 1) * 4) * 4) : 0u;
    }
    // one running source pointer per LDS row (uniform), advanced by a chunk of channels per DMA round; rows outside the image stay on the zeros
    const char *rsrc[G::ROWS];
    unsigned rstep[G::ROWS];
    {
        const char *inb = reinterpret_cast<const char *>(a.in + (long)b * CIN * HW);
        const char *zer = reinterpret_cast<const char *>(a.zeros);
#pragma unroll
        for (int j = 0; j < G::ROWS; ++j) {
            const int iy = y_first * STRIDE - 1 + j;
            const bool in = iy >= 0 && iy < H;
            rsrc[j] = in ? inb + iy * W * 4 : zer;
            rstep[j] = in ? (unsigned)(G::CPC * HW * 4) : 0u;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    auto dma_chunk = [&](auto cc) {  // (chunks are requested in order: the pointers advance)
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int i = 0; i < G::NDR; ++i)
            if (lval[i]) {
                const unsigned vo = voff[i];
#pragma unroll
                for (int j = 0; j < G::ROWS; ++j) {
                    const char *src = rsrc[j];
                    const unsigned dst = lds0 + (c % G::NBUF) * G::CHUNKB + j * G::ROWB + i * 1024;
                    // (asm: scalar base + 32-bit lane offset; the builtin took a 64-bit vector add and six scalar instructions per DMA)
                    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(src), "s"(dst) : "memory");
                }
            }
#pragma unroll
        for (int j = 0; j < G::ROWS; ++j) rsrc[j] += rstep[j];
    };

    // ---- stencil role: lane = output pixel p0 + lane (clamped for the tail of an image's last tile; those lanes are not stored)
    const int pl = min(p0 + lane, HoWo - 1);
    const int oy = pl / G::WO, ox = pl - oy * G::WO;
    const unsigned rb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds +
                        (unsigned)(((oy - y_first) * STRIDE) * G::ROWB + (ox * STRIDE + 3) * 4);  // LDS byte address of tap (0, 0), channel 0 of buffer 0

    // ---- pointwise weights: fragment stream.  The MFMAs of group g are issued under the stencil arithmetic of group g + 1, so A(g) stays
    //      live for two groups: two register sets, A(g + 1) requested when the last MFMA of group g - 1 has been issued
    half8 ah[2][NCB], al[2][NCB];
    const half_t *wf = a.wpf + (long)lane * 8;
    const int ncb_total = COUT / 32;
    auto load_a = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const half_t *p = wf + ((long)(g * ncb_total + cob + cb) * 2) * 512;
            ah[g & 1][cb] = *reinterpret_cast<const half8 *>(p);
            al[g & 1][cb] = *reinterpret_cast<const half8 *>(p + 512);
        }
    };

    floatx16 acc[2][NCB];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][cb][e] = 0.f;

    floatx4 bias4[NCB][4];  // output biases of this lane's channels (lane (r, hi): 32 cb + 8 k + 4 hi + 0..3); requested behind the last DMA wait
    WSTAMP(1);
    // prologue.  (The compiler barriers keep the plain loads where the wait counts of younger() assume them: a fragment load hoisted above
    // a DMA it is counted behind makes the count one load too generous - seen: wrong sums that any extra instruction in between cured.)
    asm volatile("" ::: "memory");
    static_for<0, G::LA>(dma_chunk);
    asm volatile("" ::: "memory");
    load_a(std::integral_constant<int, 0>{});
    asm volatile("" ::: "memory");

    // The tap reads are inline asm: ds_read2_b32 with offset1 = the next channel's plane puts (channel c, channel c + 1) of one tap into one
    // register pair = the first source of v_pk_fma_f32.  (Left to the compiler, the vectoriser pairs the two neighbouring taps of ONE
    // channel instead and 1 000 v_mov re-pair them.)  The compiler does not know that the results of an asm read are still in flight: every
    // use sits behind drain(), which names the registers.  A step's reads are issued one step ahead of its arithmetic.
    floatx2 tv[2][2][9];  // [stage = step parity][pair of the step][tap]
    auto issue_taps3 = [&](auto qc, auto dc, floatx2(&v)[9]) {  // q = pair index over the whole channel range, d = tap row (three reads)
        constexpr int q = decltype(qc)::value, d = decltype(dc)::value;
        constexpr int c = q / (G::CPC / 2), pr = q % (G::CPC / 2);
        constexpr int boff = (c % G::NBUF) * G::CHUNKB + 2 * pr * G::CHS + d * G::ROWB;
        const unsigned b0 = rb + boff;
        asm volatile("ds_read2_b32 %0, %3 offset0:0 offset1:%4\n"
                     "ds_read2_b32 %1, %3 offset0:1 offset1:%5\n"
                     "ds_read2_b32 %2, %3 offset0:2 offset1:%6"
                     : "=&v"(v[3 * d]), "=&v"(v[3 * d + 1]), "=&v"(v[3 * d + 2])
                     : "v"(b0), "n"(G::CHS / 4), "n"(G::CHS / 4 + 1), "n"(G::CHS / 4 + 2)
                     : "memory");
    };
    auto issue_taps = [&](auto qc, floatx2(&v)[9]) {
        issue_taps3(qc, std::integral_constant<int, 0>{}, v);
        issue_taps3(qc, std::integral_constant<int, 1>{}, v);
        issue_taps3(qc, std::integral_constant<int, 2>{}, v);
    };
    // The step's wait names ALL eighteen tap registers: a register the wait does not name is, to the compiler, ready since the asm that
    // issued its read, and it did move such registers while the read was in flight (wrong sums in one build, right ones in the next).
    auto landed = [&](floatx2(&u)[9], floatx2(&v)[9]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "+v"(u[8])::"memory");
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8])::"memory");
    };
    // Depthwise weights of a step: 2 pairs x (9 taps + bias) x 2 channels = 40 scalar registers, ONE set in FIXED registers s[56:95] (the
    // kernel is compiled with amdgpu_num_sgpr(48): the compiler stays below them).  Everything that touches them is inline asm - the loads,
    // the waits and the fma chains themselves:  scalar loads and LDS reads share lgkmcnt, so a wait the compiler inserts for a scalar load of
    // its own also drains the NEXT step's tap reads issued in between; and an asm OUTPUT in scalar registers gets copied and spilled by the
    // register allocator while the load is still in flight (both seen: 34 us per wave, wrong sums).
    //   s[56:71] taps 0-7 of pair A (channel pair interleaved: the 64-bit sources of v_pk_fma_f32), s[72:73] tap 8, s[74:75] bias; s[76:95] pair B
    // The next step's weights are requested as soon as this step's chains are through; their latency (a scalar-cache hit) passes under the
    // rest of the step.
#define WAVE_W_CLOBBERS                                                                                                                            \
    "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", \
        "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95"
    const float *wdp = a.wdp;
    auto load_w = [&](auto sc) {
        constexpr int st = decltype(sc)::value;
        const float *wsrc = wdp;  // (asm operands inside a generic lambda must be its own locals)
        asm volatile("s_load_dwordx16 s[56:71], %0, %1\n"
                     "s_load_dwordx4 s[72:75], %0, %2\n"
                     "s_load_dwordx16 s[76:91], %0, %3\n"
                     "s_load_dwordx4 s[92:95], %0, %4" ::"s"(wsrc),
                     "n"(st * 160), "n"(st * 160 + 64), "n"(st * 160 + 80), "n"(st * 160 + 144)
                     : "memory", WAVE_W_CLOBBERS);
    };
    const floatx2 ones = {1.f, 1.f};
    // taps 3 j .. 3 j + 2 of both chains (the first piece starts them from their biases: bias * 1, exact)
    auto chain3 = [&](auto jc, floatx2 &oa, floatx2 &ob, floatx2(&u)[9], floatx2(&v)[9]) {
        constexpr int j = decltype(jc)::value;
        const floatx2 one2 = ones;
        if constexpr (j == 0)
            asm volatile("v_pk_mul_f32 %0, %8, s[74:75]\n"
                         "v_pk_mul_f32 %1, %8, s[94:95]\n"
                         "v_pk_fma_f32 %0, %2, s[56:57], %0\n"
                         "v_pk_fma_f32 %1, %5, s[76:77], %1\n"
                         "v_pk_fma_f32 %0, %3, s[58:59], %0\n"
                         "v_pk_fma_f32 %1, %6, s[78:79], %1\n"
                         "v_pk_fma_f32 %0, %4, s[60:61], %0\n"
                         "v_pk_fma_f32 %1, %7, s[80:81], %1"
                         : "=&v"(oa), "=&v"(ob)
                         : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(one2)
                         : "memory");
        else if constexpr (j == 1)
            asm volatile("v_pk_fma_f32 %0, %2, s[62:63], %0\n"
                         "v_pk_fma_f32 %1, %5, s[82:83], %1\n"
                         "v_pk_fma_f32 %0, %3, s[64:65], %0\n"
                         "v_pk_fma_f32 %1, %6, s[84:85], %1\n"
                         "v_pk_fma_f32 %0, %4, s[66:67], %0\n"
                         "v_pk_fma_f32 %1, %7, s[86:87], %1"
                         : "+v"(oa), "+v"(ob)
                         : "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(v[3]), "v"(v[4]), "v"(v[5])
                         : "memory");
        else
            asm volatile("v_pk_fma_f32 %0, %2, s[68:69], %0\n"
                         "v_pk_fma_f32 %1, %5, s[88:89], %1\n"
                         "v_pk_fma_f32 %0, %3, s[70:71], %0\n"
                         "v_pk_fma_f32 %1, %6, s[90:91], %1\n"
                         "v_pk_fma_f32 %0, %4, s[72:73], %0\n"
                         "v_pk_fma_f32 %1, %7, s[92:93], %1"
                         : "+v"(oa), "+v"(ob)
                         : "v"(u[6]), "v"(u[7]), "v"(u[8]), "v"(v[6]), "v"(v[7]), "v"(v[8])
                         : "memory");
    };

    // Pull the depthwise weight table through the scalar cache now, under the first DMA: a scalar-cache miss goes to L2 (~ 0.5 us)
    {
        const float *wsrc = wdp;
        static_for<0, (CIN * 10 * 4 + 63) / 64>([&](auto ic) {
            unsigned sink;
            const float *src2 = wsrc;
            asm volatile("s_load_dword %0, %1, %2" : "=s"(sink) : "s"(src2), "n"(decltype(ic)::value * 64) : "memory");
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    WSTAMP(2);
    wait_vm<G::younger(0)>();  // chunk 0 has been delivered ...
    asm volatile("s_sleep 2" ::: "memory");  // ... and is readable (see younger())
    WSTAMP(3);
    load_w(std::integral_constant<int, 0>{});
    issue_taps(std::integral_constant<int, 0>{}, tv[0][0]);
    issue_taps(std::integral_constant<int, 1>{}, tv[0][1]);

    unsigned hp[8], lp[8];  // fp16 hi / lo parts of the 16 depthwise outputs of a group, two channels per register
    half8 pbh[2], pbl[2];   // the finished group's B fragments (pixel tiles A / B), consumed by the MFMAs spread over the next group's steps
    constexpr int MF_STEP = 3 * 2 * NCB / 4;
    auto mfma_n = [&](auto gc, auto mc) {  // MFMA m of group g: every accumulator's own order is (a_hi b_hi, a_hi b_lo, a_lo b_hi) as in
        constexpr int g = decltype(gc)::value, m = decltype(mc)::value;  // dwpw_mfma_kernel; dependent MFMAs sit 2 * NCB instructions apart
        constexpr int round = m / (2 * NCB), t = (m / NCB) % 2, cb = m % NCB;
#ifndef WAVE_NO_MFMA
        acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(round == 2 ? al[g & 1][cb] : ah[g & 1][cb], round == 1 ? pbl[t] : pbh[t], acc[t][cb], 0, 0, 0);
#else
        acc[t][cb][round] += (float)pbh[t][cb] + (float)pbl[t][cb] + (float)ah[g & 1][cb][round] + (float)al[g & 1][cb][round];
#endif
    };
    auto step = [&](auto sc) {
        constexpr int st = decltype(sc)::value;
        constexpr int c = st / 2, g = st / 4, k = st % 4;
        if constexpr ((st & 1) == 0 && c + G::LA < G::NCHUNK) dma_chunk(std::integral_constant<int, c + G::LA>{});
        if constexpr ((st & 1) == 0 && c + 1 < G::NCHUNK) wait_vm<G::younger(c + 1)>();  // delivered a step before its first read (see younger())
        if constexpr (st == G::NSTEP - 2) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) bias4[cb][kq] = *reinterpret_cast<const floatx4 *>(a.bp + (cob + cb) * 32 + 8 * kq + 4 * hi);
        }
        WSTAMP(7 + st);
        // Two independent fma chains (one per channel pair), nothing between them: v_pk_fma_f32 does NOT issue under a running MFMA of the
        // same wave (tools/ubench/mfma_pk_overlap.hip: MFMA 35 cycles, six v_pk_fma_f32 37, together 82), LDS reads and plain VALU do (MFMA +
        // six ds_read2_b32: 96 = the reads alone; + six v_fma_f32: 44).  So the step's MFMAs (the PREVIOUS group's) go between the six
        // three-read pieces of the next step's tap reads, and the fp16 split below is written without packed instructions.
        floatx2 oa, ob;
        landed(tv[st & 1][0], tv[st & 1][1]);
        chain3(std::integral_constant<int, 0>{}, oa, ob, tv[st & 1][0], tv[st & 1][1]);
        chain3(std::integral_constant<int, 1>{}, oa, ob, tv[st & 1][0], tv[st & 1][1]);
        chain3(std::integral_constant<int, 2>{}, oa, ob, tv[st & 1][0], tv[st & 1][1]);
        if constexpr (st + 1 < G::NSTEP) load_w(std::integral_constant<int, st + 1>{});
        static_for<0, 6>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (g > 0)
                static_for<i * MF_STEP / 6, (i + 1) * MF_STEP / 6>(
                    [&](auto mc) { mfma_n(std::integral_constant<int, g - 1>{}, std::integral_constant<int, k * MF_STEP + decltype(mc)::value>{}); });
            if constexpr (st + 1 < G::NSTEP) issue_taps3(std::integral_constant<int, 2 * st + 2 + i / 3>{}, std::integral_constant<int, i % 3>{}, tv[(st + 1) & 1][i / 3]);
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            floatx2 o = pr ? ob : oa;
            asm("v_max_f32 %0, 0, %0" : "+v"(o[0]));  // (fmaxf() costs a second, canonicalising v_max per value)
            asm("v_max_f32 %0, 0, %0" : "+v"(o[1]));
            const half2v h = __builtin_convertvector(o, half2v);
            float l0 = (float)h[0], l1 = (float)h[1];
            asm("v_sub_f32 %0, %1, %0" : "+v"(l0) : "v"(o[0]));  // (asm: the compiler would fuse the two subtractions into v_pk_add_f32)
            asm("v_sub_f32 %0, %1, %0" : "+v"(l1) : "v"(o[1]));
            const half2v l = __builtin_convertvector(floatx2{l0, l1}, half2v);
            hp[2 * k + pr] = __builtin_bit_cast(unsigned, h);
            lp[2 * k + pr] = __builtin_bit_cast(unsigned, l);
        }
        if constexpr (k == 3) {
            // lanes 0-31 hold pixel tile A, lanes 32-63 tile B: swap "channels 8-15 of A" with "channels 0-7 of B"
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint2v sh = __builtin_amdgcn_permlane32_swap(hp[i], hp[4 + i], false, false);
                const uint2v sl = __builtin_amdgcn_permlane32_swap(lp[i], lp[4 + i], false, false);
                hp[i] = sh[0];
                hp[4 + i] = sh[1];
                lp[i] = sl[0];
                lp[4 + i] = sl[1];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pbh[t] = __builtin_bit_cast(half8, uint4v{hp[4 * t], hp[4 * t + 1], hp[4 * t + 2], hp[4 * t + 3]});
                pbl[t] = __builtin_bit_cast(half8, uint4v{lp[4 * t], lp[4 * t + 1], lp[4 * t + 2], lp[4 * t + 3]});
            }
            asm volatile("" ::: "memory");
            if constexpr (g + 1 < G::NG) load_a(std::integral_constant<int, g + 1>{});
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    static_for<0, G::NSTEP>(step);
    static_for<0, 4 * MF_STEP>([&](auto mc) { mfma_n(std::integral_constant<int, G::NG - 1>{}, mc); });
    WSTAMP(4);

    // ---- epilogue: lane (r, hi) of tile t owns pixel p0 + 32 t + r and channels 32 cb + (e & 3) + 8 (e >> 2) + 4 hi; conv_dw blocks end in ReLU
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int pix = p0 + 32 * t + r;
        if (pix >= HoWo) continue;
        float *ob = a.out + ((long)b * COUT + cob * 32 + 4 * hi) * HoWo + pix;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[t][cb][e] + bias4[cb][e >> 2][e & 3];
                asm("v_max_f32 %0, 0, %0" : "+v"(v));
                ob[(long)(cb * 32 + (e & 3) + 8 * (e >> 2)) * HoWo] = v;
            }
    }
    WSTAMP(5);
}

template <int CIN, int COUT, int NCB, int H, int W, int STRIDE, int NBUF>
void launch_wave(const DwPwArgs &a, hipStream_t s) {
    using G = WaveGeom<CIN, NCB, H, W, STRIDE, NBUF>;
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dwpw_wave_kernel<CIN, COUT, NCB, H, W, STRIDE, NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  G::LDS_BYTES);
        return true;
    }();
    (void)once;
    constexpr int tpi = (G::HOWO + 63) / 64;
#ifdef WAVE_STAMP
    const WaveArgs w{a.in, a.out, a.bp, a.wdp, a.wpf, a.zeros, reinterpret_cast<long long *>(a.tmp)};
#else
    const WaveArgs w{a.in, a.out, a.bp, a.wdp, a.wpf, a.zeros};
#endif
    hipLaunchKernelGGL((dwpw_wave_kernel<CIN, COUT, NCB, H, W, STRIDE, NBUF>), dim3((unsigned)(a.B * tpi), COUT / (32 * NCB)), dim3(64), G::LDS_BYTES, s, w);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same lane = pixel formulation for the wide, shallow blocks (160x160 and up, <= 32 input channels), where a wave's halo'd LDS image
// would be tens of KB for a handful of channels: the taps come straight from global memory instead - nine buffer loads per channel whose
// per-lane offsets are computed once (a tap outside the image has an out-of-range offset: the buffer returns 0, no selects), the channel is
// the scalar offset.  There are thousands of waves on these maps, so occupancy (not a software pipeline) hides the load latency; everything
// is left to the compiler.  MEASURED at 32 frames (FRT_DWPW_PIX, tuning build): 32 -> 32 at 160x160 75.7 us against 90.0 for dwpw_row4_kernel,
// but 16 -> 32 / stride 2 120.6 against 110.7 and 32 -> 64 / stride 2 57.6 against 43.1 (nine dword loads per channel and pixel: the texture
// path, not HBM, bounds it).  Not bit-identical to the scalar row4 kernel (|d| 2e-5 on the head outputs, the oracle's tolerance holds).  Left
// OFF: 14 us of a 3.1 ms step do not pay for moving the box census.
template <int CIN, int COUT, int STRIDE>
__global__ __launch_bounds__(256) void dwpw_pix_kernel(WaveArgs a, int B, int H, int W) {
    constexpr int NCB = COUT / 32, NG = CIN / 16;
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int Ho = H / STRIDE, Wo = W / STRIDE, HW = H * W, HoWo = Ho * Wo;
    const long total = (long)B * HoWo;
    const long p0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (p0 >= total) return;
    const long pl = min(p0 + lane, total - 1);
    const int b = (int)(pl / HoWo), pp = (int)(pl - (long)b * HoWo), oy = pp / Wo, ox = pp - oy * Wo;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, (int)((long)B * CIN * HW * 4), 0x00020000);
    int vo[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = oy * STRIDE - 1 + t / 3, ix = ox * STRIDE - 1 + t % 3;
        vo[t] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? (int)((((long)b * CIN * H + iy) * W + ix) * 4) : (int)0x80000000;
    }
    const __attribute__((address_space(4))) float *wd = reinterpret_cast<const __attribute__((address_space(4))) float *>(reinterpret_cast<uintptr_t>(a.wdp));
    const half_t *wf = a.wpf + (long)lane * 8;
    floatx16 acc[2][NCB];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][cb][e] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        half8 ah[NCB], al[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            ah[cb] = *reinterpret_cast<const half8 *>(wf + ((long)(g * NCB + cb) * 2) * 512);
            al[cb] = *reinterpret_cast<const half8 *>(wf + ((long)(g * NCB + cb) * 2 + 1) * 512);
        }
        unsigned hp[8], lp[8];
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
            const int q = g * 8 + pr;
            const int so = 2 * q * HW * 4;
            floatx2 o = {wd[q * 20 + 18], wd[q * 20 + 19]};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const floatx2 v = {__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[t], so, 0)),
                                   __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[t], so + HW * 4, 0))};
                o = __builtin_elementwise_fma(v, floatx2{wd[q * 20 + 2 * t], wd[q * 20 + 2 * t + 1]}, o);
            }
            o[0] = fmaxf(o[0], 0.f);
            o[1] = fmaxf(o[1], 0.f);
            const half2v h = __builtin_convertvector(o, half2v);
            const floatx2 back = __builtin_convertvector(h, floatx2);
            const half2v l = __builtin_convertvector(o - back, half2v);
            hp[pr] = __builtin_bit_cast(unsigned, h);
            lp[pr] = __builtin_bit_cast(unsigned, l);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint2v sh = __builtin_amdgcn_permlane32_swap(hp[i], hp[4 + i], false, false);
            const uint2v sl = __builtin_amdgcn_permlane32_swap(lp[i], lp[4 + i], false, false);
            hp[i] = sh[0];
            hp[4 + i] = sh[1];
            lp[i] = sl[0];
            lp[4 + i] = sl[1];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const half8 bh = __builtin_bit_cast(half8, uint4v{hp[4 * t], hp[4 * t + 1], hp[4 * t + 2], hp[4 * t + 3]});
            const half8 bl = __builtin_bit_cast(half8, uint4v{lp[4 * t], lp[4 * t + 1], lp[4 * t + 2], lp[4 * t + 3]});
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bh, acc[t][cb], 0, 0, 0);
                acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bl, acc[t][cb], 0, 0, 0);
                acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], bh, acc[t][cb], 0, 0, 0);
            }
        }
    }
    float bq[NCB][16];  // (in front of the first store: the output may alias them for all the compiler knows - load / wait / store chains otherwise)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) bq[cb][e] = a.bp[cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long px = p0 + 32 * t + r;
        if (px >= total) continue;
        const int b2 = (int)(px / HoWo), p2 = (int)(px - (long)b2 * HoWo);
        float *ob = a.out + ((long)b2 * COUT + 4 * hi) * HoWo + p2;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = cb * 32 + (e & 3) + 8 * (e >> 2);
                ob[(long)co * HoWo] = fmaxf(acc[t][cb][e] + bq[cb][e], 0.f);
            }
    }
}

template <int CIN, int COUT, int STRIDE>
void launch_pix(const DwPwArgs &a, hipStream_t s) {
    const long tiles = ((long)a.B * a.Ho * a.Wo + 63) / 64;
    const WaveArgs w{a.in, a.out, a.bp, a.wdp, a.wpf, a.zeros};
    hipLaunchKernelGGL((dwpw_pix_kernel<CIN, COUT, STRIDE>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, w, a.B, a.H, a.W);
}

}  // namespace

size_t dwpw_wave_zero_bytes() { return (size_t)1 << 20; }  // covers ((CPC - 1) * H * W + W) * 4 + 16 for every shape below

// true: launched.  Shapes covered (640x640 detector input): the five 128 -> 128 blocks at 40x40, 64 -> 64 at 80x80, 256 -> 256 at 20x20
bool launch_dwpw_wave(const DwPwArgs &a, hipStream_t s) {
    static const bool on = !(frt_tuning_env("FRT_DWPW_WAVE") && frt_tuning_env("FRT_DWPW_WAVE")[0] == '0');
    if (!on || !a.wdp || !a.wpf || !a.zeros || a.add || !a.wd || !a.relu) return false;
    if (a.W != a.Wo * a.stride || (a.stride == 1 ? a.H != a.Ho : a.H != 2 * a.Ho)) return false;
    if ((long)a.B * a.Cin * a.H * a.W * 4 >= (1L << 31)) return false;
    static const int which = frt_tuning_env("FRT_DWPW_WAVE_SHAPES") ? atoi(frt_tuning_env("FRT_DWPW_WAVE_SHAPES")) : 7;  // bit per shape (A/B measurements)
    static const bool any_batch = frt_tuning_env("FRT_DWPW_WAVE_ANYB") != nullptr;
    // Batch thresholds: a wave is ~ 13 / 8 / 17 us long whatever the batch (one wave = 64 pixels x every input channel), the workgroup-tiled
    // kernel finishes a few frames sooner.  Measured per launch, old / this kernel (us):  128 @ 40x40: 11.6 / 14 at 4 frames, 14.7 / 16 at 8,
    // 24.4 / 18 at 16, 39 / 20 at 32;  64 @ 80x80: 10.1 / 10 at 1, 12.9 / 11 at 4, 20.3 / 14 at 8, 56 / 36 at 32;  256 @ 20x20: 19 / 21 at 4,
    // 25.5 / 24 at 8, 27.8 / 25 at 16, 36 / 22 at 32 (profiles/r04/r04t_dwpw_wave.txt)
    if (a.stride == 1 && a.Cin == 128 && a.Cout == 128 && a.W == 40 && a.H == 40 && (which & 1) && (a.B >= 12 || any_batch)) {
        launch_wave<128, 128, 4, 40, 40, 1, 3>(a, s);
        return true;
    }
    if (a.stride == 1 && a.Cin == 64 && a.Cout == 64 && a.W == 80 && a.H == 80 && (which & 2) && (a.B >= 3 || any_batch)) {
        launch_wave<64, 64, 2, 80, 80, 1, 3>(a, s);
        return true;
    }
    if (a.stride == 1 && a.Cin == 256 && a.Cout == 256 && a.W == 20 && a.H == 20 && (which & 4) && (a.B >= 12 || any_batch)) {
        launch_wave<256, 256, 2, 20, 20, 1, 4>(a, s);  // four waves per pixel tile, 64 output channels each (only 224 tiles at 32 frames)
        return true;
    }
    static const int pix = frt_tuning_env("FRT_DWPW_PIX") ? atoi(frt_tuning_env("FRT_DWPW_PIX")) : 0;  // (experiment: default off until measured)
    if ((pix & 1) && a.stride == 1 && a.Cin == 32 && a.Cout == 32) {
        launch_pix<32, 32, 1>(a, s);
        return true;
    }
    if ((pix & 2) && a.stride == 2 && a.Cin == 16 && a.Cout == 32) {
        launch_pix<16, 32, 2>(a, s);
        return true;
    }
    if ((pix & 4) && a.stride == 2 && a.Cin == 32 && a.Cout == 64) {
        launch_pix<32, 64, 2>(a, s);
        return true;
    }
    return false;
}
