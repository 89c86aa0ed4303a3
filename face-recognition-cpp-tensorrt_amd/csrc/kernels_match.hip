// Cosine-similarity GEMM with a fused top-1 epilogue (and a full-matrix variant) for gfx950.
//
// Replaces MatMul::calculate (cuBLASLt fp32 GEMM, /root/reference/src/matmul.cpp:36-77) and the O(F*N) host argmax of
// ArcFaceIR50::getOutputs (/root/reference/src/arcface.cpp:203-217).  S[q][g] = sum_k E[q][k] * G[g][k] in exact fp32:
// v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain, so duplicate gallery rows give bit-identical similarities wherever
// they sit, and "first maximum wins" (std::max_element) is decidable on the index alone.
//
// Roofline: the gallery (4*512*N bytes) is streamed from HBM exactly once per call.  For F < ~40 queries the kernel is
// HBM-bound; for the 128-face batch it is bound by the fp32 matrix rate (157 TF/s), 2*512*N*F flop.
//
// Tiling: workgroup = 4 waves, 128 gallery rows x NQ*32 queries per tile, K in steps of 32 floats staged through LDS with
// coalesced 16-byte loads (8 lanes cover one 128-byte row segment).  LDS rows are 128 B; the 16-byte chunk index is XOR-
// swizzled with (row>>1)&7 so that every 16-lane group of a ds_read_b128 touches 16 distinct 16-byte slots of the 256-byte
// bank row (conflict-free, MI355X LDS rules).  The k index inside a chunk is permuted consistently for both operands
// (lane>>5 picks the chunk, the 4 MFMAs of a chunk walk its 4 floats), which leaves every dot product unchanged.
#include <cstdlib>

#include "frt_kernels.h"

constexpr int CTL_WORDS = FRT_MATCH_CTL_WORDS;  // control words of the fast screened path (cleared by the coarse scan; layout at CTL_OVERFLOW below; frt_matcher.hpp allocates them)

#include <limits.h>

namespace {

constexpr int BM = 128;  // gallery rows per tile
constexpr int BK = 32;   // floats per k-step

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return (v > bv) || (v == bv && i < bi); }

// GT: storage type of the gallery rows (float: the reference's row-major layout; half_t: the fp16 gallery - shadow copy for the
// screening pass or the fp16-STORED shard of BASELINE config 5; stored values are widened exactly to fp32 on the way into LDS, all
// arithmetic stays the same fp32 fmaf chain).
//
// The fp16 gallery is NOT row-major: it is kept in the order the coarse kernel's MFMA A fragments consume it, so that every
// wave-level load of the 1 GB scan is ONE contiguous kilobyte (8 full 128-byte lines).  Row g, column k lives at
//   ((((g >> 7) * 4 + ((g >> 5) & 3)) * (D / 16) + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (g & 31)) * 8 + (k & 7)
// i.e. [128-row tile][32-row wave block][16-wide k step][lane = (k half, row)][8 halfs].  (Row-major rows made a lane fetch 16 bytes
// of its own 1 KB row per instruction: 32 different lines per load, each line requested four times - both earlier coarse kernels
// stalled at 4.0 TB/s on it.)  The allocation is padded to whole 128-row tiles; pad rows are zero.
__device__ __forceinline__ long g16_index(long g, int k, int D) {
    return ((((g >> 7) * 4 + ((g >> 5) & 3)) * (long)(D >> 4) + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (g & 31)) * 8 + (k & 7);
}
__device__ __forceinline__ floatx4 load_g4(const float *G, long g, int D, int k) { return *reinterpret_cast<const floatx4 *>(G + g * D + k); }
__device__ __forceinline__ floatx4 load_g4(const half_t *G, long g, int D, int k) {  // k % 4 == 0: four columns never straddle an 8-group
    const half4 h = *reinterpret_cast<const half4 *>(G + g16_index(g, k, D));
    return floatx4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// EXCL (top-k passes): only rows that come strictly AFTER the query's previous winner (prev_sim, prev_idx = its GLOBAL index) in the
// result order "higher similarity first, lower index first on ties" take part; prev_sim == nullptr: no previous winner (first pass).
// The similarities are the same accumulators either way, so pass j of a top-k search returns the j-th entry of the exact ranking.
template <int NQ, bool FULL, typename GT = float, bool EXCL = false>
__global__ __launch_bounds__(256) void match_kernel(const GT *__restrict__ G, int N, int D, const float *__restrict__ E, int F,
                                                    MatchPartial *__restrict__ partial, float *__restrict__ out_full, int num_tiles,
                                                    int row_offset, const int *__restrict__ tile_list, const int *__restrict__ d_num_tiles,
                                                    const float *__restrict__ prev_sim = nullptr, const int32_t *__restrict__ prev_idx = nullptr,
                                                    int prev_stride = 1, const int *__restrict__ gate = nullptr) {
    if (gate && !*gate) return;  // fast top-1 path: this launch only runs when the pair list overflowed (uniform; before any barrier)
    // tile_list != nullptr: run only over the listed 128-row gallery tiles (num_tiles = list length): the exact re-rank pass of
    // the screened top-1 (same code path per tile as the full scan -> bitwise-identical similarities)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);                        // [2][BM][BK]
    float *Bs = reinterpret_cast<float *>(smem) + 2 * BM * BK;          // [2][NQ*32][BK]
    constexpr int QT = NQ * 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int q0 = blockIdx.y * QT;
    const int ksteps = D / BK;

    if (tile_list) num_tiles = *d_num_tiles;  // list length produced on the device by the screening pass
    const int my_tiles = num_tiles > (int)blockIdx.x ? (num_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int total = my_tiles * ksteps;

    const int ld_row = tid >> 3, ld_ch = tid & 7;

    floatx4 ga[4], qa[NQ];
    auto load_global = [&](int it) {
        int tile = blockIdx.x + (it / ksteps) * gridDim.x;
        if (tile_list) tile = tile_list[tile];
        const int k0 = (it % ksteps) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ld_row + 32 * i;
            const long g = (long)tile * BM + row;
            ga[i] = g < N ? load_g4(G, g, D, k0 + ld_ch * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = q0 + ld_row + 32 * i;
            qa[i] = q < F ? *reinterpret_cast<const floatx4 *>(E + (long)q * D + k0 + ld_ch * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ld_row + 32 * i;
            *reinterpret_cast<floatx4 *>(As + buf * BM * BK + row * BK + swz(row, ld_ch) * 4) = ga[i];
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int row = ld_row + 32 * i;
            *reinterpret_cast<floatx4 *>(Bs + buf * QT * BK + row * BK + swz(row, ld_ch) * 4) = qa[i];
        }
    };

    floatx16 acc[NQ];
    float bv[NQ];
    int bi[NQ];
    float ps[NQ];  // EXCL: previous winner of this lane's query n (query q0 + n*32 + r)
    int pi[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        ps[n] = INFINITY;
        pi[n] = -1;
        if (EXCL && prev_sim) {
            const int q = q0 + n * 32 + r;
            if (q < F) {
                ps[n] = prev_sim[(long)q * prev_stride];
                pi[n] = prev_idx[(long)q * prev_stride];
            }
        }
        bv[n] = -INFINITY;
        bi[n] = INT_MAX;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    }

    if (total > 0) {
        load_global(0);
        store_lds(0);
    }
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < total; ++it) {
        if (it + 1 < total) load_global(it + 1);
        const float *Ab = As + cur * BM * BK;
        const float *Bb = Bs + cur * QT * BK;
        const int arow = wave * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ks * 2 + hi;
            const floatx4 a4 = *reinterpret_cast<const floatx4 *>(Ab + arow * BK + swz(arow, ch) * 4);
            floatx4 b4[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                const int brow = n * 32 + r;
                b4[n] = *reinterpret_cast<const floatx4 *>(Bb + brow * BK + swz(brow, ch) * 4);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    if (FULL)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[n][s], a4[s], acc[n], 0, 0, 0);
                    else
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[n][s], acc[n], 0, 0, 0);
                }
            }
        }
        if (it + 1 < total) store_lds(cur ^ 1);
        if ((it % ksteps) == ksteps - 1) {  // tile finished: epilogue
            int tile = blockIdx.x + (it / ksteps) * gridDim.x;
            if (tile_list) tile = tile_list[tile];
            const int gbase = tile * BM + wave * 32;
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (FULL) {
                        const int q = q0 + n * 32 + rr;
                        const int g = gbase + r;
                        if (q < F && g < N) out_full[(long)q * N + g] = acc[n][e];
                    } else {
                        const int g = gbase + rr;
                        const float v = acc[n][e];
                        bool ok = g < N;
                        if (EXCL) ok = ok && ((v < ps[n]) || (v == ps[n] && g + row_offset > pi[n]));
                        if (ok && better(v, g, bv[n], bi[n])) {
                            bv[n] = v;
                            bi[n] = g;
                        }
                    }
                    acc[n][e] = 0.f;
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    if (!FULL) {
        // lane halves -> waves -> one partial per (workgroup, query)
        float *red_v = reinterpret_cast<float *>(smem);  // safe: all tile reads are behind the final barrier above
        int *red_i = reinterpret_cast<int *>(smem) + 4 * QT;
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const float ov = __shfl_xor(bv[n], 32);
            const int oi = __shfl_xor(bi[n], 32);
            if (better(ov, oi, bv[n], bi[n])) {
                bv[n] = ov;
                bi[n] = oi;
            }
            if (hi == 0) {
                red_v[wave * QT + n * 32 + r] = bv[n];
                red_i[wave * QT + n * 32 + r] = bi[n];
            }
        }
        __syncthreads();
        if (tid < QT) {
            float v = red_v[tid];
            int i = red_i[tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float ov = red_v[w * QT + tid];
                const int oi = red_i[w * QT + tid];
                if (better(ov, oi, v, i)) {
                    v = ov;
                    i = oi;
                }
            }
            const int q = q0 + tid;
            if (q < F) {
                MatchPartial p;
                p.sim = v;
                p.idx = i == INT_MAX ? -1 : i + row_offset;
                partial[(long)blockIdx.x * F + q] = p;  // (blocks beyond a short tile list write the empty record)
            }
        }
    }
}

// one wave per query: lanes stride over the workgroup partials, then a butterfly with the same first-maximum rule
__global__ __launch_bounds__(64) void match_reduce_kernel(const MatchPartial *__restrict__ partial, int blocks, int F,
                                                          int32_t *__restrict__ idx_out, float *__restrict__ sim_out, int out_stride = 1,
                                                          const int *__restrict__ gate = nullptr) {
    if (gate && !*gate) return;
    const int q = blockIdx.x;
    float v = -INFINITY;
    int i = INT_MAX;
    for (int b = threadIdx.x; b < blocks; b += 64) {
        const MatchPartial p = partial[(long)b * F + q];
        if (p.idx >= 0 && better(p.sim, p.idx, v, i)) {
            v = p.sim;
            i = p.idx;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        if (better(ov, oi, v, i)) {
            v = ov;
            i = oi;
        }
    }
    if (threadIdx.x == 0) {
        idx_out[(long)q * out_stride] = i == INT_MAX ? -1 : i;
        sim_out[(long)q * out_stride] = v;
    }
}

// ---------------------------------------------------------------- screened top-1: fp16 coarse pass + exact re-rank of a few tiles
// The exact fp32 scan above is bound by the fp32 matrix rate for a 128-face batch (2*512*N*F flop at <= 157 TF/s).  Screening
// makes the common case HBM-bound instead without changing a single result bit:
//   1. coarse: S~[q][g] on v_mfma_f32_32x32x16_f16 from an fp16 shadow copy of the gallery (half the bytes, 16x the matrix
//      rate; kept in MFMA-fragment order so that the scan is made of contiguous 1 KB loads); only the maximum per (query, 32-row
//      block of a 128-row tile) is kept;
//   2. select: with fp16-rounded inputs and fp32 accumulation |S~ - S| <= delta_q = 1.2e-3 * ||q|| * max_g ||g|| (2^-10 from the
//      two roundings via Cauchy-Schwarz, plus accumulation slack), so every row that attains the exact maximum of query q lives
//      in a tile whose coarse maximum is >= (best coarse maximum of q) - 2*delta_q.  Those tiles go on a list (each once);
//   3. exact: match_kernel runs over the listed tiles only, for all queries - the same code path per tile as the full scan,
//      so similarities are bitwise those of the full scan and the first-index tie rule is unchanged.  Extra tiles are harmless.

// (also clears `nzero` ints at `zero`: the screened search's tile flags + candidate count, which used to be two memset launches)
__global__ __launch_bounds__(256) void to_half_kernel(const float *__restrict__ in, half_t *__restrict__ out, long n8, int *__restrict__ zero, long nzero) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nzero) zero[i] = 0;
    if (i >= n8) return;
    const floatx4 a = *reinterpret_cast<const floatx4 *>(in + i * 8), b = *reinterpret_cast<const floatx4 *>(in + i * 8 + 4);
    half8 o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
    *reinterpret_cast<half8 *>(out + i * 8) = o;
}

// fp32 rows [n_rows][D] (row-major, row 0 = global row row0 of the gallery) -> fragment-ordered fp16 gallery.  thread = one 16-byte
// output piece (8 halfs); consecutive threads write consecutive pieces.
__global__ __launch_bounds__(256) void gallery_to_half_kernel(const float *__restrict__ in, long row0, long n_rows, int D, half_t *__restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;  // piece index relative to the first tile touched
    const int KS = D >> 4;
    const long tile0 = row0 >> 7;
    const long piece = t + tile0 * 4 * KS * 64;           // absolute piece index in the fragment-ordered gallery
    const long blk = piece >> 6;                           // (tile, wave, kstep)
    const int lane = (int)(piece & 63), hi = lane >> 5, r = lane & 31;
    const int ks = (int)(blk % KS);
    const long tw = blk / KS;
    const long g = (tw >> 2) * 128 + (tw & 3) * 32 + r;
    if (g < row0 || g >= row0 + n_rows) return;           // rows of other chunks / the zero padding of the last tile
    const float *src = in + (g - row0) * D + ks * 16 + hi * 8;
    const floatx4 a = *reinterpret_cast<const floatx4 *>(src), b = *reinterpret_cast<const floatx4 *>(src + 4);
    half8 o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
    *reinterpret_cast<half8 *>(out + piece * 8) = o;
}

// max over rows of ||g||^2 (non-negative floats order like their bit patterns -> atomicMax on the int view); one wave per row
template <typename GT>
__global__ __launch_bounds__(256) void row_norm_max_kernel(const GT *__restrict__ G, int N, int D, int *__restrict__ out_bits) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float s = 0.f;
    for (int k = lane * 4; k < D && row < N; k += 256) {
        const floatx4 v = load_g4(G, (long)row, D, k);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    __shared__ float wmax[4];
    if (lane == 0) wmax[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_int(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// ---------------------------------------------------------------- coarse pass: queries resident in LDS, gallery fragments straight to registers
// A persistent workgroup (one per CU) keeps the whole fp16 query block [128][D] in LDS (133 KB at D = 512, rows padded by 16 B:
// conflict-free ds_read_b128) for its lifetime and streams its gallery tiles - 128 rows, one 32-row MFMA block per wave - from HBM
// straight into A-fragment registers: thanks to the fragment-ordered gallery layout every load instruction of a wave is one contiguous
// kilobyte.  The A registers form a ring exactly one tile deep: register ks is consumed by the MFMAs of k-step ks and immediately
// refilled with the same k-step of the NEXT tile, i.e. every load has a full tile of MFMAs (4096 clk) to land, with 128 KB per CU in
// flight.  LDS carries only the query fragments.  No barrier in the loop (a barrier per tile stalls the load stream - loads are only
// issued from MFMA steps - and cost ~20 % of the bandwidth): every wave writes its own maximum, four coarse entries per tile.
// Round 3 (fast top-1 path): Q32 != nullptr -> the fp32 queries are rounded to fp16 on their way into LDS (the separate conversion
// launch disappears); wgmax != nullptr -> every workgroup also leaves the maximum of ALL its coarse entries per query in
// wgmax[blockIdx.x][query] (plain stores; the selection kernel takes the maximum of those <= 256 values instead of a separate
// pass over the 31 252 entries per query); ctl / qkey: the pair counter, the overflow flag and the packed per-query results of the
// scalar re-rank are cleared here, i.e. in front of every kernel of this call that touches them.
template <int D>
__global__ __launch_bounds__(256) void match_coarse_kernel(const half_t *__restrict__ G, int N, const half_t *__restrict__ Q, int F,
                                                           float *__restrict__ tilemax, int num_tiles, const float *__restrict__ Q32 = nullptr,
                                                           float *__restrict__ wgmax = nullptr, int *__restrict__ ctl = nullptr,
                                                           unsigned long long *__restrict__ qkey = nullptr) {
    constexpr int KS = D / 16;   // k-steps
    constexpr int QP = D + 8;    // halves per query row in LDS
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    half_t *Qs = reinterpret_cast<half_t *>(smem2);                        // [128][QP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int q0 = blockIdx.y * 128;
    if (ctl && blockIdx.x == 0 && blockIdx.y == 0) {
        for (int i = tid; i < CTL_WORDS; i += 256) ctl[i] = 0;
        for (int i = tid; i < F; i += 256) qkey[i] = 0ull;
    }
    for (int i = tid; i < 128 * (D / 8); i += 256) {
        const int q = i / (D / 8), c = i - q * (D / 8);
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q0 + q < F) {
            if (Q32) {
                const floatx4 a = *reinterpret_cast<const floatx4 *>(Q32 + (long)(q0 + q) * D + c * 8), b = *reinterpret_cast<const floatx4 *>(Q32 + (long)(q0 + q) * D + c * 8 + 4);
                v = half8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
            } else {
                v = *reinterpret_cast<const half8 *>(Q + (long)(q0 + q) * D + c * 8);
            }
        }
        *reinterpret_cast<half8 *>(Qs + q * QP + c * 8) = v;
    }
    float rmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // this wave's running maximum per query (NaN entries are tracked apart)
    bool rnan = false;
    int tile = blockIdx.x;
    if (tile >= num_tiles) {  // (uniform per workgroup; no barrier has been executed yet)
        if (wgmax)
            for (int i = tid; i < 128; i += 256)
                if (q0 + i < F) wgmax[(long)blockIdx.x * F + q0 + i] = -INFINITY;
        return;
    }
    auto frag_ptr = [&](int t) { return G + (((long)t * 4 + wave) * KS) * 512 + lane * 8; };  // + ks * 512 halfs (1 KB) per k-step
    half8 areg[KS];
    {
        const half_t *gp = frag_ptr(tile);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) areg[ks] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(gp + ks * 512));
    }
    __syncthreads();
    const half_t *qb = Qs + r * QP + 8 * hi;
    for (; tile < num_tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const half_t *gn = frag_ptr(next < num_tiles ? next : tile);
        floatx16 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
        half8 bq[2][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bq[0][n] = *reinterpret_cast<const half8 *>(qb + n * 32 * QP);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int n = 0; n < 4; ++n) bq[(ks + 1) & 1][n] = *reinterpret_cast<const half8 *>(qb + n * 32 * QP + (ks + 1) * 16);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[ks], bq[ks & 1][n], acc[n], 0, 0, 0);
            // refill with the next tile's fragment; streamed once per call: non-temporal, so that the 1 GB scan does not push the
            // recogniser's working set out of L2 / MALL
            areg[ks] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(gn + ks * 512));
        }
        __builtin_amdgcn_sched_barrier(0);
        const int gbase = tile * 128 + wave * 32;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int g = gbase + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (g < N) m = fmaxf(m, acc[n][e]);  // (pad rows of the last tile are excluded)
            }
            m = fmaxf(m, __shfl_xor(m, 32));
            const int q = q0 + n * 32 + r;
            if (hi == 0 && q < F) tilemax[((long)q * num_tiles + tile) * 4 + wave] = m;
            rnan = rnan || (m != m);
            rmax[n] = fmaxf(rmax[n], m);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the iterations apart (without it the scheduler's version needs 512 registers + spills)
    }
    if (wgmax) {  // the four waves' maxima -> one value per (workgroup, query); a NaN anywhere makes it NaN (the selection then takes every tile)
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem2);  // the query block is dead now
        if (hi == 0) {
#pragma unroll
            for (int n = 0; n < 4; ++n) red[wave * 128 + n * 32 + r] = rnan ? NAN : rmax[n];
        }
        __syncthreads();
        if (tid < 128 && q0 + tid < F) {
            float m = red[tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float o = red[w * 128 + tid];
                m = (m != m || o != o) ? NAN : fmaxf(m, o);
            }
            wgmax[(long)blockIdx.x * F + q0 + tid] = m;
        }
    }
}

// ---------------------------------------------------------------- int8 shadow gallery (round 4)
// The coarse scan is HBM-bound: one pass over the shadow gallery per call, whatever the number of queries (1.02 GB as fp16 at N = 1M:
// 0.20 - 0.24 ms of a 3.2 ms step, and 0.20 of a 0.80 ms four-frame step).  The screening only needs a similarity that is within a KNOWN
// distance of the exact one, so the shadow can be coarser than fp16 as long as its error is bounded: rows are stored as int8 with a
// per-row scale, row = scale * q8 + e, and the largest error norm E = max_r ||e_r|| is measured when the shadow is built.  Then
//   |S~ - S| <= ||q16 - q|| * ||g|| + ||q16|| * ||e|| + accumulation  <=  ||q|| * (0.7e-3 * max||g|| + 1.02 * E)
// (Cauchy-Schwarz; the query is still rounded to fp16, the products q16 * q8 are exact in fp32).  For unit-norm 512-d rows E is ~ 0.01, the
// band 2 * delta ~ 0.02 keeps a handful of tiles per query, and the exact pair re-rank behind it is unchanged - so are the results, bit
// for bit.  Half the bytes: 0.51 GB per scan.  The matrix cores still run fp16 x fp16: the bytes are widened in registers - a byte u =
// q8 + 128 becomes the fp16 bit pattern 0x6400 | u = 1024 + u (one v_perm_b32 per two values), minus 1152 (one v_pk_add_f16 per two).
// Layout: [128-row tile][32-row wave block][32-wide k pair][lane = (k half, row)][16 bytes = k-step 2p, k-step 2p + 1]: every load of a
// wave is one contiguous kilobyte covering two MFMA k-steps.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half8 i8x8_to_half8(unsigned d0, unsigned d1) {
    union {
        unsigned u[4];
        half2_t h[4];
        half8 v;
    } x;
    x.u[0] = __builtin_amdgcn_perm(0x64646464u, d0, 0x04010400u);
    x.u[1] = __builtin_amdgcn_perm(0x64646464u, d0, 0x04030402u);
    x.u[2] = __builtin_amdgcn_perm(0x64646464u, d1, 0x04010400u);
    x.u[3] = __builtin_amdgcn_perm(0x64646464u, d1, 0x04030402u);
    const half2_t bias = {(_Float16)-1152.0f, (_Float16)-1152.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) x.h[i] = x.h[i] + bias;
    return x.v;
}

// one wave per gallery row (D = 512: 8 values per lane): scale = max|g| / 127, q = rint(g / scale), error norm^2 and row norm^2 -> atomicMax
__global__ __launch_bounds__(256) void gallery_to_i8_kernel(const float *__restrict__ in, int N, uint8_t *__restrict__ out, float *__restrict__ scale,
                                                            int *__restrict__ max_err2_bits, int *__restrict__ max_norm2_bits) {
    constexpr int D = 512;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= N) return;
    const float *row = in + (long)g * D + lane * 8;
    const floatx4 a = *reinterpret_cast<const floatx4 *>(row), b = *reinterpret_cast<const floatx4 *>(row + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    float m = 0.f, n2 = 0.f;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        m = fmaxf(m, fabsf(v[e]));
        n2 += v[e] * v[e];
        bad = bad || !(fabsf(v[e]) < INFINITY);
    }
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off));
        n2 += __shfl_xor(n2, off);
        bad = bad || __shfl_xor((int)bad, off);
    }
    const float sc = m > 0.f ? m / 127.f : 0.f, inv = m > 0.f ? 127.f / m : 0.f;
    float e2 = 0.f;
    unsigned long long packed = 0ull;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float q = rintf(v[e] * inv);
        q = fminf(fmaxf(q, -127.f), 127.f);
        if (bad) q = 0.f;
        const float d = v[e] - sc * q;
        e2 += d * d;
        packed |= (unsigned long long)(unsigned)((int)q + 128) << (8 * e);
    }
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    // k = lane * 8 + e: k-step ks = lane >> 1, k half hi = lane & 1
    const int ks = lane >> 1, hi = lane & 1;
    const long off = ((((long)(g >> 7) * 4 + ((g >> 5) & 3)) * (D / 32) + (ks >> 1)) * 64 + hi * 32 + (g & 31)) * 16 + (ks & 1) * 8;
    *reinterpret_cast<unsigned long long *>(out + off) = packed;
    if (lane == 0) {
        scale[g] = bad ? 0.f : sc;
        // a non-finite row has no bound: NaN poisons the maxima (the NaN bit pattern is above every finite float's) and the selection
        // then takes every tile, which sends the call through the exact scan
        atomicMax(max_err2_bits, __float_as_int(bad ? NAN : e2 * 1.0001f));
        atomicMax(max_norm2_bits, __float_as_int(bad ? NAN : n2));
    }
}

// match_coarse_kernel over the int8 shadow (D = 512): same persistent structure, same outputs (per-wave maxima per tile, per-workgroup
// maxima per query); the A ring holds 16 x 16 bytes per lane = one tile, each register feeds two k-steps after widening.
// NQB = 32-query blocks per workgroup (1, 2 or 4): a call with few queries - one frame's faces, a four-frame batch - neither computes nor
// keeps in LDS the empty query blocks, and two to four workgroups then share a CU (with all four blocks the scan is not HBM-bound:
// 173 us for 0.51 GB; with one block it is).
template <int NQB>
__global__ __launch_bounds__(256) void match_coarse_i8_kernel(const uint8_t *__restrict__ G8, const float *__restrict__ gscale, int N, int F,
                                                              float *__restrict__ tilemax, int num_tiles, const float *__restrict__ Q32,
                                                              float *__restrict__ wgmax, int *__restrict__ ctl, unsigned long long *__restrict__ qkey) {
    constexpr int D = 512, KS = D / 16, KP = D / 32, QP = D + 8;
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    half_t *Qs = reinterpret_cast<half_t *>(smem2);  // [128][QP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int q0 = blockIdx.y * (32 * NQB);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int i = tid; i < CTL_WORDS; i += 256) ctl[i] = 0;
        for (int i = tid; i < F; i += 256) qkey[i] = 0ull;
    }
    for (int i = tid; i < 32 * NQB * (D / 8); i += 256) {
        const int q = i / (D / 8), c = i - q * (D / 8);
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q0 + q < F) {
            const floatx4 a = *reinterpret_cast<const floatx4 *>(Q32 + (long)(q0 + q) * D + c * 8), b = *reinterpret_cast<const floatx4 *>(Q32 + (long)(q0 + q) * D + c * 8 + 4);
            v = half8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
        }
        *reinterpret_cast<half8 *>(Qs + q * QP + c * 8) = v;
    }
    float rmax[NQB];
#pragma unroll
    for (int n = 0; n < NQB; ++n) rmax[n] = -INFINITY;
    bool rnan = false;
    int tile = blockIdx.x;
    if (tile >= num_tiles) {
        for (int i = tid; i < 32 * NQB; i += 256)
            if (q0 + i < F) wgmax[(long)blockIdx.x * F + q0 + i] = -INFINITY;
        return;
    }
    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    auto frag_ptr = [&](int t) { return reinterpret_cast<const uint4_t *>(G8 + (((long)t * 4 + wave) * KP) * 1024) + lane; };  // + kp * 64 (1 KB) per k pair
    auto scale_ptr = [&](int t) { return gscale + (long)t * 128 + wave * 32 + 4 * hi; };                                        // + 8 j: rows 8 j + 4 hi + (0..3)
    uint4_t areg[KP];
    floatx4 sreg[4];
    {
        const uint4_t *gp = frag_ptr(tile);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) areg[kp] = __builtin_nontemporal_load(gp + kp * 64);
        const float *sp = scale_ptr(tile);
#pragma unroll
        for (int j = 0; j < 4; ++j) sreg[j] = *reinterpret_cast<const floatx4 *>(sp + 8 * j);
    }
    __syncthreads();
    const half_t *qb = Qs + r * QP + 8 * hi;
    for (; tile < num_tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const int nt = next < num_tiles ? next : tile;
        const uint4_t *gn = frag_ptr(nt);
        floatx16 acc[NQB];
#pragma unroll
        for (int n = 0; n < NQB; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
        // Software-pipelined by hand, one k-step ahead: the query fragments AND the widened gallery fragment of step ks + 1 are produced
        // before the MFMAs of step ks, and sched_barrier pins that order.  Left to the compiler the LDS reads sat directly in front of the
        // MFMAs that consume them (two to three exposed LDS latencies per k-step with one wave per SIMD: 370 cycles per step, the scan at
        // 2.95 TB/s with the matrix pipe a third busy).
        if constexpr (NQB == 4) {
            constexpr int LA = 1;  // k-steps of lookahead (ring of LA + 1); 2 measured the same (165 against 163 us)
            half8 bq[LA + 1][NQB], af[LA + 1];
            auto produce = [&](int k) {  // query fragments + widened gallery fragment of k-step k into ring slot k % (LA + 1)
#pragma unroll
                for (int n = 0; n < NQB; ++n) bq[k % (LA + 1)][n] = *reinterpret_cast<const half8 *>(qb + n * 32 * QP + k * 16);
                const uint4_t raw = areg[k >> 1];
                af[k % (LA + 1)] = (k & 1) ? i8x8_to_half8(raw[2], raw[3]) : i8x8_to_half8(raw[0], raw[1]);
                // (k odd: both k-steps of register k >> 1 are widened now - refill it with the next tile's)
                if (k & 1) areg[k >> 1] = __builtin_nontemporal_load(gn + (k >> 1) * 64);
            };
#pragma unroll
            for (int k = 0; k < LA; ++k) produce(k);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + LA < KS) produce(ks + LA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NQB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks % (LA + 1)], bq[ks % (LA + 1)][n], acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {  // one or two query blocks: HBM-bound either way, and the compiler's own schedule measured faster (64 queries: 164 against 182 us per call)
            half8 bq[2][NQB];
#pragma unroll
            for (int n = 0; n < NQB; ++n) bq[0][n] = *reinterpret_cast<const half8 *>(qb + n * 32 * QP);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int n = 0; n < NQB; ++n) bq[(ks + 1) & 1][n] = *reinterpret_cast<const half8 *>(qb + n * 32 * QP + (ks + 1) * 16);
                }
                const uint4_t raw = areg[ks >> 1];
                const half8 af = (ks & 1) ? i8x8_to_half8(raw[2], raw[3]) : i8x8_to_half8(raw[0], raw[1]);
#pragma unroll
                for (int n = 0; n < NQB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bq[ks & 1][n], acc[n], 0, 0, 0);
                if (ks & 1) areg[ks >> 1] = __builtin_nontemporal_load(gn + (ks >> 1) * 64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int gbase = tile * 128 + wave * 32;
#pragma unroll
        for (int n = 0; n < NQB; ++n) {
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int g = gbase + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (g < N) m = fmaxf(m, acc[n][e] * sreg[e >> 2][e & 3]);
            }
            m = fmaxf(m, __shfl_xor(m, 32));
            const int q = q0 + n * 32 + r;
            if (hi == 0 && q < F) tilemax[((long)q * num_tiles + tile) * 4 + wave] = m;
            rnan = rnan || (m != m);
            rmax[n] = fmaxf(rmax[n], m);
        }
        {
            const float *sp = scale_ptr(nt);
#pragma unroll
            for (int j = 0; j < 4; ++j) sreg[j] = *reinterpret_cast<const floatx4 *>(sp + 8 * j);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem2);
    if (hi == 0) {
#pragma unroll
        for (int n = 0; n < NQB; ++n) red[wave * (32 * NQB) + n * 32 + r] = rnan ? NAN : rmax[n];
    }
    __syncthreads();
    if (tid < 32 * NQB && q0 + tid < F) {
        float m = red[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float o = red[w * (32 * NQB) + tid];
            m = (m != m || o != o) ? NAN : fmaxf(m, o);
        }
        wgmax[(long)blockIdx.x * F + q0 + tid] = m;
    }
}

// The same scan for a full 128-query block with a 2 x 2 wave tiling: wave = (row half rb, query half qh) owns the 32-row blocks 2 rb, 2 rb + 1
// and the query blocks 2 qh, 2 qh + 1 - four accumulators as before, but every query fragment read from LDS and every widened gallery
// fragment now feeds TWO MFMAs.  With one row block x four query blocks per wave (above) the four waves read 4 x 4 KB of query fragments per
// k-step, exactly the CU's LDS bandwidth, and the int8 scan ran at 2.95 TB/s with the matrix pipe a third busy (profiles/r04/r04j); here the LDS
// traffic is halved (and the widening doubled: 512 VALU operations per tile and wave, still under the MFMA time).
__global__ __launch_bounds__(256) void match_coarse_i8x4_kernel(const uint8_t *__restrict__ G8, const float *__restrict__ gscale, int N, int F,
                                                                float *__restrict__ tilemax, int num_tiles, const float *__restrict__ Q32,
                                                                float *__restrict__ wgmax, int *__restrict__ ctl, unsigned long long *__restrict__ qkey) {
    constexpr int D = 512, KS = D / 16, KP = D / 32, QP = D + 8;
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    half_t *Qs = reinterpret_cast<half_t *>(smem2);  // [128][QP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int rb = wave >> 1, qh = wave & 1;
    const int q0 = blockIdx.y * 128;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int i = tid; i < CTL_WORDS; i += 256) ctl[i] = 0;
        for (int i = tid; i < F; i += 256) qkey[i] = 0ull;
    }
    for (int i = tid; i < 128 * (D / 8); i += 256) {
        const int q = i / (D / 8), c = i - q * (D / 8);
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q0 + q < F) {
            const floatx4 a = *reinterpret_cast<const floatx4 *>(Q32 + (long)(q0 + q) * D + c * 8), b = *reinterpret_cast<const floatx4 *>(Q32 + (long)(q0 + q) * D + c * 8 + 4);
            v = half8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
        }
        *reinterpret_cast<half8 *>(Qs + q * QP + c * 8) = v;
    }
    float rmax[2] = {-INFINITY, -INFINITY};
    bool rnan = false;
    int tile = blockIdx.x;
    if (tile >= num_tiles) {
        for (int i = tid; i < 128; i += 256)
            if (q0 + i < F) wgmax[(long)blockIdx.x * F + q0 + i] = -INFINITY;
        return;
    }
    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    auto frag_ptr = [&](int t, int b) { return reinterpret_cast<const uint4_t *>(G8 + (((long)t * 4 + 2 * rb + b) * KP) * 1024) + lane; };
    auto scale_ptr = [&](int t, int b) { return gscale + (long)t * 128 + (2 * rb + b) * 32 + 4 * hi; };
    uint4_t areg[2][KP];
    floatx4 sreg[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const uint4_t *gp = frag_ptr(tile, b);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) areg[b][kp] = __builtin_nontemporal_load(gp + kp * 64);
        const float *sp = scale_ptr(tile, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) sreg[b][j] = *reinterpret_cast<const floatx4 *>(sp + 8 * j);
    }
    __syncthreads();
    const half_t *qb = Qs + (qh * 64 + r) * QP + 8 * hi;
    for (; tile < num_tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const int nt = next < num_tiles ? next : tile;
        const uint4_t *gn0 = frag_ptr(nt, 0), *gn1 = frag_ptr(nt, 1);
        floatx16 acc[2][2];  // [row block][query block]
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][n][e] = 0.f;
        half8 bq[2][2], af[2][2];  // [ring slot][query block | row block]
#pragma unroll
        for (int n = 0; n < 2; ++n) bq[0][n] = *reinterpret_cast<const half8 *>(qb + n * 32 * QP);
        af[0][0] = i8x8_to_half8(areg[0][0][0], areg[0][0][1]);
        af[0][1] = i8x8_to_half8(areg[1][0][0], areg[1][0][1]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {  // one k-step ahead, order pinned (see match_coarse_i8_kernel)
#pragma unroll
                for (int n = 0; n < 2; ++n) bq[(ks + 1) & 1][n] = *reinterpret_cast<const half8 *>(qb + n * 32 * QP + (ks + 1) * 16);
                const uint4_t raw0 = areg[0][(ks + 1) >> 1], raw1 = areg[1][(ks + 1) >> 1];
                af[(ks + 1) & 1][0] = ((ks + 1) & 1) ? i8x8_to_half8(raw0[2], raw0[3]) : i8x8_to_half8(raw0[0], raw0[1]);
                af[(ks + 1) & 1][1] = ((ks + 1) & 1) ? i8x8_to_half8(raw1[2], raw1[3]) : i8x8_to_half8(raw1[0], raw1[1]);
                if ((ks + 1) & 1) {
                    areg[0][(ks + 1) >> 1] = __builtin_nontemporal_load(gn0 + ((ks + 1) >> 1) * 64);
                    areg[1][(ks + 1) >> 1] = __builtin_nontemporal_load(gn1 + ((ks + 1) >> 1) * 64);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][0], bq[ks & 1][n], acc[0][n], 0, 0, 0);
                acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][1], bq[ks & 1][n], acc[1][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int gbase = tile * 128 + (2 * rb + b) * 32;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float m = -INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int g = gbase + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (g < N) m = fmaxf(m, acc[b][n][e] * sreg[b][e >> 2][e & 3]);
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                const int q = q0 + (2 * qh + n) * 32 + r;
                if (hi == 0 && q < F) tilemax[((long)q * num_tiles + tile) * 4 + 2 * rb + b] = m;
                rnan = rnan || (m != m);
                rmax[n] = fmaxf(rmax[n], m);
            }
            const float *sp = scale_ptr(nt, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) sreg[b][j] = *reinterpret_cast<const floatx4 *>(sp + 8 * j);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem2);  // [row half][128 queries]
    if (hi == 0) {
#pragma unroll
        for (int n = 0; n < 2; ++n) red[rb * 128 + (2 * qh + n) * 32 + r] = rnan ? NAN : rmax[n];
    }
    __syncthreads();
    if (tid < 128 && q0 + tid < F) {
        const float a = red[tid], b = red[128 + tid];
        wgmax[(long)blockIdx.x * F + q0 + tid] = (a != a || b != b) ? NAN : fmaxf(a, b);
    }
}

// Tile selection in two fully parallel steps (it used to be one workgroup per query walking its 31 252 coarse entries twice: 128
// workgroups, 80 us).  Step 1: grid (segments, queries) - maximum of a segment of the query's coarse entries.  Step 2: same grid -
// the query's best coarse maximum from the segment maxima, ||q||, then every tile of the segment within 2*delta of the best goes on the
// list (once, over all queries).
constexpr int SEL_SEG = 16;  // segments per query

__global__ __launch_bounds__(256) void match_segmax_kernel(const float *__restrict__ tilemax, int n_ent, float *__restrict__ segmax) {
    __shared__ float sm[4];
    const int q = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
    const int e0 = (int)((long)n_ent * seg / SEL_SEG), e1 = (int)((long)n_ent * (seg + 1) / SEL_SEG);
    const float *row = tilemax + (long)q * n_ent;
    float m = -INFINITY;
    for (int t = e0 + tid; t < e1; t += 256) m = fmaxf(m, row[t]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) sm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) segmax[q * SEL_SEG + seg] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// kth != nullptr (top-k): the threshold hangs on the query's k-th largest coarse entry instead of its largest.  With |S~ - S| <= delta:
// the k-th largest coarse entry m_k is the maximum of a 32-row block, k blocks have one >= m_k, hence k distinct rows have S >= m_k -
// delta, hence the exact k-th best similarity s_k >= m_k - delta, and every row with S >= s_k has S~ >= m_k - 2*delta: it lives in a
// listed tile.  (k = 1 is the rule above.)
__global__ __launch_bounds__(256) void match_select_kernel(const float *__restrict__ tilemax, int num_tiles, int sub, const float *__restrict__ segmax,
                                                           const float *__restrict__ Q, int D, float gmax_norm, int *__restrict__ tile_flags,
                                                           int *__restrict__ tile_list, int *__restrict__ count, const float *__restrict__ kth = nullptr) {
    // `sub` coarse entries per 128-row tile (match_coarse_kernel writes one per wave: 4)
    __shared__ float sm[4];
    const int q = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
    const int n_ent = num_tiles * sub;
    const float *row = tilemax + (long)q * n_ent;
    float m = -INFINITY, n2 = 0.f;
    if (kth) m = kth[q];
    else
        for (int s = 0; s < SEL_SEG; ++s) m = fmaxf(m, segmax[q * SEL_SEG + s]);
    for (int k = tid; k < D; k += 256) n2 += Q[(long)q * D + k] * Q[(long)q * D + k];
    for (int off = 32; off > 0; off >>= 1) n2 += __shfl_xor(n2, off);
    if ((tid & 63) == 0) sm[tid >> 6] = n2;
    __syncthreads();
    const float qn = sqrtf(sm[0] + sm[1] + sm[2] + sm[3]);
    const float delta = 1.2e-3f * qn * gmax_norm;
    // fp16 overflow / non-finite inputs: no valid bound -> take every tile (degenerates to the exact full scan)
    // (top-k with fewer than k coarse entries: kth = -inf -> every tile)
    const float thr = (qn < 6.0e4f && gmax_norm < 6.0e4f && m == m && m > -INFINITY && m < INFINITY) ? m - 2.f * delta : -INFINITY;
    const int e0 = (int)((long)n_ent * seg / SEL_SEG), e1 = (int)((long)n_ent * (seg + 1) / SEL_SEG);
    for (int t = e0 + tid; t < e1; t += 256)
        if (!(row[t] < thr) && atomicExch(&tile_flags[t / sub], 1) == 0) tile_list[atomicAdd(count, 1)] = t / sub;
}

// ---------------------------------------------------------------- top-k support (BASELINE configs[4]: "RCCL top-k all-gather")
constexpr int TOPK_MAX = 16;

// k-th largest coarse entry of every query (grid = queries).  Every thread keeps the TOPK_MAX largest of its strided entries in a
// sorted register list; the block then pops the overall maximum k times (each pop removes one entry: the owner advances its list).
__global__ __launch_bounds__(256) void match_kth_kernel(const float *__restrict__ tilemax, int n_ent, int k, float *__restrict__ kth) {
    __shared__ float sv[4];
    __shared__ int so[4];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *row = tilemax + (long)q * n_ent;
    float top[TOPK_MAX];
#pragma unroll
    for (int i = 0; i < TOPK_MAX; ++i) top[i] = -INFINITY;
    for (int t = tid; t < n_ent; t += 256) {
        float v = row[t];
        if (!(v > top[TOPK_MAX - 1])) continue;  // (NaN never enters)
#pragma unroll
        for (int i = 0; i < TOPK_MAX; ++i) {    // insert, keeping the list sorted (descending)
            const float hi = fmaxf(top[i], v), lo = fminf(top[i], v);
            top[i] = hi;
            v = lo;
        }
    }
    int pos = 0;
    float result = -INFINITY;
    for (int j = 0; j < k; ++j) {
        float head = -INFINITY;
#pragma unroll
        for (int i = 0; i < TOPK_MAX; ++i)
            if (i == pos) head = top[i];
        float v = head;
        int o = tid;
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int oo = __shfl_xor(o, off);
            if (ov > v || (ov == v && oo < o)) {
                v = ov;
                o = oo;
            }
        }
        if (lane == 0) {
            sv[wave] = v;
            so[wave] = o;
        }
        __syncthreads();
        v = sv[0];
        o = so[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > v || (sv[w] == v && so[w] < o)) {
                v = sv[w];
                o = so[w];
            }
        __syncthreads();
        result = v;
        if (o == tid && pos < TOPK_MAX) ++pos;
    }
    if (tid == 0) kth[q] = result;
}

// fp16 <-> fp32 rows (the embedding exchange of configs[4] travels as fp16: half the xGMI bytes; the widening is exact)
__global__ __launch_bounds__(256) void half_to_float_kernel(const half_t *__restrict__ in, float *__restrict__ out, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const half8 h = *reinterpret_cast<const half8 *>(in + i * 8);
    *reinterpret_cast<floatx4 *>(out + i * 8) = floatx4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    *reinterpret_cast<floatx4 *>(out + i * 8 + 4) = floatx4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
}

// k-way merge of per-shard top-k lists (each sorted: higher similarity first, lower GLOBAL index first on ties; idx < 0 = empty
// slot): idx_all / sim_all [shards][n][k] -> [n][k].  One thread per query; shards * k is a few dozen entries.
__global__ __launch_bounds__(64) void merge_topk_kernel(const int32_t *__restrict__ idx_all, const float *__restrict__ sim_all, int shards, int n, int k,
                                                        int32_t *__restrict__ idx_out, float *__restrict__ sim_out) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= n) return;
    // the heads are found by rescanning: entry (s, j) is a candidate iff it comes strictly after the last winner in the result order
    float lv = INFINITY;
    int li = -1;
    for (int o = 0; o < k; ++o) {
        float bv = -INFINITY;
        int bi = INT_MAX;
        for (int s = 0; s < shards; ++s)
            for (int j = 0; j < k; ++j) {
                const long e = ((long)s * n + q) * k + j;
                const int i = idx_all[e];
                const float v = sim_all[e];
                if (i < 0) continue;
                if (!((v < lv) || (v == lv && i > li))) continue;
                if (better(v, i, bv, bi)) {
                    bv = v;
                    bi = i;
                }
            }
        idx_out[(long)q * k + o] = bi == INT_MAX ? -1 : bi;
        sim_out[(long)q * k + o] = bi == INT_MAX ? -INFINITY : bv;
        if (bi == INT_MAX) {  // exhausted: the remaining slots are empty too
            for (int o2 = o + 1; o2 < k; ++o2) {
                idx_out[(long)q * k + o2] = -1;
                sim_out[(long)q * k + o2] = -INFINITY;
            }
            return;
        }
        lv = bv;
        li = bi;
    }
}

// ---------------------------------------------------------------- fast screened top-1 (round 3)
// The tile list + MFMA re-rank above recomputes ALL queries against every listed tile (128 x 128 dot products per tile where typically
// one query asked for it) behind a per-tile barrier chain: 59 us for a few hundred tiles.  Here the selection emits (query, tile)
// PAIRS and a scalar kernel recomputes exactly those 128 dot products per pair - one thread per gallery row walking k in the order the
// exact kernel's v_mfma_f32_32x32x2_f32 chain consumes it, so the similarities are the same bits as the full scan's (checked by
// tools/ubench/mfma_f32_order.hip and by the bit-identity tests).  Per pair the best row (first-maximum rule) goes into the query's
// 64-bit slot with one atomicMax: high word = the similarity mapped monotonically to an unsigned integer, low word = ~row, so a higher
// similarity wins and among equal similarities the LOWER row.  More pairs than the list holds (non-finite inputs, degenerate
// galleries): the overflow flag routes the call through the unscreened exact scan instead (gate argument below).
struct MatchPair {
    int q, tile;  // tile: bits 0 - 27 the 128-row tile, bits 28 - 31 which of its four 32-row blocks are inside the band
};
constexpr int PAIR_TILE_MASK = (1 << 28) - 1;
// ctl words: [1] overflow flag, [CTL_SEG0 + s * CTL_STRIDE] number of pairs in sub-list s.  The pair list is SEL_SUB sub-lists of pair_cap / SEL_SUB
// entries, one per (tile segment = blockIdx.x of the selection, query & 3): 1 400 increments of ONE counter cost 35 us of atomics (queries that match
// nothing, int8 band).  Round 4 shared them out over sixteen counters in ONE cache line - the selection still took 27 us with ~ 2 000 pairs against
// 7 us with 128 (returning atomics on one line queue in one L2 channel whatever the address); round 6: 64 counters, each in its own 128-byte line.
constexpr int CTL_OVERFLOW = 1, CTL_SEG0 = 32, CTL_STRIDE = 32, SEL_QSUB = 4, SEL_SUB = SEL_SEG * SEL_QSUB;
static_assert(CTL_SEG0 + SEL_SUB * CTL_STRIDE <= CTL_WORDS, "control words");
constexpr int FB_BLOCKS = 128;  // workgroups of the gated fallback scan (it returns at once in the normal case: keep the empty launch small)

__device__ __forceinline__ unsigned mono_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unmono_bits(unsigned m) { return __uint_as_float((m & 0x80000000u) ? (m & 0x7fffffffu) : ~m); }

// grid (SEL_SEG, F): the query's best coarse entry from the per-workgroup maxima, ||q||, then every tile of the segment with a coarse
// entry within 2*delta of it becomes a pair
__global__ __launch_bounds__(256) void match_select_pairs_kernel(const float *__restrict__ tilemax, int num_tiles, const float *__restrict__ wgmax, int n_wg,
                                                                 int F, const float *__restrict__ Q, int D, float gmax_norm, MatchPair *__restrict__ pairs,
                                                                 int pair_cap, int *__restrict__ ctl, const float *__restrict__ kth = nullptr,
                                                                 float c_round = 1.2e-3f, float gerr = 0.f) {
    // c_round / gerr: |S~ - S| <= ||q|| * (c_round * max||g|| + 1.02 * gerr).  fp16 shadow: two fp16 roundings (1.2e-3), no storage error;
    // int8 shadow: one rounding (the query's: 0.7e-3) + the measured quantisation error norm of the rows
    // kth != nullptr (top-k): the threshold hangs on the query's k-th largest coarse entry (match_kth_kernel; see match_select_kernel)
    __shared__ float sm[4], sn[4];
    __shared__ int snan[4];
    const int q = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
    float m = -INFINITY, n2 = 0.f;
    int isnan_ = 0;
    for (int w = tid; w < n_wg; w += 256) {
        const float v = wgmax[(long)w * F + q];
        isnan_ |= v != v;
        m = fmaxf(m, v);
    }
    for (int k = tid; k < D; k += 256) n2 += Q[(long)q * D + k] * Q[(long)q * D + k];
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off));
        n2 += __shfl_xor(n2, off);
        isnan_ |= __shfl_xor(isnan_, off);
    }
    if ((tid & 63) == 0) {
        sm[tid >> 6] = m;
        sn[tid >> 6] = n2;
        snan[tid >> 6] = isnan_;
    }
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (kth) m = kth[q];
    const bool anynan = snan[0] | snan[1] | snan[2] | snan[3];
    const float qn = sqrtf(sn[0] + sn[1] + sn[2] + sn[3]);
    const float delta = qn * (c_round * gmax_norm + 1.02f * gerr);
    // same rule as match_select_kernel: fp16 overflow / non-finite inputs -> no valid bound -> every tile (which overflows the pair list
    // and sends the call through the exact full scan)
    const float thr = (!anynan && qn < 6.0e4f && gmax_norm < 6.0e4f && gerr < 6.0e4f && m == m && m > -INFINITY && m < INFINITY) ? m - 2.f * delta : -INFINITY;
    const int t0 = (int)((long)num_tiles * seg / SEL_SEG), t1 = (int)((long)num_tiles * (seg + 1) / SEL_SEG);
    const floatx4 *row = reinterpret_cast<const floatx4 *>(tilemax + (long)q * num_tiles * 4);
    const int lane = tid & 63;
    for (int tb = t0; tb < t1; tb += 256) {  // (uniform trip count: the ballot below needs whole waves)
        const int t = tb + tid;
        unsigned mask = 0u;
        if (t < t1) {
            const floatx4 e = row[t];
            // (the re-rank reads only the flagged 32-row blocks: with the wider band of the int8 shadow a query that matches nothing lists
            //  a dozen tiles, and whole tiles were 256 KB of fp32 rows per pair - 125 us of re-rank at 128 such queries)
            mask = (!(e[0] < thr) ? 1u : 0u) | (!(e[1] < thr) ? 2u : 0u) | (!(e[2] < thr) ? 4u : 0u) | (!(e[3] < thr) ? 8u : 0u);
        }
        // one atomic per wave (1 400 single increments of one counter cost 35 us)
        const unsigned long long bal = __ballot(mask != 0u);
        if (bal) {
            const int first = __ffsll((long long)bal) - 1;
            const int sub_cap = pair_cap / SEL_SUB, sub = seg + SEL_SEG * (q & (SEL_QSUB - 1));
            int base = 0;
            if (lane == first) base = atomicAdd(&ctl[CTL_SEG0 + sub * CTL_STRIDE], __popcll(bal));
            base = __shfl(base, first);
            if (mask) {
                const int i = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (i < sub_cap) pairs[sub * sub_cap + i] = MatchPair{q, (int)((unsigned)t | (mask << 28))};
                else ctl[CTL_OVERFLOW] = 1;
            }
        }
    }
}

// one workgroup (128 threads = the tile's 128 rows) per pair, pairs dealt round-robin over the grid
template <typename GT>
__global__ __launch_bounds__(128) void match_rerank_pairs_kernel(const GT *__restrict__ G, int N, int D, const float *__restrict__ E, const MatchPair *__restrict__ pairs,
                                                                 int pair_cap, const int *__restrict__ ctl, unsigned long long *__restrict__ qkey,
                                                                 const float *__restrict__ prev_sim = nullptr, const int32_t *__restrict__ prev_idx = nullptr,
                                                                 int prev_stride = 1, int row_offset = 0) {
    // prev_sim != nullptr (top-k pass j > 0): only rows strictly AFTER the query's previous winner (its similarity, its GLOBAL index) in the
    // result order take part - the EXCL rule of match_kernel
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    float *qs = reinterpret_cast<float *>(smem3);  // [D]
    __shared__ float rv[2];
    __shared__ int ri[2];
    const int tid = threadIdx.x;
    if (ctl[CTL_OVERFLOW]) return;  // the exact full scan answers this call
    const int sub_cap = pair_cap / SEL_SUB;
    int maxc = 0;
    for (int sgi = 0; sgi < SEL_SUB; ++sgi) maxc = max(maxc, min(ctl[CTL_SEG0 + sgi * CTL_STRIDE], sub_cap));
    for (int p = blockIdx.x; p < maxc * SEL_SUB; p += gridDim.x) {  // virtual index over the sub-lists, interleaved
        const int sgi = p % SEL_SUB, pi_ = p / SEL_SUB;
        if (pi_ >= ctl[CTL_SEG0 + sgi * CTL_STRIDE]) continue;  // (block-uniform)
        const MatchPair pr = pairs[sgi * sub_cap + pi_];
        __syncthreads();  // the previous pair's readers of qs / rv are done
        for (int k = tid * 4; k < D; k += 512) *reinterpret_cast<floatx4 *>(qs + k) = *reinterpret_cast<const floatx4 *>(E + (long)pr.q * D + k);
        __syncthreads();
        const long g = (long)(pr.tile & PAIR_TILE_MASK) * 128 + tid;
        const bool live = g < N && ((((unsigned)pr.tile >> 28) >> (tid >> 5)) & 1u);  // this row's 32-row block is inside the band
        float acc = 0.f;
        if (live) {
            // k order of the exact kernel: per 32-wide step, for ks in 0..3, for s in 0..3: k = 8 ks + s (lanes 0-31 of the MFMA), then
            // k = 8 ks + 4 + s (lanes 32-63) - one fused multiply-add each
#pragma unroll 8
            for (int k0 = 0; k0 < D; k0 += 8) {  // (unrolled: 16 row loads in flight per thread; the fma chain itself stays sequential)
                const floatx4 a = load_g4(G, g, D, k0), b = load_g4(G, g, D, k0 + 4);
                const floatx4 x = *reinterpret_cast<const floatx4 *>(qs + k0), y = *reinterpret_cast<const floatx4 *>(qs + k0 + 4);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    acc = __builtin_fmaf(a[s2], x[s2], acc);
                    acc = __builtin_fmaf(b[s2], y[s2], acc);
                }
            }
        }
        float v = live ? acc : -INFINITY;
        int i = live ? (int)g : INT_MAX;
        if (prev_sim && i != INT_MAX) {
            const float ps = prev_sim[(long)pr.q * prev_stride];
            const int pi = prev_idx[(long)pr.q * prev_stride];
            if (!((v < ps) || (v == ps && i + row_offset > pi))) {
                v = -INFINITY;
                i = INT_MAX;
            }
        }
        if (v != v) {  // NaN never wins (std::max_element with operator<: a NaN is never greater)
            v = -INFINITY;
            i = INT_MAX;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int oi = __shfl_xor(i, off);
            if (better(ov, oi, v, i)) {
                v = ov;
                i = oi;
            }
        }
        if ((tid & 63) == 0) {
            rv[tid >> 6] = v;
            ri[tid >> 6] = i;
        }
        __syncthreads();
        if (tid == 0) {
            if (better(rv[1], ri[1], v, i)) {
                v = rv[1];
                i = ri[1];
            }
            if (i != INT_MAX) atomicMax(&qkey[pr.q], ((unsigned long long)mono_bits(v) << 32) | (unsigned long long)(~(unsigned)i));
        }
    }
}

// Last kernel of a pass.  Normal case: the packed winners of the pair re-rank.  Overflow case: the gated exact scan ran in front of this
// kernel (fb_blocks workgroups); its per-workgroup partials are reduced here, one thread per query, with the same first-maximum rule.
__global__ __launch_bounds__(256) void match_unpack_kernel(unsigned long long *__restrict__ qkey, int F, int row_offset, const int *__restrict__ ctl,
                                                           int32_t *__restrict__ idx_out, float *__restrict__ sim_out, int out_stride,
                                                           const MatchPartial *__restrict__ fb_partial, int fb_blocks) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= F) return;
    if (ctl[CTL_OVERFLOW]) {
        float v = -INFINITY;
        int i = INT_MAX;
        for (int b = 0; b < fb_blocks; ++b) {
            const MatchPartial p = fb_partial[(long)b * F + q];
            if (p.idx >= 0 && better(p.sim, p.idx, v, i)) {
                v = p.sim;
                i = p.idx;
            }
        }
        qkey[q] = 0ull;
        idx_out[(long)q * out_stride] = i == INT_MAX ? -1 : i;  // (the scan's partial indices already carry the row offset)
        sim_out[(long)q * out_stride] = v;
        return;
    }
    const unsigned long long k = qkey[q];
    qkey[q] = 0ull;  // ready for the next top-k pass of this call
    idx_out[(long)q * out_stride] = k ? (int)(~(unsigned)(k & 0xffffffffull)) + row_offset : -1;
    sim_out[(long)q * out_stride] = k ? unmono_bits((unsigned)(k >> 32)) : -INFINITY;
}

template <int NQ, bool FULL, typename GT = float, bool EXCL = false>
void launch_t(const GT *G, int N, int D, const float *E, int F, MatchPartial *partial, float *out_full, int blocks, int row_offset,
              hipStream_t s, const int *tile_list = nullptr, const int *d_num_tiles = nullptr, const float *prev_sim = nullptr,
              const int32_t *prev_idx = nullptr, int prev_stride = 1, const int *gate = nullptr) {
    const int tiles = (N + BM - 1) / BM;
    const size_t lds = (size_t)2 * (BM + NQ * 32) * BK * sizeof(float);
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_kernel<NQ, FULL, GT, EXCL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    dim3 grid(blocks, (F + NQ * 32 - 1) / (NQ * 32));
    hipLaunchKernelGGL((match_kernel<NQ, FULL, GT, EXCL>), grid, dim3(256), lds, s, G, N, D, E, F, partial, out_full, tiles, row_offset, tile_list, d_num_tiles,
                       prev_sim, prev_idx, prev_stride, gate);
}

}  // namespace

int match_top1_blocks(int N, int F) {
    (void)F;
    const int tiles = (N + BM - 1) / BM;
    return tiles < 512 ? (tiles > 0 ? tiles : 1) : 512;
}

void launch_match_top1(const float *gallery, int N, int D, const float *queries, int F, MatchPartial *partial, int partial_blocks,
                       int32_t *idx_out, float *sim_out, int row_offset, hipStream_t s) {
    // NOTE: with grid.y > 1 every query tile writes its own columns of `partial` ([blocks][F]).
    if (F <= 32)
        launch_t<1, false>(gallery, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s);
    else if (F <= 64)
        launch_t<2, false>(gallery, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s);
    else
        launch_t<4, false>(gallery, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s);
    hipLaunchKernelGGL(match_reduce_kernel, dim3(F), dim3(64), 0, s, partial, partial_blocks, F, idx_out, sim_out, 1, (const int *)nullptr);
}

void launch_match_top1_h(const half_t *g16, int N, int D, const float *queries, int F, MatchPartial *partial, int partial_blocks,
                         int32_t *idx_out, float *sim_out, int row_offset, hipStream_t s) {
    if (F <= 32)
        launch_t<1, false, half_t>(g16, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s);
    else if (F <= 64)
        launch_t<2, false, half_t>(g16, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s);
    else
        launch_t<4, false, half_t>(g16, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s);
    hipLaunchKernelGGL(match_reduce_kernel, dim3(F), dim3(64), 0, s, partial, partial_blocks, F, idx_out, sim_out, 1, (const int *)nullptr);
}

void launch_match_full_h(const half_t *g16, int N, int D, const float *queries, int F, float *out, hipStream_t s) {
    const int blocks = match_top1_blocks(N, F);
    if (F <= 32)
        launch_t<1, true, half_t>(g16, N, D, queries, F, nullptr, out, blocks, 0, s);
    else
        launch_t<4, true, half_t>(g16, N, D, queries, F, nullptr, out, blocks, 0, s);
}

void launch_match_full(const float *gallery, int N, int D, const float *queries, int F, float *out, hipStream_t s) {
    const int blocks = match_top1_blocks(N, F);
    if (F <= 32)
        launch_t<1, true>(gallery, N, D, queries, F, nullptr, out, blocks, 0, s);
    else
        launch_t<4, true>(gallery, N, D, queries, F, nullptr, out, blocks, 0, s);
}

// ---------------------------------------------------------------- screened top-1 (host side)
bool match_screen_supported(int D) { return D == 64 || D == 128 || D == 256 || D == 512; }  // coarse kernel instantiations (queries must fit LDS)

size_t gallery16_elems(int N, int D) { return (size_t)((N + 127) / 128) * 128 * D; }  // whole 128-row tiles

// fp32 rows [n_rows][D] whose first row is global row row0 (a multiple of 128) -> their place in the fragment-ordered fp16 gallery
void launch_rows_to_half(const float *in, long row0, long n_rows, int D, half_t *g16, hipStream_t s) {
    if (n_rows <= 0) return;
    const long tiles = (n_rows + 127) / 128;
    const long pieces = tiles * 4 * (D / 16) * 64;
    hipLaunchKernelGGL(gallery_to_half_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, in, row0, n_rows, D, g16);
}

void launch_gallery_shadow(const float *gallery, int N, int D, half_t *g16, int *max_norm2_bits, hipStream_t s) {
    (void)hipMemsetAsync(max_norm2_bits, 0, sizeof(int), s);
    (void)hipMemsetAsync(g16, 0, gallery16_elems(N, D) * sizeof(half_t), s);  // (pad rows of the last tile stay zero)
    launch_rows_to_half(gallery, 0, N, D, g16, s);
    hipLaunchKernelGGL(row_norm_max_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, s, gallery, N, D, max_norm2_bits);
}

// fp16-STORED gallery (no fp32 copy on the device): only the largest row norm is needed, the stored rows are their own shadow
void launch_gallery_norm16(const half_t *g16, int N, int D, int *max_norm2_bits, hipStream_t s) {
    (void)hipMemsetAsync(max_norm2_bits, 0, sizeof(int), s);
    hipLaunchKernelGGL(row_norm_max_kernel<half_t>, dim3((N + 3) / 4), dim3(256), 0, s, g16, N, D, max_norm2_bits);
}

size_t gallery8_bytes(int N, int D) { return (size_t)((N + 127) / 128) * 128 * D; }
void launch_gallery_shadow8(const float *gallery, int N, int D, uint8_t *g8, float *scale, int *max_err2_bits, int *max_norm2_bits, hipStream_t s) {
    const size_t rows = (size_t)((N + 127) / 128) * 128;
    (void)hipMemsetAsync(max_err2_bits, 0, sizeof(int), s);
    (void)hipMemsetAsync(max_norm2_bits, 0, sizeof(int), s);
    (void)hipMemsetAsync(g8, 0x80, gallery8_bytes(N, D), s);  // pad rows of the last tile: value 0
    (void)hipMemsetAsync(scale, 0, rows * sizeof(float), s);
    if (D != 512 || N <= 0) return;
    hipLaunchKernelGGL(gallery_to_i8_kernel, dim3((N + 3) / 4), dim3(256), 0, s, gallery, N, g8, scale, max_err2_bits, max_norm2_bits);
}

constexpr int COARSE_WG = 256;  // persistent workgroups of the coarse scan (one per CU)
template <int NQB>
static void launch_coarse_i8_t(const ScreenScratch &w, int N, const float *q32, int F, int tiles, hipStream_t s) {
    const size_t lds = (size_t)32 * NQB * (512 + 8) * sizeof(half_t);
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_coarse_i8_kernel<NQB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // wgmax is [COARSE_WG][F]: the grid never exceeds COARSE_WG workgroups per query block (smaller query blocks leave room for several per CU)
    dim3 g(tiles < COARSE_WG ? tiles : COARSE_WG, (F + 32 * NQB - 1) / (32 * NQB));
    hipLaunchKernelGGL(match_coarse_i8_kernel<NQB>, g, dim3(256), lds, s, w.g8, w.g8_scale, N, F, w.tilemax, tiles, q32, w.wgmax, w.ctl, w.qkey);
}
static void launch_coarse_i8(const ScreenScratch &w, int N, const float *q32, int F, int tiles, hipStream_t s) {
    if (F <= 32) return launch_coarse_i8_t<1>(w, N, q32, F, tiles, s);
    if (F <= 64) return launch_coarse_i8_t<2>(w, N, q32, F, tiles, s);
    // (the 2 x 2 wave tiling below halves the LDS reads and doubles the widening: 179 us against 163 for the 1 x 4 tiling - tuning switch only)
    static const bool x4 = frt_tuning_env("FRT_MATCH_I8X4") && frt_tuning_env("FRT_MATCH_I8X4")[0] == '1';
    if (!x4) return launch_coarse_i8_t<4>(w, N, q32, F, tiles, s);
    const size_t lds = (size_t)128 * (512 + 8) * sizeof(half_t);
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_coarse_i8x4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 g(tiles < COARSE_WG ? tiles : COARSE_WG, (F + 127) / 128);
    hipLaunchKernelGGL(match_coarse_i8x4_kernel, g, dim3(256), lds, s, w.g8, w.g8_scale, N, F, w.tilemax, tiles, q32, w.wgmax, w.ctl, w.qkey);
}

template <int D>
static void launch_coarse_t(const half_t *g16, int N, const half_t *q16, int F, float *tilemax, int tiles, hipStream_t s, const float *q32 = nullptr,
                            float *wgmax = nullptr, int *ctl = nullptr, unsigned long long *qkey = nullptr) {
    const size_t lds = (size_t)128 * (D + 8) * sizeof(half_t);
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_coarse_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 g(tiles < COARSE_WG ? tiles : COARSE_WG, (F + 127) / 128);
    hipLaunchKernelGGL((match_coarse_kernel<D>), g, dim3(256), lds, s, g16, N, q16, F, tilemax, tiles, q32, wgmax, ctl, qkey);
}

// coarse pass + tile selection: leaves the candidate tile list (w.tile_list, length *w.count on the device).  k > 1: the threshold hangs
// on every query's k-th largest coarse entry (kth_scratch [F] floats).
static void screen_tiles(const half_t *g16, int N, int D, const float *queries, int F, float gmax_norm, const ScreenScratch &w, int k,
                         float *kth_scratch, hipStream_t s) {
    const int tiles = (N + BM - 1) / BM;
    const long q8 = (long)F * D / 8;
    const long nzero = (long)tiles + 1;  // tile flags + the candidate count behind them (ScreenScratch: count == tile_flags + tiles)
    const long n_thr = q8 > nzero ? q8 : nzero;
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, s, queries, w.q16, q8, w.tile_flags, nzero);
    switch (D) {
        case 64: launch_coarse_t<64>(g16, N, w.q16, F, w.tilemax, tiles, s); break;
        case 128: launch_coarse_t<128>(g16, N, w.q16, F, w.tilemax, tiles, s); break;
        case 256: launch_coarse_t<256>(g16, N, w.q16, F, w.tilemax, tiles, s); break;
        default: launch_coarse_t<512>(g16, N, w.q16, F, w.tilemax, tiles, s); break;  // match_screen_supported() gates the callers
    }
    if (k > 1) {
        hipLaunchKernelGGL(match_kth_kernel, dim3(F), dim3(256), 0, s, w.tilemax, tiles * 4, k, kth_scratch);
        hipLaunchKernelGGL(match_select_kernel, dim3(SEL_SEG, F), dim3(256), 0, s, w.tilemax, tiles, 4, w.segmax, queries, D, gmax_norm, w.tile_flags,
                           w.tile_list, w.count, kth_scratch);
    } else {
        hipLaunchKernelGGL(match_segmax_kernel, dim3(SEL_SEG, F), dim3(256), 0, s, w.tilemax, tiles * 4, w.segmax);
        hipLaunchKernelGGL(match_select_kernel, dim3(SEL_SEG, F), dim3(256), 0, s, w.tilemax, tiles, 4, w.segmax, queries, D, gmax_norm, w.tile_flags,
                           w.tile_list, w.count, (const float *)nullptr);
    }
}

// Exact top-k: idx_out / sim_out [F][k], row j of a query = the j-th entry of its exact ranking (higher similarity first, lower global
// index first among equal similarities; -1 / -inf when the gallery has fewer than k rows).  Pass j is the top-1 search restricted to
// the rows that come after winner j-1 - the same accumulators, the same first-maximum rule; with a screened gallery the coarse scan
// runs ONCE (threshold on the k-th largest coarse entry) and only the short exact re-rank is repeated.
void launch_match_topk(const float *gallery, const half_t *g16, int N, int D, const float *queries, int F, int k, bool screen, float gmax_norm,
                       const ScreenScratch &w, float *kth_scratch, MatchPartial *partial, int partial_blocks, int32_t *idx_out, float *sim_out,
                       int row_offset, hipStream_t s) {
    const int tiles_ = (N + BM - 1) / BM;
    if (screen && w.pairs && w.wgmax && tiles_ >= COARSE_WG) {
        // fast path (round 3): ONE coarse scan, (query, tile) pairs within the rounding band of the k-th largest coarse entry, then k
        // passes of the scalar pair re-rank (a few microseconds each), pass j restricted to the rows behind winner j - 1.  With the
        // all-gathered queries of a node (configs[4]: 2 048 of them per rank) the tile-list path would re-rank every query against every
        // listed tile, k times.
        const bool i8 = w.g8 && D == 512;
        if (i8) launch_coarse_i8(w, N, queries, F, tiles_, s);
        else switch (D) {
            case 64: launch_coarse_t<64>(g16, N, nullptr, F, w.tilemax, tiles_, s, queries, w.wgmax, w.ctl, w.qkey); break;
            case 128: launch_coarse_t<128>(g16, N, nullptr, F, w.tilemax, tiles_, s, queries, w.wgmax, w.ctl, w.qkey); break;
            case 256: launch_coarse_t<256>(g16, N, nullptr, F, w.tilemax, tiles_, s, queries, w.wgmax, w.ctl, w.qkey); break;
            default: launch_coarse_t<512>(g16, N, nullptr, F, w.tilemax, tiles_, s, queries, w.wgmax, w.ctl, w.qkey); break;
        }
        hipLaunchKernelGGL(match_kth_kernel, dim3(F), dim3(256), 0, s, w.tilemax, tiles_ * 4, k, kth_scratch);
        hipLaunchKernelGGL(match_select_pairs_kernel, dim3(SEL_SEG, F), dim3(256), 0, s, w.tilemax, tiles_, w.wgmax, COARSE_WG, F, queries, D, gmax_norm,
                           reinterpret_cast<MatchPair *>(w.pairs), w.pair_cap, w.ctl, (const float *)kth_scratch, i8 ? 0.7e-3f : 1.2e-3f, i8 ? w.gerr : 0.f);
        const int *gate = w.ctl + CTL_OVERFLOW;
        for (int j = 0; j < k; ++j) {
            const float *ps = j ? sim_out + (j - 1) : nullptr;
            const int32_t *pi = j ? idx_out + (j - 1) : nullptr;
            if (gallery)
                hipLaunchKernelGGL((match_rerank_pairs_kernel<float>), dim3(2048), dim3(128), (size_t)D * sizeof(float), s, gallery, N, D, queries,
                                   reinterpret_cast<const MatchPair *>(w.pairs), w.pair_cap, w.ctl, w.qkey, ps, pi, k, row_offset);
            else
                hipLaunchKernelGGL((match_rerank_pairs_kernel<half_t>), dim3(2048), dim3(128), (size_t)D * sizeof(float), s, g16, N, D, queries,
                                   reinterpret_cast<const MatchPair *>(w.pairs), w.pair_cap, w.ctl, w.qkey, ps, pi, k, row_offset);
            // pair-list overflow: the unscreened exact scan answers this pass instead (gated on the flag; see the top-1 path)
            const int fb = partial_blocks < FB_BLOCKS ? partial_blocks : FB_BLOCKS;
            if (gallery) launch_t<4, false, float, true>(gallery, N, D, queries, F, partial, nullptr, fb, row_offset, s, nullptr, nullptr, ps, pi, k, gate);
            else launch_t<4, false, half_t, true>(g16, N, D, queries, F, partial, nullptr, fb, row_offset, s, nullptr, nullptr, ps, pi, k, gate);
            hipLaunchKernelGGL(match_unpack_kernel, dim3((F + 255) / 256), dim3(256), 0, s, w.qkey, F, row_offset, w.ctl, idx_out + j, sim_out + j, k, partial, fb);
        }
        return;
    }
    if (screen) screen_tiles(g16, N, D, queries, F, gmax_norm, w, k, kth_scratch, s);
    for (int j = 0; j < k; ++j) {
        const float *ps = j ? sim_out + (j - 1) : nullptr;
        const int32_t *pi = j ? idx_out + (j - 1) : nullptr;
        if (screen) {
            if (gallery)
                launch_t<1, false, float, true>(gallery, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s, w.tile_list, w.count, ps, pi, k);
            else
                launch_t<1, false, half_t, true>(g16, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s, w.tile_list, w.count, ps, pi, k);
        } else if (gallery) {
            if (F <= 32) launch_t<1, false, float, true>(gallery, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s, nullptr, nullptr, ps, pi, k);
            else launch_t<4, false, float, true>(gallery, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s, nullptr, nullptr, ps, pi, k);
        } else {
            if (F <= 32) launch_t<1, false, half_t, true>(g16, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s, nullptr, nullptr, ps, pi, k);
            else launch_t<4, false, half_t, true>(g16, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s, nullptr, nullptr, ps, pi, k);
        }
        hipLaunchKernelGGL(match_reduce_kernel, dim3(F), dim3(64), 0, s, partial, partial_blocks, F, idx_out + j, sim_out + j, k, (const int *)nullptr);
    }
}
int match_topk_max() { return TOPK_MAX; }

void launch_half_to_float(const half_t *in, long n, float *out, hipStream_t s) {  // n % 8 == 0
    hipLaunchKernelGGL(half_to_float_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, in, out, n / 8);
}
void launch_float_to_half(const float *in, long n, half_t *out, hipStream_t s) {  // n % 8 == 0
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, in, out, n / 8, (int *)nullptr, 0L);
}
void launch_merge_topk(const int32_t *idx_all, const float *sim_all, int shards, int n, int k, int32_t *idx_out, float *sim_out, hipStream_t s) {
    hipLaunchKernelGGL(merge_topk_kernel, dim3((n + 63) / 64), dim3(64), 0, s, idx_all, sim_all, shards, n, k, idx_out, sim_out);
}

void launch_match_top1_screened(const float *gallery, const half_t *g16, int N, int D, const float *queries, int F, float gmax_norm,
                                const ScreenScratch &w, MatchPartial *partial, int partial_blocks, int32_t *idx_out, float *sim_out,
                                int row_offset, hipStream_t s) {
    const int tiles = (N + BM - 1) / BM;
    if (w.pairs && w.wgmax && tiles >= COARSE_WG) {  // fast path (round 3): coarse (queries converted on load, per-workgroup maxima) -> pairs -> scalar re-rank -> unpack
        const int n_wg = COARSE_WG;
        const bool i8 = w.g8 && D == 512;  // int8 shadow (fp32-stored galleries): half the bytes of the scan, same answers
        if (i8) launch_coarse_i8(w, N, queries, F, tiles, s);
        else switch (D) {
            case 64: launch_coarse_t<64>(g16, N, nullptr, F, w.tilemax, tiles, s, queries, w.wgmax, w.ctl, w.qkey); break;
            case 128: launch_coarse_t<128>(g16, N, nullptr, F, w.tilemax, tiles, s, queries, w.wgmax, w.ctl, w.qkey); break;
            case 256: launch_coarse_t<256>(g16, N, nullptr, F, w.tilemax, tiles, s, queries, w.wgmax, w.ctl, w.qkey); break;
            default: launch_coarse_t<512>(g16, N, nullptr, F, w.tilemax, tiles, s, queries, w.wgmax, w.ctl, w.qkey); break;
        }
        // (with F > 128 every query block y writes its own columns of wgmax: [n_wg][F])
        hipLaunchKernelGGL(match_select_pairs_kernel, dim3(SEL_SEG, F), dim3(256), 0, s, w.tilemax, tiles, w.wgmax, n_wg, F, queries, D, gmax_norm,
                           reinterpret_cast<MatchPair *>(w.pairs), w.pair_cap, w.ctl, (const float *)nullptr, i8 ? 0.7e-3f : 1.2e-3f, i8 ? w.gerr : 0.f);
        const int rr_grid = 2048;  // (one pair per workgroup up to 2 048 pairs: the re-rank is one 512-long dependent fma chain per row, i.e. per-pair latency)
        if (gallery)
            hipLaunchKernelGGL((match_rerank_pairs_kernel<float>), dim3(rr_grid), dim3(128), (size_t)D * sizeof(float), s, gallery, N, D, queries,
                               reinterpret_cast<const MatchPair *>(w.pairs), w.pair_cap, w.ctl, w.qkey, (const float *)nullptr, (const int32_t *)nullptr, 1, 0);
        else
            hipLaunchKernelGGL((match_rerank_pairs_kernel<half_t>), dim3(rr_grid), dim3(128), (size_t)D * sizeof(float), s, g16, N, D, queries,
                               reinterpret_cast<const MatchPair *>(w.pairs), w.pair_cap, w.ctl, w.qkey, (const float *)nullptr, (const int32_t *)nullptr, 1, 0);
        // overflow (more candidate pairs than the list holds): the unscreened exact scan answers instead - launched always, gated on the
        // flag (FB_BLOCKS workgroups that return at once in the normal case); the unpack kernel behind it finishes either path
        const int *gate = w.ctl + CTL_OVERFLOW;
        const int fb = partial_blocks < FB_BLOCKS ? partial_blocks : FB_BLOCKS;
        if (gallery) launch_t<4, false, float>(gallery, N, D, queries, F, partial, nullptr, fb, row_offset, s, nullptr, nullptr, nullptr, nullptr, 1, gate);
        else launch_t<4, false, half_t>(g16, N, D, queries, F, partial, nullptr, fb, row_offset, s, nullptr, nullptr, nullptr, nullptr, 1, gate);
        hipLaunchKernelGGL(match_unpack_kernel, dim3((F + 255) / 256), dim3(256), 0, s, w.qkey, F, row_offset, w.ctl, idx_out, sim_out, 1, partial, fb);
        return;
    }
    const long q8 = (long)F * D / 8;
    const long nzero = (long)tiles + 1;  // tile flags + the candidate count behind them (ScreenScratch: count == tile_flags + tiles)
    const long n_thr = q8 > nzero ? q8 : nzero;
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, s, queries, w.q16, q8, w.tile_flags, nzero);
    switch (D) {
        case 64: launch_coarse_t<64>(g16, N, w.q16, F, w.tilemax, tiles, s); break;
        case 128: launch_coarse_t<128>(g16, N, w.q16, F, w.tilemax, tiles, s); break;
        case 256: launch_coarse_t<256>(g16, N, w.q16, F, w.tilemax, tiles, s); break;
        default: launch_coarse_t<512>(g16, N, w.q16, F, w.tilemax, tiles, s); break;  // match_screen_supported() gates the callers
    }
    hipLaunchKernelGGL(match_segmax_kernel, dim3(SEL_SEG, F), dim3(256), 0, s, w.tilemax, tiles * 4, w.segmax);
    hipLaunchKernelGGL(match_select_kernel, dim3(SEL_SEG, F), dim3(256), 0, s, w.tilemax, tiles, 4, w.segmax, queries, D, gmax_norm, w.tile_flags, w.tile_list,
                       w.count, (const float *)nullptr);
    // exact re-rank over the listed tiles (count lives on the device); the partial scratch is [partial_blocks][F]
    // 32 queries per workgroup (grid.y = query blocks): the list is short (a few hundred tiles), so the pass is bound by the
    // time ONE workgroup needs for a tile - 1024 fp32 MFMAs per wave with 128 queries (27 us), 256 with 32 (7 us).  Same
    // per-(row, query) arithmetic as the full scan, hence still bit-identical.
    if (gallery)
        launch_t<1, false>(gallery, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s, w.tile_list, w.count);
    else  // fp16-stored gallery: the exact pass widens the stored rows (same per-tile code path as launch_match_top1_h's full scan)
        launch_t<1, false, half_t>(g16, N, D, queries, F, partial, nullptr, partial_blocks, row_offset, s, w.tile_list, w.count);
    hipLaunchKernelGGL(match_reduce_kernel, dim3(F), dim3(64), 0, s, partial, partial_blocks, F, idx_out, sim_out, 1, (const int *)nullptr);
}
