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
#include "frt_kernels.h"

#include <limits.h>

namespace {

constexpr int BM = 128;  // gallery rows per tile
constexpr int BK = 32;   // floats per k-step

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return (v > bv) || (v == bv && i < bi); }

template <int NQ, bool FULL>
__global__ __launch_bounds__(256) void match_kernel(const float *__restrict__ G, int N, int D, const float *__restrict__ E, int F,
                                                    MatchPartial *__restrict__ partial, float *__restrict__ out_full, int num_tiles,
                                                    int row_offset) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);                        // [2][BM][BK]
    float *Bs = reinterpret_cast<float *>(smem) + 2 * BM * BK;          // [2][NQ*32][BK]
    constexpr int QT = NQ * 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int q0 = blockIdx.y * QT;
    const int ksteps = D / BK;

    const int my_tiles = (num_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * ksteps;

    const int ld_row = tid >> 3, ld_ch = tid & 7;

    floatx4 ga[4], qa[NQ];
    auto load_global = [&](int it) {
        const int tile = blockIdx.x + (it / ksteps) * gridDim.x;
        const int k0 = (it % ksteps) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ld_row + 32 * i;
            const long g = (long)tile * BM + row;
            ga[i] = g < N ? *reinterpret_cast<const floatx4 *>(G + g * D + k0 + ld_ch * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
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
            const int tile = blockIdx.x + (it / ksteps) * gridDim.x;
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
                        if (g < N && better(v, g, bv[n], bi[n])) {
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
                partial[(long)blockIdx.x * F + q] = p;
            }
        }
    }
}

// one wave per query: lanes stride over the workgroup partials, then a butterfly with the same first-maximum rule
__global__ __launch_bounds__(64) void match_reduce_kernel(const MatchPartial *__restrict__ partial, int blocks, int F,
                                                          int32_t *__restrict__ idx_out, float *__restrict__ sim_out) {
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
        idx_out[q] = i == INT_MAX ? -1 : i;
        sim_out[q] = v;
    }
}

template <int NQ, bool FULL>
void launch_t(const float *G, int N, int D, const float *E, int F, MatchPartial *partial, float *out_full, int blocks, int row_offset,
              hipStream_t s) {
    const int tiles = (N + BM - 1) / BM;
    const size_t lds = (size_t)2 * (BM + NQ * 32) * BK * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_kernel<NQ, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid(blocks, (F + NQ * 32 - 1) / (NQ * 32));
    hipLaunchKernelGGL((match_kernel<NQ, FULL>), grid, dim3(256), lds, s, G, N, D, E, F, partial, out_full, tiles, row_offset);
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
    hipLaunchKernelGGL(match_reduce_kernel, dim3(F), dim3(64), 0, s, partial, partial_blocks, F, idx_out, sim_out);
}

void launch_match_full(const float *gallery, int N, int D, const float *queries, int F, float *out, hipStream_t s) {
    const int blocks = match_top1_blocks(N, F);
    if (F <= 32)
        launch_t<1, true>(gallery, N, D, queries, F, nullptr, out, blocks, 0, s);
    else
        launch_t<4, true>(gallery, N, D, queries, F, nullptr, out, blocks, 0, s);
}
