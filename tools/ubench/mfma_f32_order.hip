// Which scalar formula reproduces v_mfma_f32_32x32x2_f32 bit for bit?  The exact match kernel (csrc/kernels_match.hip) accumulates
// S[g][q] over k with that instruction, two k values per issue (k index 0 from lanes 0-31, index 1 from lanes 32-63).  A scalar re-rank
// of a few (query, 128-row tile) pairs can replace the MFMA re-rank pass only if it returns the SAME bits.  Candidates per issue:
//   A  acc = fma(a1, b1, fma(a0, b0, acc))          sequential fused multiply-adds, k index 0 first
//   B  acc = fma(a0, b0, fma(a1, b1, acc))          ... k index 1 first
//   C  acc = acc + (a0*b0 + a1*b1)                  products rounded, summed, then added
//   D  acc = fma(a0, b0, acc) + a1*b1 (unfused 2nd) etc. are not tried: A or B is what "an fmaf chain" means.
// Random operands (normal range, mixed signs, long chains of 512 k like the gallery rows), 1M dot products; prints the mismatch counts.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int D = 512;

// one wave: G rows 0..31 (A operand), Q cols 0..31 (B operand) -> acc[e]: row (e&3) + 8*(e>>2) + 4*hi, col r
__global__ __launch_bounds__(64) void mfma_dots(const float *G, const float *Q, float *out) {
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
    const float *g = G + (long)blockIdx.x * 32 * D, *q = Q + (long)blockIdx.x * 32 * D;
    floatx16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k = 0; k < D; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[r * D + k + hi], q[r * D + k + hi], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) out[((long)blockIdx.x * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi) * 32 + r] = acc[e];
}

__global__ void scalar_dots(const float *G, const float *Q, float *outA, float *outB, float *outC, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (block, row, col)
    if (i >= n) return;
    const long blk = i / 1024;
    const int row = (int)(i / 32 % 32), col = (int)(i % 32);
    const float *g = G + (blk * 32 + row) * D, *q = Q + (blk * 32 + col) * D;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int k = 0; k < D; k += 2) {
        a = __builtin_fmaf(g[k + 1], q[k + 1], __builtin_fmaf(g[k], q[k], a));
        b = __builtin_fmaf(g[k], q[k], __builtin_fmaf(g[k + 1], q[k + 1], b));
        const float p0 = g[k] * q[k], p1 = g[k + 1] * q[k + 1];
        c = c + (p0 + p1);
    }
    outA[i] = a;
    outB[i] = b;
    outC[i] = c;
}

int main() {
    const int blocks = 1024;
    const long n = (long)blocks * 1024;
    std::vector<float> hg((size_t)blocks * 32 * D), hq(hg.size());
    srand(7);
    for (size_t i = 0; i < hg.size(); ++i) {
        hg[i] = (float)((rand() % 200001) - 100000) * 1e-5f * 0.0442f;  // ~ unit-norm rows like the gallery
        hq[i] = (float)((rand() % 200001) - 100000) * 1e-5f * 0.0442f;
    }
    float *G, *Q, *om, *oa, *ob, *oc;
    (void)hipMalloc(&G, hg.size() * 4); (void)hipMalloc(&Q, hq.size() * 4);
    (void)hipMalloc(&om, n * 4); (void)hipMalloc(&oa, n * 4); (void)hipMalloc(&ob, n * 4); (void)hipMalloc(&oc, n * 4);
    (void)hipMemcpy(G, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(Q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_dots, dim3(blocks), dim3(64), 0, 0, G, Q, om);
    hipLaunchKernelGGL(scalar_dots, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, G, Q, oa, ob, oc, n);
    (void)hipDeviceSynchronize();
    std::vector<float> m(n), a(n), b(n), c(n);
    (void)hipMemcpy(m.data(), om, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(a.data(), oa, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), ob, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), oc, n * 4, hipMemcpyDeviceToHost);
    long da = 0, db = 0, dc = 0;
    double maxrel = 0;
    for (long i = 0; i < n; ++i) {
        da += m[i] != a[i]; db += m[i] != b[i]; dc += m[i] != c[i];
        if (m[i] != 0) maxrel = std::fmax(maxrel, std::fabs((double)m[i] - a[i]) / std::fabs((double)m[i]));
    }
    std::printf("dots %ld  mismatches vs MFMA:  A(seq fma, k0 first) %ld   B(seq fma, k1 first) %ld   C(rounded products) %ld   max rel diff A %.3g\n", n, da, db, dc, maxrel);
    return 0;
}
