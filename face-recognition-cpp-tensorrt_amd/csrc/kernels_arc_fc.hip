// ArcFace output layer: Linear 25088 -> 512 over the NHWC-flattened 7x7x512 map (model_irse.py:143-147), fp16 operands, fp32
// accumulation, as 49 K-slices (one 7x7 position = 512 channels each) whose partial sums fc_finalize_kernel adds in slice order
// (deterministic) before bias + BatchNorm1d + L2 normalisation.
//
// It ran on the generic im2col kernel (conv_glds_kernel, EPI_PARTIAL): 62 us for 3.3 GFLOP and 32 MB - a K step of 64 channels through
// a two-stage LDS ring is all latency at this shape.  A slice is exactly the coarse match's problem (kernels_match.hip): 128 "queries"
// (the faces' activations of one 7x7 position, 128 KB) resident in LDS, 128 weight rows per workgroup streamed straight into A-fragment
// registers from a FRAGMENT-ORDERED copy of the weights ([32-output block][k step][lane][8 halfs]: every load of a wave is one
// contiguous kilobyte, all 32 k steps of the slice in flight at once), 128 MFMAs per wave, one partial tile out.
// grid = 4 output tiles x 49 slices = 196 workgroups, one round on 256 CUs.
#include "frt_kernels.h"

namespace {

constexpr int FC_K = 25088, FC_O = 512, FC_SLICE = 512, FC_KS = FC_SLICE / 16;  // 32 k steps per slice
constexpr int FC_QP = FC_SLICE + 8;                                              // halves per face row in LDS (16 B pad: conflict-free b128 reads)

__global__ __launch_bounds__(256) void fc_slice_kernel(const half_t *__restrict__ z, const half_t *__restrict__ wfrag, int F, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem_fc[];
    half_t *Zs = reinterpret_cast<half_t *>(smem_fc);  // [128][FC_QP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int otile = blockIdx.x & 3, slice = blockIdx.x >> 2;
    const int f0 = blockIdx.y * 128;  // face block (batches above 128 faces take grid.y > 1)
    // weights first (HBM / L2 -> registers, 32 KB per wave in flight), then the activations of this slice into LDS
    const int ob = otile * 4 + wave;  // 32-output block of this wave
    const half_t *wp = wfrag + (((long)ob * (FC_K / 16) + (long)slice * FC_KS) * 64 + lane) * 8;
    half8 areg[FC_KS];
#pragma unroll
    for (int ks = 0; ks < FC_KS; ++ks) areg[ks] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(wp + (long)ks * 512));
    for (int i = tid; i < 128 * (FC_SLICE / 8); i += 256) {
        const int f = i / (FC_SLICE / 8), c = i - f * (FC_SLICE / 8);
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (f0 + f < F) v = *reinterpret_cast<const half8 *>(z + (long)(f0 + f) * FC_K + slice * FC_SLICE + c * 8);
        *reinterpret_cast<half8 *>(Zs + f * FC_QP + c * 8) = v;
    }
    __syncthreads();
    floatx16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    const half_t *qb = Zs + r * FC_QP + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < FC_KS; ++ks) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const half8 bq = *reinterpret_cast<const half8 *>(qb + n * 32 * FC_QP + ks * 16);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[ks], bq, acc[n], 0, 0, 0);
        }
    }
    // acc[n][e]: output ob*32 + (e & 3) + 8*(e >> 2) + 4*hi of face n*32 + r  ->  partial[slice][face][output]
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int f = f0 + n * 32 + r;
        if (f >= F) continue;
        float *dst = partial + ((long)slice * F + f) * FC_O + ob * 32 + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 v = {acc[n][4 * g], acc[n][4 * g + 1], acc[n][4 * g + 2], acc[n][4 * g + 3]};
            *reinterpret_cast<floatx4 *>(dst + 8 * g) = v;
        }
    }
}

}  // namespace

// z [F][25088] fp16, wfrag: fragment-ordered weights, partial [49][F][512] fp32
void launch_fc_slices(const half_t *z, const half_t *wfrag, int F, float *partial, hipStream_t s) {
    constexpr size_t lds = (size_t)128 * FC_QP * sizeof(half_t);
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fc_slice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fc_slice_kernel, dim3(4 * (FC_K / FC_SLICE), (F + 127) / 128), dim3(256), lds, s, z, wfrag, F, partial);
}
