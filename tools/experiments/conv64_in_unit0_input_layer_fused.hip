// PARKED EXPERIMENT (round 4) - not part of the build.  Unit 0 of the recogniser with the input layer fused into conv1's patch build.
// Context: face-recognition-cpp-tensorrt_amd/csrc/kernels_arc_c64.hip (this code sat behind conv64_kernel and uses its constants SW, PW,
// PROWB, PATCH_B, RING, EROW); frt_embedder::forward called launch_conv64_in(conv1 args, input-layer args) instead of launch_arc_input +
// conv1's launch.  Result: bit-identical embeddings (tests/test_gpu_embedder.py, goldens), 379 us per launch at 128 faces against
// 78 + 169 us for the two kernels it replaces (profiles/r04/r04k_unit0_fused_in.txt) - see the note in kernels_arc_c64.hip.

// ---------------------------------------------------------------- unit 0: the input layer fused into conv1's patch (round 4)
// Unit 0 of the recogniser moved 1 024 MB through three kernels (input layer 256 MB, conv1 410, stride-2 conv2 358): the input layer wrote
// z = BN(PReLU(BN(conv3x3(x)))) for every pixel (205 MB) only for conv1 to read it back.  Here conv1's workgroup computes its strip's halo'd
// z patch ITSELF from the 3-channel crop (model_irse.py:139-141 + the BatchNorm that opens unit 0, :58): per strip 4 x 58 patch pixels =
// 8 MFMA pixel tiles, four per wave, exactly arc_input_mfma_kernel's arithmetic (K = 27 taps + bias slot, BN folded into the fp16 weights,
// PReLU and the second BN in fp32, one rounding to fp16) written straight into the LDS patch - same bits as the two-kernel path.  Patch
// pixels outside the image are zeros (conv1 pads z, not x).  The raw activation y (unit 0's MaxPool(1, 2) shortcut) leaves for the even
// pixels of the strip's own two rows.  The crop taps of the NEXT strip are requested before the current strip's K loop and turned into
// the patch behind it, so their latency hides under the MFMAs; the patch build itself (4 x [16 gathers, 4 MFMAs, 64 fp32 epilogue values,
// 8 ds_write_b64]) is serial work in a kernel that was waiting on LDS, not on instruction issue.
__global__ __launch_bounds__(128) void conv64_in_kernel(ConvMfmaArgs p, ArcInputArgs ia, int n_strips) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *patch = smem;                                           // [2][PATCH_B]
    float *etile = reinterpret_cast<float *>(smem + 2 * PATCH_B); // [2 waves][32][EROW]
    const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    constexpr int H = 112, W = 112, HW = H * W;
    constexpr int strips_x = W / SW, strips_per_img = (H / 2) * strips_x;

    const int nwg = gridDim.x;
    const int bq = nwg >> 3, brem = nwg & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wid = (xcd < brem ? xcd * (bq + 1) : brem * (bq + 1) + (xcd - brem) * bq) + slot;
    const int k_full = n_strips / nwg, rem_strips = n_strips - k_full * nwg;
    auto strip_of = [&](int k) {
        if (k < k_full) return wid + k * nwg;
        return (k == k_full && (int)blockIdx.x < rem_strips) ? k_full * nwg + (int)blockIdx.x : n_strips;
    };

    // ---- conv1 weights (as conv64_kernel) and epilogue parameters (PReLU)
    half8 wreg[9][4];
    {
        const half_t *wrow = p.w + (long)(cb * 32 + r) * 576 + 8 * hi;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wreg[tap][kk] = *reinterpret_cast<const half8 *>(wrow + tap * 64 + kk * 16);
    }
    const int ec0 = cb * 32 + (lane & 3) * 8;
    float q0[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q0[e] = p.p0[ec0 + e];

    // ---- input layer: weights [64 cout][32 k] (lane (cout r of block c, half hi): k = ks*16 + 8*hi .. +7), per-lane tap table and the
    //      parameters of the 32 channels a lane owns after the MFMA: c = cbi*32 + 8*q + 4*hi + j
    half8 wa[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wa[c][ks] = *reinterpret_cast<const half8 *>(ia.wh + (c * 32 + r) * 32 + ks * 16 + 8 * hi);
    int koff[16];
    unsigned need[16];  // bit0 needs the row above, bit1 the row below, bit2 the column left, bit3 right, bit4 = constant one, bit5 = zero
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = (s >> 3) * 16 + 8 * hi + (s & 7);
        const int ci = k / 9, rem = k - ci * 9, kh = rem / 3, kw = rem - kh * 3;
        koff[s] = k < 27 ? ci * HW + (kh - 1) * W + (kw - 1) : 0;
        need[s] = k < 27 ? ((kh == 0 ? 1u : 0u) | (kh == 2 ? 2u : 0u) | (kw == 0 ? 4u : 0u) | (kw == 2 ? 8u : 0u)) : (k == 27 ? 16u : 32u);
    }
    // (the 3 x 64 per-channel parameters of the input layer live in LDS: as 96 registers per lane they pushed the kernel into spills)
    float *prm = etile + 2 * 32 * EROW;  // [slope | s1 | b1][64]
    if (tid < 64) {
        prm[tid] = ia.slope[tid];
        prm[64 + tid] = ia.s1[tid];
        prm[128 + tid] = ia.b1[tid];
    }
    __syncthreads();

    // ---- B fragment bases of conv1 (as conv64_kernel)
    int bbase[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = 32 * t + r;
        const int qq = q < 2 * SW ? q : 0;
        const int row = qq / SW, col = qq - row * SW;
        bbase[t] = (row * PW + col) * PROWB + hi * 16;
    }

    // ---- the patch builder, in two halves: gather (global loads only) and build (everything else).  This wave owns patch tiles cb, cb + 2,
    //      cb + 4, cb + 6; lane pixel pp = tile * 32 + r of the 4 x 58 patch (pp >= 232: nothing)
    struct Geo {
        int b, sy, sx;
    };
    auto geo_of = [&](int strip) {
        Geo g;
        g.b = strip / strips_per_img;
        const int rem = strip - g.b * strips_per_img;
        g.sy = rem / strips_x;
        g.sx = rem - g.sy * strips_x;
        return g;
    };
    float xv[4][16];
    auto gather = [&](const Geo &g) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int pp = (cb + 2 * t) * 32 + r;
            const int pr = pp / PW, pc = pp - pr * PW;
            const int iy = g.sy * 2 - 1 + pr, ix = g.sx * SW - 1 + pc;
            const bool in = pp < 4 * PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const unsigned edge = (iy == 0 ? 1u : 0u) | (iy == H - 1 ? 2u : 0u) | (ix == 0 ? 4u : 0u) | (ix == W - 1 ? 8u : 0u);
            const float *xb = ia.x + (long)g.b * 3 * HW + (in ? iy * W + ix : 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const bool live = in && (need[s] & (edge | 48u)) == 0;  // unconditional load from a clamped address, masked in build()
                xv[t][s] = xb[live ? koff[s] : 0];
            }
        }
    };
    auto build = [&](const Geo &g, char *dst) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int pp = (cb + 2 * t) * 32 + r;
            const int pr = pp / PW, pc = pp - pr * PW;
            const int iy = g.sy * 2 - 1 + pr, ix = g.sx * SW - 1 + pc;
            const bool in = pp < 4 * PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const unsigned edge = (iy == 0 ? 1u : 0u) | (iy == H - 1 ? 2u : 0u) | (ix == 0 ? 4u : 0u) | (ix == W - 1 ? 8u : 0u);
            half8 bf[2];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const bool live = in && (need[s] & (edge | 48u)) == 0;
                const float v = live ? xv[t][s] : ((need[s] & 16u) ? 1.f : 0.f);
                bf[s >> 3][s & 7] = (half_t)v;
            }
            floatx16 acc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[c][0], bf[0], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[c][1], bf[1], acc[c], 0, 0, 0);
            }
            // y leaves for the even pixels of the strip's own rows (patch row 1 = image row 2 sy); z goes into the patch (zeros outside the image)
            const bool own_even = in && pr == 1 && pc >= 1 && pc <= SW && (ix & 1) == 0;
            half_t *yrow = ia.y + (((long)g.b * (H / 2) + g.sy) * (W / 2) + (ix >> 1)) * 64;
            char *zrow = dst + pp * PROWB;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = c * 32 + 8 * q + 4 * hi;
                    const floatx4 sl = *reinterpret_cast<const floatx4 *>(prm + c0);
                    const floatx4 s1 = *reinterpret_cast<const floatx4 *>(prm + 64 + c0);
                    const floatx4 b1 = *reinterpret_cast<const floatx4 *>(prm + 128 + c0);
                    half4 y4, z4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[c][4 * q + j];
                        v = v > 0.f ? v : v * sl[j];
                        y4[j] = (half_t)v;
                        z4[j] = in ? (half_t)(v * s1[j] + b1[j]) : (half_t)0.f;
                    }
                    if (pp < 4 * PW) *reinterpret_cast<half4 *>(zrow + c0 * 2) = z4;
                    if (own_even) *reinterpret_cast<half4 *>(yrow + c0) = y4;
                }
        }
    };

    int k = 0;
    int strip = strip_of(0);
    if (strip >= n_strips) return;
    {
        const Geo g0 = geo_of(strip);
        gather(g0);
        build(g0, patch);
    }
    __syncthreads();
    int cur = 0;

    float *et = etile + cb * 32 * EROW;
    for (;;) {
        const int next = strip_of(k + 1);
        const bool has_next = next < n_strips;
        const Geo gn = geo_of(has_next ? next : strip);
        if (has_next) gather(gn);  // the next strip's crop taps: in flight under this strip's K loop

        floatx16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const char *pb = patch + cur * PATCH_B;
        half8 bf[RING][4];
        auto read_b = [&](int s, half8 (&dst)[4]) {
            const int tap = s >> 2, kk = s & 3;
            const int kh = tap / 3, kw = tap - kh * 3;
            const int off = (kh * PW + kw) * PROWB + kk * 32;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = *reinterpret_cast<const half8 *>(pb + bbase[t] + off);
        };
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) read_b(s, bf[s]);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + RING - 1 < 36) read_b(s + RING - 1, bf[(s + RING - 1) % RING]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s >> 2][s & 3], bf[s % RING][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue of this strip (PReLU), as conv64_kernel
        {
            const Geo g = geo_of(strip);
            const long img_base = (long)g.b * HW;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    floatx4 v = {acc[t][4 * gq], acc[t][4 * gq + 1], acc[t][4 * gq + 2], acc[t][4 * gq + 3]};
                    *reinterpret_cast<floatx4 *>(et + r * EROW + 8 * gq + 4 * hi) = v;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int px = (lane >> 2) + 16 * i;
                    const floatx4 v0 = *reinterpret_cast<const floatx4 *>(et + px * EROW + (lane & 3) * 8);
                    const floatx4 v1 = *reinterpret_cast<const floatx4 *>(et + px * EROW + (lane & 3) * 8 + 4);
                    const int q = 32 * t + px;
                    if (q >= 2 * SW) continue;
                    const int row = q / SW, col = q - row * SW;
                    const long m = img_base + (long)(g.sy * 2 + row) * W + g.sx * SW + col;
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * q0[e];
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                    *reinterpret_cast<half8 *>(p.out0 + m * 64 + ec0) = o;
                }
            }
        }
        if (!has_next) break;
        build(gn, patch + (cur ^ 1) * PATCH_B);  // that buffer's readers retired at the last barrier
        __syncthreads();
        cur ^= 1;
        strip = next;
        ++k;
    }
}

// Unit 0's conv1 with the input layer inside it (conv64_in_kernel): `a` = conv1's arguments (PReLU epilogue, a.x unused), `ia` = the input
// layer's (ia.z unused).  false: not this shape - the caller launches the two kernels.
bool launch_conv64_in(const ConvMfmaArgs &a, const ArcInputArgs &ia, hipStream_t s) {
    static const bool off = frt_tuning_env("FRT_ARC_FUSED_IN") && frt_tuning_env("FRT_ARC_FUSED_IN")[0] == '0';
    if (off || !ia.wh || ia.H != 112 || ia.W != 112 || a.H != 112 || a.W != 112 || a.mode != EPI_PRELU || !conv64_applies(a) || a.B != ia.F) return false;
    const int n_strips = a.B * (a.H / 2) * (a.W / SW);
    int grid = 512;
    if (grid > n_strips) grid = n_strips;
    const size_t lds = 2 * PATCH_B + 2 * 32 * EROW * sizeof(float) + 3 * 64 * sizeof(float);
    static bool attr_done[FRT_MAX_DEVICES] = {};
    if (frt_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv64_in_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv64_in_kernel, dim3(grid), dim3(128), lds, s, a, ia, n_strips);
    return true;
}
