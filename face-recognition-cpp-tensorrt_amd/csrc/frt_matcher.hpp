// struct frt_matcher: the object behind frt_matcher_* (include/frt.h).  Internal header of libfrt.so.
#pragma once
#include "frt_internal.hpp"

struct frt_matcher {
    unsigned generation = 0;  // bumped whenever gallery pointers / sizes / offsets change (invalidates captured graphs)
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    hipEvent_t ev_busy = nullptr;  // end of the last pipeline match stage that used this object's scratch (see wait_idle)
    bool busy = false;
    float *d_gallery = nullptr;  // fp32 rows [N][D]; null when the gallery is STORED as fp16 (store16)
    int N = 0, D = 0;
    int row_offset = 0;  // global index of local row 0 (sharded galleries)
    // scratch (grown on demand)
    float *d_q = nullptr, *d_sim = nullptr, *d_full = nullptr, *d_kth = nullptr;
    int32_t *d_idx = nullptr;
    static constexpr int KCAP = 16;  // d_sim / d_idx hold [q_cap][KCAP] (top-k lists of the host entry point)
    MatchPartial *d_partial = nullptr;
    int q_cap = 0;
    size_t full_cap = 0;
    int blocks = 0;
    // screened top-1 (fp16 shadow gallery; see kernels_match.hip).  Off for small galleries, for widths the coarse kernel is not
    // instantiated for (anything but 64 / 128 / 256 / 512) and with FRT_MATCH_SCREEN=0.
    half_t *d_g16 = nullptr;   // fp16 shadow of d_gallery, or the fp16-STORED gallery itself
    uint8_t *d_g8 = nullptr;   // int8 shadow of d_gallery (round 4: fp32-stored galleries with 512 columns take this instead of the fp16 shadow)
    float *d_g8_scale = nullptr;
    float gerr = 0.f;          // largest quantisation error norm of the int8 rows (part of the screening bound)
    bool store16 = false;      // current gallery is fp16-stored
    bool want16 = false;       // storage mode of the NEXT init / gallery_begin (frt_matcher_set_storage)
    float gmax_norm = 0.f;
    bool screen = false;
    bool screen_on = true;     // frt_matcher_set_screening: false = every top-1 call takes the exact fp32 scan (the shadow gallery stays resident)
    ScreenScratch scr{};
    void free_screen_scratch() {
        for (void *p : {(void *)scr.q16, (void *)scr.tilemax, (void *)scr.tile_flags, (void *)scr.tile_list, (void *)scr.segmax, (void *)scr.wgmax, scr.pairs,
                        (void *)scr.ctl, (void *)scr.qkey})  // scr.count lives behind tile_flags
            if (p) (void)hipFree(p);
        scr = ScreenScratch{};
    }
    // Object-level entry points share the scratch buffers with the pipeline's match stage, which keeps running on the pipeline's
    // stream after frt_pipeline_run_dev / submit returned: order this object's stream behind it (one event wait, no host sync).
    void wait_idle(hipStream_t s) {
        if (busy) HIPCHK(hipStreamWaitEvent(s, ev_busy, 0));
    }

    // ---- streaming gallery load (frt_matcher_gallery_begin / append / commit == initKnownEmbeds / addEmbedding / initMatMul,
    //      src/db.cpp:316-346): rows are copied into pinned staging chunks as they arrive (the caller's pointer may be SQLite's
    //      blob buffer) and every full chunk goes to the device with an asynchronous copy while the next one fills.  The previous
    //      gallery stays live (and searchable) until commit swaps the pointers.
    struct Load {
        static constexpr int NCH = 3;
        static constexpr int CH_ROWS = 4096;   // x 512 floats = 8 MB per chunk
        bool active = false;
        bool f16 = false;
        int cap = 0, D = 0, rows = 0, fill = 0, cur = 0;
        float *d_new32 = nullptr;
        half_t *d_new16 = nullptr;
        float *h_stage[NCH] = {};
        float *d_stage[NCH] = {};   // fp16 storage only: fp32 landing buffers in front of the conversion kernel
        hipEvent_t ev[NCH] = {};
        bool pending[NCH] = {};
        size_t stage_floats = 0;
        hipStream_t s = nullptr;
    } ld;
    void load_release_staging() {
        for (int i = 0; i < Load::NCH; ++i) {
            if (ld.h_stage[i]) (void)hipHostFree(ld.h_stage[i]);
            if (ld.d_stage[i]) (void)hipFree(ld.d_stage[i]);
            if (ld.ev[i]) (void)hipEventDestroy(ld.ev[i]);
            ld.h_stage[i] = ld.d_stage[i] = nullptr;
            ld.ev[i] = nullptr;
            ld.pending[i] = false;
        }
        ld.stage_floats = 0;
    }
    void load_abort() {
        if (ld.s) (void)hipStreamSynchronize(ld.s);
        if (ld.d_new32) (void)hipFree(ld.d_new32);
        if (ld.d_new16) (void)hipFree(ld.d_new16);
        ld.d_new32 = nullptr;
        ld.d_new16 = nullptr;
        ld.active = false;
    }
    void load_begin(int cap, int cols) {
        if (ld.active) load_abort();
        // The load runs on the matcher's own stream.  NOT on a stream of its own: one more hipStreamCreateWithFlags(hipStreamNonBlocking)
        // stream in the process before the pipeline's stage streams exist changes how ROCm maps those onto hardware queues, and the
        // stages of consecutive calls stop overlapping (measured: batch-1 step 0.56 -> 1.39 ms, batch-32 step +10 %).
        ld.s = stream;
        const size_t need = (size_t)Load::CH_ROWS * cols;
        if (ld.stage_floats != need) {
            load_release_staging();
            for (int i = 0; i < Load::NCH; ++i) {
                HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&ld.h_stage[i]), need * sizeof(float), hipHostMallocDefault));
                HIPCHK(hipEventCreateWithFlags(&ld.ev[i], hipEventDisableTiming));
            }
            ld.stage_floats = need;
        }
        ld.f16 = want16;
        ld.cap = cap;
        ld.D = cols;
        ld.rows = ld.fill = ld.cur = 0;
        if (cap > 0) {
            if (ld.f16) {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&ld.d_new16), gallery16_elems(cap, cols) * sizeof(half_t)));
                HIPCHK(hipMemsetAsync(ld.d_new16, 0, gallery16_elems(cap, cols) * sizeof(half_t), ld.s));  // fragment order, zero pad rows
                for (int i = 0; i < Load::NCH; ++i)
                    if (!ld.d_stage[i]) HIPCHK(hipMalloc(reinterpret_cast<void **>(&ld.d_stage[i]), need * sizeof(float)));
            } else {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&ld.d_new32), (size_t)cap * cols * sizeof(float)));
            }
        }
        ld.active = true;
    }
    void load_flush() {  // current chunk -> device
        if (!ld.fill) return;
        const int c = ld.cur;
        const size_t off = (size_t)(ld.rows - ld.fill) * ld.D, n = (size_t)ld.fill * ld.D;
        if (ld.f16) {
            HIPCHK(hipMemcpyAsync(ld.d_stage[c], ld.h_stage[c], n * sizeof(float), hipMemcpyHostToDevice, ld.s));
            launch_rows_to_half(ld.d_stage[c], (long)(ld.rows - ld.fill), (long)ld.fill, ld.D, ld.d_new16, ld.s);  // (chunks start on 128-row tiles)
        } else {
            HIPCHK(hipMemcpyAsync(ld.d_new32 + off, ld.h_stage[c], n * sizeof(float), hipMemcpyHostToDevice, ld.s));
        }
        HIPCHK(hipEventRecord(ld.ev[c], ld.s));
        ld.pending[c] = true;
        ld.cur = (c + 1) % Load::NCH;
        ld.fill = 0;
        if (ld.pending[ld.cur]) {  // the chunk about to be refilled must have left the host (and its landing buffer)
            HIPCHK(hipEventSynchronize(ld.ev[ld.cur]));
            ld.pending[ld.cur] = false;
        }
    }
    void load_append(const float *rows, int n) {
        if (!ld.active) raise(FRT_ERR_INVALID, "gallery_append: no load in progress (call frt_matcher_gallery_begin first)");
        if (n < 0 || (n > 0 && !rows)) raise(FRT_ERR_INVALID, "gallery_append: bad argument");
        if ((long)ld.rows + n > ld.cap) raise(FRT_ERR_CAPACITY, "gallery_append: more rows than gallery_begin reserved (initKnownEmbeds)");
        while (n > 0) {
            const int take = std::min(n, Load::CH_ROWS - ld.fill);
            std::memcpy(ld.h_stage[ld.cur] + (size_t)ld.fill * ld.D, rows, (size_t)take * ld.D * sizeof(float));
            ld.fill += take;
            ld.rows += take;
            rows += (size_t)take * ld.D;
            n -= take;
            if (ld.fill == Load::CH_ROWS) load_flush();
        }
    }
    // make the loaded rows THE gallery: swap pointers, rebuild the screening data, free the previous gallery
    void load_commit() {
        if (!ld.active) raise(FRT_ERR_INVALID, "gallery_commit: no load in progress");
        load_flush();
        HIPCHK(hipStreamSynchronize(ld.s));
        for (bool &p : ld.pending) p = false;
        HIPCHK(hipStreamSynchronize(stream));
        if (busy) HIPCHK(hipEventSynchronize(ev_busy));
        float *old32 = d_gallery;
        half_t *old16 = d_g16;
        if (d_g8) (void)hipFree(d_g8);  // (the streams were synchronised above: no scan is reading it)
        if (d_g8_scale) (void)hipFree(d_g8_scale);
        d_g8 = nullptr;
        d_g8_scale = nullptr;
        gerr = 0.f;
        ++generation;
        N = ld.rows;
        D = ld.D;
        store16 = ld.f16;
        d_gallery = ld.rows > 0 ? ld.d_new32 : nullptr;
        d_g16 = ld.rows > 0 ? ld.d_new16 : nullptr;
        if (ld.rows == 0) {  // empty gallery: nothing to keep
            if (ld.d_new32) (void)hipFree(ld.d_new32);
            if (ld.d_new16) (void)hipFree(ld.d_new16);
        }
        ld.d_new32 = nullptr;
        ld.d_new16 = nullptr;
        ld.active = false;
        if (old32) (void)hipFree(old32);  // (hipFree waits for the device: stages of earlier pipeline calls have finished with it)
        if (old16) (void)hipFree(old16);
        blocks = match_top1_blocks(N, 0);
        const char *scr_env = frt_tuning_env("FRT_MATCH_SCREEN");  // (tuning build only; the product's switch is frt_matcher_set_screening)
        screen = N >= 32768 && match_screen_supported(D) && !(scr_env && scr_env[0] == '0');
        gmax_norm = 0.f;
        if (N > 0 && (screen || store16)) {  // fp16 shadow copy (fp32 storage) + the largest row norm (rounding bound of the screening pass)
            int *d_bits = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_bits), sizeof(int)));
            // fp32-stored galleries of 512 columns are screened through an INT8 shadow (half the bytes of the per-call scan; kernels_match.hip);
            // FRT_MATCH_I8=0 / FRT_MATCH_FAST=0 keep the fp16 shadow (A/B measurements, the round-2 tile-list path)
            const char *i8_env = frt_tuning_env("FRT_MATCH_I8"), *fast_env = frt_tuning_env("FRT_MATCH_FAST");
            const bool use_i8 = screen && !store16 && D == 512 && !(i8_env && i8_env[0] == '0') && !(fast_env && fast_env[0] == '0');
            int *d_ebits = nullptr;
            if (store16) {
                launch_gallery_norm16(d_g16, N, D, d_bits, stream);
            } else if (use_i8) {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_ebits), sizeof(int)));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_g8), gallery8_bytes(N, D)));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_g8_scale), ((size_t)(N + 127) / 128) * 128 * sizeof(float)));
                launch_gallery_shadow8(d_gallery, N, D, d_g8, d_g8_scale, d_ebits, d_bits, stream);
            } else {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_g16), gallery16_elems(N, D) * sizeof(half_t)));
                launch_gallery_shadow(d_gallery, N, D, d_g16, d_bits, stream);
            }
            int bits = 0, ebits = 0;
            HIPCHK(hipMemcpyAsync(&bits, d_bits, sizeof(int), hipMemcpyDeviceToHost, stream));
            if (d_ebits) HIPCHK(hipMemcpyAsync(&ebits, d_ebits, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            (void)hipFree(d_bits);
            if (d_ebits) (void)hipFree(d_ebits);
            float n2, e2;
            std::memcpy(&n2, &bits, 4);
            std::memcpy(&e2, &ebits, 4);
            gmax_norm = std::sqrt(n2);
            gerr = std::sqrt(e2);
        }
        q_cap = 0;  // partial scratch depends on `blocks`
        if (d_partial) {
            (void)hipFree(d_partial);
            d_partial = nullptr;
        }
    }

    void ensure_queries(int F) {
        if (F <= q_cap && d_partial) return;
        const int cap = std::max(F, 128);
        ++generation;  // scratch buffers move
        if (d_q) (void)hipFree(d_q);
        if (d_sim) (void)hipFree(d_sim);
        if (d_idx) (void)hipFree(d_idx);
        if (d_kth) (void)hipFree(d_kth);
        if (d_partial) (void)hipFree(d_partial);
        d_q = d_sim = d_kth = nullptr;
        d_idx = nullptr;
        d_partial = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_q), (size_t)cap * D * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_sim), (size_t)cap * KCAP * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_idx), (size_t)cap * KCAP * sizeof(int32_t)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_kth), (size_t)cap * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_partial), (size_t)blocks * cap * sizeof(MatchPartial)));
        if (screen) {
            const size_t tiles = ((size_t)N + 127) / 128;
            free_screen_scratch();
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.q16), (size_t)cap * D * sizeof(half_t)));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.tilemax), (size_t)cap * tiles * 4 * sizeof(float)));  // 4 coarse entries per tile (one per wave)
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.tile_flags), (tiles + 1) * sizeof(int)));  // [tiles] flags + the candidate count:
            scr.count = scr.tile_flags + tiles;                                                           // one contiguous range to clear per call
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.tile_list), tiles * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.segmax), (size_t)cap * 16 * sizeof(float)));
            const char *fe = frt_tuning_env("FRT_MATCH_FAST");  // "0": the round-2 tile-list re-rank (diagnostics, tuning build)
            if (!(fe && fe[0] == '0')) {
                scr.pair_cap = std::max(cap * 64, 8192);  // (a multiple of the 64 sub-lists)
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.wgmax), (size_t)256 * cap * sizeof(float)));
                HIPCHK(hipMalloc(&scr.pairs, (size_t)scr.pair_cap * 8));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.ctl), FRT_MATCH_CTL_WORDS * sizeof(int)));
                HIPCHK(hipMemset(scr.ctl, 0, FRT_MATCH_CTL_WORDS * sizeof(int)));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&scr.qkey), (size_t)cap * sizeof(unsigned long long)));
            }
        }
        q_cap = cap;
    }
    // queries_dev [F][D] -> idx_dev, sim_dev (device pointers)
    void top1_dev(const float *queries_dev, int F, int32_t *idx_dev, float *sim_dev, hipStream_t s) {
        ProfScope ps(2, "match_top1", 2.0 * D * (double)N * F, s);
        // the partial scratch is [blocks][F]
        if (screen && screen_on) {  // (d_gallery == nullptr with fp16 storage: the exact re-rank then reads the stored fp16 rows)
            ScreenScratch w = scr;
            w.g8 = d_g8;
            w.g8_scale = d_g8_scale;
            w.gerr = gerr;
            launch_match_top1_screened(d_gallery, d_g16, N, D, queries_dev, F, gmax_norm, w, d_partial, blocks, idx_dev, sim_dev, row_offset, s);
        }
        else if (store16)
            launch_match_top1_h(d_g16, N, D, queries_dev, F, d_partial, blocks, idx_dev, sim_dev, row_offset, s);
        else
            launch_match_top1(d_gallery, N, D, queries_dev, F, d_partial, blocks, idx_dev, sim_dev, row_offset, s);
        HIPCHK(hipGetLastError());
    }
    // exact top-k lists [F][k] (idx_dev / sim_dev device pointers); queries fp32 on the device
    void topk_dev(const float *queries_dev, int F, int k, int32_t *idx_dev, float *sim_dev, hipStream_t s) {
        ProfScope ps(2, "match_topk", 2.0 * D * (double)N * F, s);
        ScreenScratch w = scr;
        w.g8 = d_g8;
        w.g8_scale = d_g8_scale;
        w.gerr = gerr;
        launch_match_topk(d_gallery, d_g16, N, D, queries_dev, F, k, screen, gmax_norm, w, d_kth, d_partial, blocks, idx_dev, sim_dev, row_offset, s);
        HIPCHK(hipGetLastError());
    }
};

