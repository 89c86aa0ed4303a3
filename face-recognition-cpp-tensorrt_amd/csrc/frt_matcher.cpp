// libfrt.so: the matcher object's C ABI (frt_matcher_*, top-1 / top-k merges, pinned host memory).
// All device work is hand-written HIP (kernels_*.hip); there is no CPU fallback anywhere in this file: without a HIP
// device every entry point that needs one fails with FRT_ERR_DEVICE.
#include "frt_matcher.hpp"

extern "C" {

// ------------------------------------------------------------------------------------------------------------- matcher
int frt_matcher_create(int device, frt_matcher **out) {
    return guarded([&] {
        if (!out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        use_device(device);
        std::unique_ptr<frt_matcher> m(new frt_matcher);
        m->device = device;
        HIPCHK(hipStreamCreate(&m->stream));
        HIPCHK(hipEventCreateWithFlags(&m->ev_busy, hipEventDisableTiming));
        *out = m.release();
    });
}

void frt_matcher_destroy(frt_matcher *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    // an unfinished streaming load runs on m->stream (ld.s == stream): abort it while the stream still exists
    m->load_abort();
    m->load_release_staging();
    m->ld.s = nullptr;
    if (m->stream) {
        (void)hipStreamSynchronize(m->stream);
        (void)hipStreamDestroy(m->stream);
    }
    if (m->ev_busy) (void)hipEventDestroy(m->ev_busy);
    for (void *p : {(void *)m->d_gallery, (void *)m->d_q, (void *)m->d_sim, (void *)m->d_idx, (void *)m->d_partial, (void *)m->d_full, (void *)m->d_g16, (void *)m->d_kth,
                    (void *)m->d_g8, (void *)m->d_g8_scale})
        if (p) (void)hipFree(p);
    m->free_screen_scratch();
    delete m;
}

static void check_gallery_shape(int num_row, int num_col) {
    if (num_row < 0) raise(FRT_ERR_INVALID, "MatMul::init: bad argument");
    if (num_col < 32 || num_col % 32) raise(FRT_ERR_INVALID, "MatMul::init: numCol must be a multiple of 32");
}

int frt_matcher_set_storage(frt_matcher *m, int fp16) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        m->want16 = fp16 != 0;
    });
}

int frt_matcher_set_screening(frt_matcher *m, int on) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        m->screen_on = on != 0;
    });
}

unsigned frt_matcher_generation(frt_matcher *m) {
    if (!m) return 0;
    std::lock_guard<std::mutex> lk(m->mu);
    return m->generation;
}

size_t frt_matcher_scan_bytes(frt_matcher *m) {
    if (!m) return 0;
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t n = (size_t)m->N, d = (size_t)m->D;
    if (m->screen && m->screen_on) return (m->d_g8 ? 1 : 2) * n * d;   // the coarse scan reads the shadow copy once per call
    return (m->store16 ? 2 : 4) * n * d;
}

int frt_matcher_init(frt_matcher *m, const float *gallery, int num_row, int num_col) {
    return guarded([&] {
        if (!m || (num_row > 0 && !gallery)) raise(FRT_ERR_INVALID, "MatMul::init: bad argument");
        check_gallery_shape(num_row, num_col);
        std::lock_guard<std::mutex> lk(m->mu);
        use_device(m->device);
        // one path for every gallery load: pinned staging chunks + asynchronous copies (idempotent: the previous device copy is
        // freed at commit - the reference leaks it on every /reload)
        m->load_begin(num_row, num_col);
        try {
            m->load_append(gallery, num_row);
            m->load_commit();
        } catch (...) {
            m->load_abort();
            throw;
        }
    });
}

int frt_matcher_gallery_begin(frt_matcher *m, int row_capacity, int num_col) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        check_gallery_shape(row_capacity, num_col);
        std::lock_guard<std::mutex> lk(m->mu);
        use_device(m->device);
        m->load_begin(row_capacity, num_col);
    });
}

int frt_matcher_gallery_append(frt_matcher *m, const void *rows, int n_rows) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        use_device(m->device);
        m->load_append(reinterpret_cast<const float *>(rows), n_rows);
    });
}

int frt_matcher_gallery_commit(frt_matcher *m) {
    return guarded([&] {
        if (!m) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        use_device(m->device);
        try {
            m->load_commit();
        } catch (...) {
            m->load_abort();
            throw;
        }
    });
}

int frt_matcher_num_rows(const frt_matcher *m) { return m ? m->N : 0; }

int frt_matcher_set_row_offset(frt_matcher *m, int row_offset) {
    return guarded([&] {
        if (!m || row_offset < 0) raise(FRT_ERR_INVALID, "set_row_offset: bad argument");
        std::lock_guard<std::mutex> lk(m->mu);
        m->row_offset = row_offset;
        ++m->generation;
    });
}

int frt_matcher_calculate(frt_matcher *m, const float *embeds, int embed_count, float *outputs) {
    return guarded([&] {
        if (!m || !embeds || !outputs) raise(FRT_ERR_INVALID, "MatMul::calculate: null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = m->stream;
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        const size_t need = (size_t)embed_count * m->N;
        if (need > m->full_cap) {
            if (m->d_full) (void)hipFree(m->d_full);
            m->d_full = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_full), need * sizeof(float)));
            m->full_cap = need;
        }
        HIPCHK(hipMemcpyAsync(m->d_q, embeds, sizeof(float) * (size_t)embed_count * m->D, hipMemcpyHostToDevice, s));
        for (int f0 = 0; f0 < embed_count; f0 += 128) {
            const int nf = std::min(128, embed_count - f0);
            if (m->store16)
                launch_match_full_h(m->d_g16, m->N, m->D, m->d_q + (size_t)f0 * m->D, nf, m->d_full + (size_t)f0 * m->N, s);
            else
                launch_match_full(m->d_gallery, m->N, m->D, m->d_q + (size_t)f0 * m->D, nf, m->d_full + (size_t)f0 * m->N, s);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(outputs, m->d_full, need * sizeof(float), hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_matcher_calculate_top1(frt_matcher *m, const float *embeds, int embed_count, float *outputs, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (!m || !embeds || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "calculate_top1: null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = m->stream;
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        HIPCHK(hipMemcpyAsync(m->d_q, embeds, sizeof(float) * (size_t)embed_count * m->D, hipMemcpyHostToDevice, s));
        if (outputs) {
            const size_t need = (size_t)embed_count * m->N;
            if (need > m->full_cap) {
                if (m->d_full) (void)hipFree(m->d_full);
                m->d_full = nullptr;
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_full), need * sizeof(float)));
                m->full_cap = need;
            }
            for (int f0 = 0; f0 < embed_count; f0 += 128) {
                const int nf = std::min(128, embed_count - f0);
                if (m->store16)
                    launch_match_full_h(m->d_g16, m->N, m->D, m->d_q + (size_t)f0 * m->D, nf, m->d_full + (size_t)f0 * m->N, s);
                else
                    launch_match_full(m->d_gallery, m->N, m->D, m->d_q + (size_t)f0 * m->D, nf, m->d_full + (size_t)f0 * m->N, s);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(outputs, m->d_full, need * sizeof(float), hipMemcpyDeviceToHost, s));  // (the top-1 search below runs under this copy's tail)
        }
        m->top1_dev(m->d_q, embed_count, m->d_idx, m->d_sim, s);
        HIPCHK(hipMemcpyAsync(idx_out, m->d_idx, sizeof(int32_t) * embed_count, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sim_out, m->d_sim, sizeof(float) * embed_count, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_pinned_alloc(size_t bytes, int device, void **out) {
    return guarded([&] {
        if (!out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (device >= 0) use_device(device);
        HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    });
}

void frt_pinned_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int frt_matcher_top1(frt_matcher *m, const float *embeds, int embed_count, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (!m || !embeds || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "top1: null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = m->stream;
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        HIPCHK(hipMemcpyAsync(m->d_q, embeds, sizeof(float) * (size_t)embed_count * m->D, hipMemcpyHostToDevice, s));
        m->top1_dev(m->d_q, embed_count, m->d_idx, m->d_sim, s);
        HIPCHK(hipMemcpyAsync(idx_out, m->d_idx, sizeof(int32_t) * embed_count, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sim_out, m->d_sim, sizeof(float) * embed_count, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

/* device-resident queries (sharded-gallery path, dist.py: the all-gathered embeddings never visit the host) */
int frt_matcher_top1_dev(frt_matcher *m, const void *embeds_dev, int embed_count, void *idx_dev, void *sim_dev, void *hip_stream) {
    return guarded([&] {
        if (!m || !embeds_dev || !idx_dev || !sim_dev) raise(FRT_ERR_INVALID, "top1_dev: null argument");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        m->top1_dev(reinterpret_cast<const float *>(embeds_dev), embed_count, reinterpret_cast<int32_t *>(idx_dev), reinterpret_cast<float *>(sim_dev), s);
        HIPCHK(hipEventRecord(m->ev_busy, s));  // the scratch stays in use until this call has run
        m->busy = true;
    });
}

int frt_merge_top1(int n, const int32_t *idx_a, const float *sim_a, const int32_t *idx_b, const float *sim_b, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (n < 0 || !idx_a || !sim_a || !idx_b || !sim_b || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "merge: bad argument");
        for (int i = 0; i < n; ++i) {
            const bool a_ok = idx_a[i] >= 0, b_ok = idx_b[i] >= 0;
            bool take_b = false;
            if (!a_ok)
                take_b = b_ok;
            else if (b_ok)
                take_b = (sim_b[i] > sim_a[i]) || (sim_b[i] == sim_a[i] && idx_b[i] < idx_a[i]);
            idx_out[i] = take_b ? idx_b[i] : idx_a[i];
            sim_out[i] = take_b ? sim_b[i] : sim_a[i];
        }
    });
}

static void check_k(int k) {
    if (k < 1 || k > match_topk_max() || k > frt_matcher::KCAP) raise(FRT_ERR_INVALID, "top-k: k must be in 1..16");
}

int frt_matcher_topk(frt_matcher *m, const float *embeds, int embed_count, int k, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (!m || !embeds || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "topk: null argument");
        check_k(k);
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = m->stream;
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        HIPCHK(hipMemcpyAsync(m->d_q, embeds, sizeof(float) * (size_t)embed_count * m->D, hipMemcpyHostToDevice, s));
        m->topk_dev(m->d_q, embed_count, k, m->d_idx, m->d_sim, s);
        HIPCHK(hipMemcpyAsync(idx_out, m->d_idx, sizeof(int32_t) * (size_t)embed_count * k, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sim_out, m->d_sim, sizeof(float) * (size_t)embed_count * k, hipMemcpyDeviceToHost, s));
        sync_stream_spinning(s);
    });
}

int frt_matcher_topk_dev(frt_matcher *m, const void *embeds_dev, int embeds_fp16, int embed_count, int k, void *idx_dev, void *sim_dev, void *hip_stream) {
    return guarded([&] {
        if (!m || !embeds_dev || !idx_dev || !sim_dev) raise(FRT_ERR_INVALID, "topk_dev: null argument");
        check_k(k);
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->N <= 0 || embed_count <= 0) raise(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found");
        use_device(m->device);
        hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
        m->wait_idle(s);
        m->ensure_queries(embed_count);
        const float *q = reinterpret_cast<const float *>(embeds_dev);
        if (embeds_fp16) {  // exact widening into the query scratch
            launch_half_to_float(reinterpret_cast<const half_t *>(embeds_dev), (long)embed_count * m->D, m->d_q, s);
            q = m->d_q;
        }
        m->topk_dev(q, embed_count, k, reinterpret_cast<int32_t *>(idx_dev), reinterpret_cast<float *>(sim_dev), s);
        HIPCHK(hipEventRecord(m->ev_busy, s));  // the scratch stays in use until this call has run
        m->busy = true;
    });
}

int frt_merge_topk(int shards, int n, int k, const int32_t *idx_all, const float *sim_all, int32_t *idx_out, float *sim_out) {
    return guarded([&] {
        if (shards < 1 || n < 0 || k < 1 || !idx_all || !sim_all || !idx_out || !sim_out) raise(FRT_ERR_INVALID, "merge_topk: bad argument");
        std::vector<int> pos((size_t)shards);
        for (int q = 0; q < n; ++q) {
            std::fill(pos.begin(), pos.end(), 0);
            for (int o = 0; o < k; ++o) {
                int best = -1, bi = 0;
                float bv = 0.f;
                for (int sh = 0; sh < shards; ++sh) {
                    while (pos[(size_t)sh] < k && idx_all[((size_t)sh * n + q) * k + pos[(size_t)sh]] < 0) ++pos[(size_t)sh];  // empty slots
                    if (pos[(size_t)sh] >= k) continue;
                    const size_t e = ((size_t)sh * n + q) * k + pos[(size_t)sh];
                    const float v = sim_all[e];
                    const int i = idx_all[e];
                    if (best < 0 || v > bv || (v == bv && i < bi)) {
                        best = sh;
                        bv = v;
                        bi = i;
                    }
                }
                if (best < 0) {
                    idx_out[(size_t)q * k + o] = -1;
                    sim_out[(size_t)q * k + o] = -INFINITY;
                } else {
                    idx_out[(size_t)q * k + o] = bi;
                    sim_out[(size_t)q * k + o] = bv;
                    ++pos[(size_t)best];
                }
            }
        }
    });
}

int frt_merge_topk_dev(int shards, int n, int k, const void *idx_all_dev, const void *sim_all_dev, void *idx_out_dev, void *sim_out_dev, void *hip_stream) {
    return guarded([&] {
        if (shards < 1 || n < 0 || k < 1 || !idx_all_dev || !sim_all_dev || !idx_out_dev || !sim_out_dev) raise(FRT_ERR_INVALID, "merge_topk_dev: bad argument");
        if (n == 0) return;
        launch_merge_topk(reinterpret_cast<const int32_t *>(idx_all_dev), reinterpret_cast<const float *>(sim_all_dev), shards, n, k,
                          reinterpret_cast<int32_t *>(idx_out_dev), reinterpret_cast<float *>(sim_out_dev), reinterpret_cast<hipStream_t>(hip_stream));
        HIPCHK(hipGetLastError());
    });
}

int frt_embeds_to_half_dev(const void *embeds_dev, size_t n_values, void *half_out_dev, void *hip_stream) {
    return guarded([&] {
        if (!embeds_dev || !half_out_dev || n_values % 8) raise(FRT_ERR_INVALID, "embeds_to_half: bad argument (n_values must be a multiple of 8)");
        if (n_values == 0) return;
        launch_float_to_half(reinterpret_cast<const float *>(embeds_dev), (long)n_values, reinterpret_cast<half_t *>(half_out_dev),
                             reinterpret_cast<hipStream_t>(hip_stream));
        HIPCHK(hipGetLastError());
    });
}


}  // extern "C"
