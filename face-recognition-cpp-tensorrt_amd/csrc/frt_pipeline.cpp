// libfrt.so: the pipeline's C ABI (frt_pipeline_*): device-resident calls, the asynchronous host boundary, pairing / merging of calls.
// All device work is hand-written HIP (kernels_*.hip); there is no CPU fallback anywhere in this file: without a HIP
// device every entry point that needs one fails with FRT_ERR_DEVICE.
#include "frt_pipeline.hpp"

extern "C" {

// ------------------------------------------------------------------------------------------------------------ pipeline
int frt_pipeline_create(frt_detector *d, frt_embedder *e, frt_matcher *m, int max_frames, frt_pipeline **out) {
    return guarded([&] {
        if (!d || !e || !out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (max_frames < 1 || max_frames > d->max_batch) raise(FRT_ERR_CAPACITY, "pipeline: max_frames exceeds det_maxBatchSize");
        if (d->device != e->device || (m && m->device != d->device)) raise(FRT_ERR_INVALID, "pipeline: objects live on different devices");
        use_device(d->device);
        std::unique_ptr<frt_pipeline> p(new frt_pipeline);
        p->det = d; p->emb = e; p->mat = m;
        p->max_frames = max_frames;
        p->max_faces = d->g.max_faces;
        p->F_cap = max_frames * p->max_faces;
        // the pipeline's own join stream is created on first use: ROCm maps streams onto 4 hardware queues round-robin and streams that
        // share a queue serialise, so a stream nobody uses (callers usually pass theirs) should not take a slot among the stage streams
        p->stream = nullptr;
        // the stage streams are created at the highest stream priority: ROCm keeps a separate hardware-queue pool per priority, so they
        // never share a queue with the caller's (normal priority) stream, whose queue holds the pending joins of the batches in flight
        // (the FRT_PIPELINE_* switches below are A/B switches of measurement builds - frt_tuning_env is getenv under make TUNING=1 and nullptr in
        //  the product; the product's switches are frt_pipeline_set_overlap / _set_graph / _check_overlap)
        int prio_lo = 0, prio_hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        {
            const char *pe = frt_tuning_env("FRT_PIPELINE_STREAM_PRIO");
            if (pe && pe[0] == '0') prio_hi = 0;   // "0": normal priority (stage streams share the caller's queue pool)
        }
        // hipStreamDefault (blocking), not hipStreamNonBlocking: a gallery reload between calls (hipFree / hipMalloc / synchronous
        // hipMemcpy on the legacy default stream) is then ordered against the stages still in flight without the caller
        // synchronising anything (tests/test_gpu_pipeline.py::test_gallery_reload_between_pipelined_calls); non-blocking
        // streams also measured 1 % slower
        p->copy_prio = prio_hi;
        auto mk = [&](hipStream_t *st) { HIPCHK(hipStreamCreateWithPriority(st, hipStreamDefault, prio_hi)); };
        {
            // FRT_PIPELINE_DET_PRIO=lo / normal: the detector's stream below the recogniser's (A/B: does the hardware then give the recogniser -
            // the longer stage - the CUs first and let the detector fill its gaps?)
            const char *dp = frt_tuning_env("FRT_PIPELINE_DET_PRIO");
            if (dp && dp[0] == 'l') HIPCHK(hipStreamCreateWithPriority(&p->det_stream, hipStreamDefault, prio_lo));
            else if (dp && dp[0] == 'n') HIPCHK(hipStreamCreateWithPriority(&p->det_stream, hipStreamDefault, 0));
            else mk(&p->det_stream);
        }
        mk(&p->emb_stream);
        mk(&p->emb_stream2);
        {
            const char *de = frt_tuning_env("FRT_PIPELINE_DUAL_EMBED");
            p->dual_embed = !(de && de[0] == '0');
        }
        if (p->dual_embed) {
            e->ensure_alt();
            p->d_chw2 = p->arena.alloc<float>((size_t)max_frames * d->g.max_faces * 3 * 112 * 112);
        }
        const size_t F = (size_t)p->F_cap;
        HIPCHK(hipEventCreateWithFlags(&p->ev_serial, hipEventDisableTiming));
        for (int i = 0; i < frt_pipeline::NSLOT; ++i) {
            HIPCHK(hipEventCreateWithFlags(&p->ev_det[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p->ev_emb[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p->ev_done[i], hipEventDisableTiming));
            p->slot_embeds[i] = p->arena.alloc<float>(F * 512);
            p->slot_valid[i] = p->arena.alloc<int>(F);
            p->slot_boxes[i] = p->arena.alloc<frt_bbox>(F);
            p->slot_nout[i] = p->arena.alloc<int>((size_t)max_frames);
            if (d->has_landmarks) p->slot_landmarks[i] = p->arena.alloc<float>(F * 10);
        }
        {
            const char *e = frt_tuning_env("FRT_PIPELINE_OVERLAP");
            p->overlap = !(e && e[0] == '0');
            const char *gph = frt_tuning_env("FRT_PIPELINE_GRAPH");
            p->use_graphs = gph && gph[0] == '1';  // opt-in: measured no gain on this workload (see the note at run_part)
        }
        HIPCHK(hipEventCreateWithFlags(&p->ev_input, hipEventDisableTiming));
        p->d_chw = p->arena.alloc<float>(F * 3 * 112 * 112);
        p->d_sim = p->arena.alloc<float>(F);
        p->d_idx = p->arena.alloc<int32_t>(F);
        if (p->overlap) {  // create-time self-check of the stage streams (~1 ms)
            const char *sc = frt_tuning_env("FRT_PIPELINE_SELFCHECK");
            if (!(sc && sc[0] == '0')) p->self_check(false);
        }
        *out = p.release();
    });
}

static void pipeline_flush_locked(frt_pipeline *p);
static void pipeline_start_held(frt_pipeline *p);

void frt_pipeline_destroy(frt_pipeline *p) {
    if (!p) return;
    (void)hipSetDevice(p->det->device);
    if (p->npend || p->held.on) {  // pairing / merging: calls still waiting for partners run now - a submitted batch is never dropped
        try {
            std::lock_guard<std::mutex> lk(p->run_mu);
            pipeline_flush_locked(p);
        } catch (...) {
        }
    }
    if (p->det_stream) (void)hipStreamSynchronize(p->det_stream);
    if (p->emb_stream) (void)hipStreamSynchronize(p->emb_stream);
    if (p->emb_stream2) (void)hipStreamSynchronize(p->emb_stream2);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    p->drop_graphs();
    if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    if (p->det_stream) (void)hipStreamDestroy(p->det_stream);
    if (p->emb_stream) (void)hipStreamDestroy(p->emb_stream);
    if (p->emb_stream2) (void)hipStreamDestroy(p->emb_stream2);
    if (p->ev_serial) (void)hipEventDestroy(p->ev_serial);
    if (p->ev_input) (void)hipEventDestroy(p->ev_input);
    if (p->copy_stream) {
        (void)hipStreamSynchronize(p->copy_stream);
        (void)hipStreamDestroy(p->copy_stream);
    }
    for (frt_pipeline::AsyncBuf &b : p->abuf) {
        if (b.ev_h2d) (void)hipEventDestroy(b.ev_h2d);
        if (b.ev_out) (void)hipEventDestroy(b.ev_out);
    }
    for (int i = 0; i < frt_pipeline::NSLOT; ++i) {
        if (p->ev_det[i]) (void)hipEventDestroy(p->ev_det[i]);
        if (p->ev_emb[i]) (void)hipEventDestroy(p->ev_emb[i]);
        if (p->ev_done[i]) (void)hipEventDestroy(p->ev_done[i]);
    }
    p->arena.release();
    delete p;
}

// Caller holds p->run_mu.
static void pipeline_lock_run(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev) {
    if (n_frames < 1 || n_frames > p->max_frames) raise(FRT_ERR_CAPACITY, "pipeline: more frames than max_frames");
    std::lock_guard<std::mutex> l1(p->det->mu);
    std::lock_guard<std::mutex> l2(p->emb->mu);
    std::unique_lock<std::mutex> l3;
    if (p->mat) {
        l3 = std::unique_lock<std::mutex>(p->mat->mu);
        if (p->mat->N > 0) p->mat->ensure_queries(p->F_cap);
    }
    p->run(reinterpret_cast<const uint8_t *>(frames_dev), n_frames, reinterpret_cast<frt_face_result *>(results_dev),
           reinterpret_cast<float *>(embeds_dev));
}

int frt_pipeline_run_dev(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev) {
    return guarded([&] {
        if (!p || !frames_dev || !results_dev) raise(FRT_ERR_INVALID, "null argument");
        use_device(p->det->device);
        std::lock_guard<std::mutex> lk(p->run_mu);
        pipeline_start_held(p);  // (submits held back at the host boundary go first)
        pipeline_lock_run(p, frames_dev, n_frames, results_dev, embeds_dev);
    });
}

int frt_pipeline_run_dev_after(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev, void *ready_event) {
    return guarded([&] {
        if (!p || !frames_dev || !results_dev) raise(FRT_ERR_INVALID, "null argument");
        use_device(p->det->device);
        std::lock_guard<std::mutex> lk(p->run_mu);
        pipeline_start_held(p);
        p->ev_ready = reinterpret_cast<hipEvent_t>(ready_event);
        try {
            pipeline_lock_run(p, frames_dev, n_frames, results_dev, embeds_dev);
        } catch (...) {
            p->ev_ready = nullptr;
            throw;
        }
    });
}

int frt_pipeline_check_overlap(frt_pipeline *p, float *ratio_out) {
    int rc = guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        use_device(p->det->device);
        std::lock_guard<std::mutex> la(p->async_mu);  // same order as pipeline_submit_impl: async_mu, then run_mu
        std::lock_guard<std::mutex> lk(p->run_mu);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        if (p->stream) HIPCHK(hipStreamSynchronize(p->stream));
        p->ensure_stream();
        p->ensure_async();  // the upload stream of frt_pipeline_submit / run takes part
        p->self_check(true);
        if (ratio_out) *ratio_out = p->overlap_ratio;
    });
    if (rc == FRT_OK && p && !p->warning.empty()) frthost::last_error() = p->warning;  // FRT_OK + a message: a warning, not a failure
    return rc;
}

int frt_pipeline_set_input_sync(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        p->input_sync = enable != 0;
    });
}

int frt_pipeline_sync(frt_pipeline *p) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        use_device(p->det->device);
        {
            std::lock_guard<std::mutex> lk(p->run_mu);
            pipeline_flush_locked(p);
        }
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->emb->check_se_error();
    });
}

int frt_pipeline_set_stream(frt_pipeline *p, void *hip_stream) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : p->own_stream;  // null own_stream: created at the next run
    });
}

int frt_pipeline_set_overlap(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->overlap = enable != 0;
        p->seq = 0;
        p->drop_graphs();
    });
}

int frt_pipeline_set_graph(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->use_graphs = enable != 0;
        p->drop_graphs();
    });
}

int frt_pipeline_set_align(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        if (enable && !p->det->has_landmarks) raise(FRT_ERR_FORMAT, "pipeline: alignment needs a detector blob with the LandmarkHead");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        HIPCHK(hipStreamSynchronize(p->det_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream));
        HIPCHK(hipStreamSynchronize(p->emb_stream2));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->align = enable != 0;
    });
}

int frt_pipeline_set_pairing(frt_pipeline *p, int enable) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        use_device(p->det->device);
        pipeline_flush_locked(p);
        p->group = enable < 0 ? -1 : (enable == 0 ? 0 : std::min(std::max(enable, 2), (int)frt_pipeline::MAXG));
        p->adaptive_dev = enable == -2;
        p->merge_submits = enable != -3;
    });
}

int frt_pipeline_merge_stats(frt_pipeline *p, long *merged_calls, long *merged_tickets) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        if (merged_calls) *merged_calls = p->merged_calls;
        if (merged_tickets) *merged_tickets = p->merged_tickets;
    });
}

int frt_pipeline_graph_stats(frt_pipeline *p, long *captured, long *replayed) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        if (captured) *captured = p->graphs_captured;
        if (replayed) *replayed = p->graphs_replayed;
    });
}

int frt_pipeline_pairing_stats(frt_pipeline *p, long *paired_passes, long *single_passes) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lk(p->run_mu);
        if (paired_passes) *paired_passes = p->paired_passes;
        if (single_passes) *single_passes = p->single_passes;
    });
}

static void pipeline_lock_run(frt_pipeline *p, const void *frames_dev, int n_frames, void *results_dev, void *embeds_dev);

// Caller holds p->run_mu: the held submits (merged at the host boundary) go out as ONE call.  Never throws: a failure is left on the tickets
// (AsyncBuf::failed, reported by frt_pipeline_wait) - the caller of the moment may be somebody else's submit or wait.
static void pipeline_start_held(frt_pipeline *p) {
    if (!p->held.on) return;
    frt_pipeline::Held h = p->held;
    p->held = frt_pipeline::Held{};
    p->ev_frames = h.base->ev_h2d;  // recorded behind the last ticket's upload
    p->crops_req = h.want_crops ? h.base->d_crops : nullptr;
    p->serial_call = false;
    p->host_req = frt_pipeline::CallRec{};
    p->host_req.nsub = h.nsub;
    for (int j = 0; j < h.nsub; ++j) p->host_req.sub[j] = h.sub[j];
    try {
        pipeline_lock_run(p, h.base->d_frames, h.n, h.base->d_results, h.want_embeds ? h.base->d_embeds : nullptr);
        p->merged_calls += h.nsub > 1;
        p->merged_tickets += h.nsub > 1 ? h.nsub : 0;
    } catch (const std::exception &e) {
        p->held_error = e.what();
        p->ev_frames = nullptr;
        p->crops_req = nullptr;
        p->host_req = frt_pipeline::CallRec{};
        for (int j = 0; j < h.nsub; ++j) {
            h.sub[j].ab->failed = true;
            (void)hipEventRecord(h.sub[j].ab->ev_out, p->stream);
        }
    }
}

// queue one batch through a staging set; caller holds neither mutex
static long pipeline_submit_impl(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out, bool synchronous = false,
                                 uint8_t *crops_host = nullptr) {
    if (n_frames < 1 || n_frames > p->max_frames) raise(FRT_ERR_CAPACITY, "pipeline: more frames than max_frames");
    use_device(p->det->device);
    std::lock_guard<std::mutex> lk(p->async_mu);   // staging sets + ticket order
    std::lock_guard<std::mutex> lr(p->run_mu);     // the stage enqueue itself (shared with frt_pipeline_run_dev)
    p->ensure_stream();
    p->ensure_async();
    const long ticket = p->next_ticket;
    frt_pipeline::AsyncBuf &b = p->abuf[ticket % frt_pipeline::NBUF];
    if (b.ticket >= 0) {
        // (a ticket that is still held back has no "results have left" event yet: its set's event is its previous occupant's)
        for (int j = 0; j < p->held.nsub; ++j)
            if (p->held.on && p->held.sub[j].ticket == b.ticket) pipeline_start_held(p);
        if (p->is_pending(b.ticket)) pipeline_flush_locked(p);
        wait_event_spinning(b.ev_out);  // the staging set is free once its previous batch has left
    }
    b.failed = false;
    hipStream_t s = p->stream;
    const size_t fbytes = (size_t)p->det->g.frame_h * p->det->g.frame_w * 3;
    // ---- adaptive merging at the host boundary (frt_pipeline::Held): join the held call, or become one when the detector is busy
    {
        const int K = p->max_faces;
        const bool mergeable = p->group < 0 && p->merge_submits && p->overlap && g_prof_kind == 0;
        frt_pipeline::Sub me;
        me.ab = &b;
        me.h_results = results;
        me.h_embeds = embeds_out;
        me.h_crops = crops_host;
        me.n = n_frames;
        me.ticket = ticket;
        auto join = [&](frt_pipeline::Held &h) {  // this ticket's frames behind the held ones, in the FIRST ticket's staging set
            HIPCHK(hipMemcpyAsync(h.base->d_frames + fbytes * (size_t)h.n, frames, fbytes * n_frames, hipMemcpyHostToDevice, p->copy_stream));
            HIPCHK(hipEventRecord(h.base->ev_h2d, p->copy_stream));
            h.sub[h.nsub++] = me;
            h.n += n_frames;
            h.want_embeds = h.want_embeds || embeds_out;
            h.want_crops = h.want_crops || crops_host;
            b.ticket = ticket;
            p->next_ticket = ticket + 1;
        };
        if (p->held.on) {
            const int nt = p->held.n + n_frames;
            if (mergeable && p->held.nsub < frt_pipeline::MAXSUB && nt <= p->max_frames && nt * K <= p->emb->max_batch) {
                join(p->held);
                if (p->held.nsub == frt_pipeline::MAXSUB || 2 * p->held.n > p->max_frames || !p->backed_up() || p->tickets_running() < frt_pipeline::HOLD_MIN)
                    pipeline_start_held(p);
                return ticket;
            }
            pipeline_start_held(p);  // cannot join: first in, first out
        }
        if (mergeable && 2 * n_frames <= p->max_frames && 2 * n_frames * K <= p->emb->max_batch && p->backed_up() && p->tickets_running() >= frt_pipeline::HOLD_MIN) {
            p->held.on = true;
            p->held.base = &b;
            join(p->held);
            return ticket;
        }
    }
    // A synchronous call that finds nothing else in flight (the reference's request / reply shape: one frame, one caller) has nothing to
    // overlap with: upload, detector, recogniser, match and download go down ONE stream - no stream-to-stream event hand-overs on its
    // critical path (5 of them otherwise; one 4-face call 1.02 -> 0.94 ms, profiles/r03/r03u_sync_overlap.txt).  Calls that arrive while
    // another is in flight take the stage streams as before (and are ordered behind this one through ev_serial).
    bool lone = synchronous && p->overlap && !p->npend && !p->held.on;
    for (int i = 0; lone && i < frt_pipeline::NBUF; ++i)
        if (p->abuf[i].ticket >= 0 && i != (int)(ticket % frt_pipeline::NBUF) && hipEventQuery(p->abuf[i].ev_out) != hipSuccess) lone = false;
    if (lone) {
        HIPCHK(hipMemcpyAsync(b.d_frames, frames, fbytes * n_frames, hipMemcpyHostToDevice, s));
    } else {
        HIPCHK(hipMemcpyAsync(b.d_frames, frames, fbytes * n_frames, hipMemcpyHostToDevice, p->copy_stream));
        HIPCHK(hipEventRecord(b.ev_h2d, p->copy_stream));
        p->ev_frames = b.ev_h2d;  // the stages that read the frames (detector, crop) wait for the copy; the caller's stream does not
    }
    p->serial_call = lone;
    p->crops_req = crops_host ? b.d_crops : nullptr;
    // the downloads and the "results have left" event are queued by the pipeline behind this call's match stage - now, or (pairing) with the next call
    p->host_req = frt_pipeline::CallRec{};
    p->host_req.nsub = 1;
    p->host_req.sub[0].ab = &b;
    p->host_req.sub[0].h_results = results;
    p->host_req.sub[0].h_embeds = embeds_out;
    p->host_req.sub[0].h_crops = crops_host;
    p->host_req.sub[0].n = n_frames;
    p->host_req.sub[0].ticket = ticket;
    try {
        pipeline_lock_run(p, b.d_frames, n_frames, b.d_results, embeds_out ? b.d_embeds : nullptr);
    } catch (...) {
        p->ev_frames = nullptr;
        p->crops_req = nullptr;
        p->host_req = frt_pipeline::CallRec{};
        p->serial_call = false;
        throw;
    }
    p->serial_call = false;
    b.ticket = ticket;
    p->next_ticket = ticket + 1;
    return ticket;
}

// Caller holds p->run_mu: queue the later stages of a call that is waiting for a partner (pairing).
static void pipeline_flush_locked(frt_pipeline *p) {
    pipeline_start_held(p);  // (takes the object mutexes itself)
    if (!p->npend) return;
    std::lock_guard<std::mutex> l1(p->det->mu);
    std::lock_guard<std::mutex> l2(p->emb->mu);
    std::unique_lock<std::mutex> l3;
    if (p->mat) l3 = std::unique_lock<std::mutex>(p->mat->mu);
    p->flush_pending();
}

static void pipeline_wait_impl(frt_pipeline *p, long ticket) {
    use_device(p->det->device);
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(p->async_mu);
        if (ticket < 0 || ticket >= p->next_ticket) raise(FRT_ERR_INVALID, "pipeline: unknown ticket");
        frt_pipeline::AsyncBuf &b = p->abuf[ticket % frt_pipeline::NBUF];
        if (b.ticket > ticket) return;  // its staging set was reused, which submit only does after that batch completed
        {
            std::lock_guard<std::mutex> lr(p->run_mu);
            // pairing: the partners that would share its recogniser pass have not come; adaptive pairing: calls held back behind a busy
            // recogniser go out as soon as a waiting caller finds it idle (they would be running by now had they not been held)
            if (p->held.on) {  // submits merged at the host boundary: one of its tickets is being waited for, or the detector has gone idle
                bool mine = false;
                for (int j = 0; j < p->held.nsub; ++j) mine = mine || p->held.sub[j].ticket == ticket;
                if (mine || !p->backed_up() || p->tickets_running() < frt_pipeline::HOLD_MIN) pipeline_start_held(p);
            }
            if (p->is_pending(ticket) || (p->npend && p->group < 0 && !p->recogniser_busy())) pipeline_flush_locked(p);
        }
        ev = b.ev_out;
    }
    wait_event_spinning(ev);
    {
        std::lock_guard<std::mutex> lk(p->async_mu);
        frt_pipeline::AsyncBuf &b = p->abuf[ticket % frt_pipeline::NBUF];
        if (b.ticket == ticket && b.failed)
            raise(FRT_ERR_DEVICE, "pipeline: the held stages of this call could not be queued" + (p->held_error.empty() ? std::string() : ": " + p->held_error));
    }
    p->emb->check_se_error();
}

// Synchronous host entry point.  Thread-safe: every call takes its own staging set (device frames / results / embeddings) under the
// pipeline's mutexes, so concurrent callers (the reference's Crow server is .multithreaded(), src/app.cpp:367) never share a buffer;
// with several threads calling, their batches overlap in the stage pipeline exactly like submit()/wait() batches do.
int frt_pipeline_run(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out) {
    return guarded([&] {
        if (!p || !frames || !results) raise(FRT_ERR_INVALID, "null argument");
        const long t = pipeline_submit_impl(p, frames, n_frames, results, embeds_out, true);
        pipeline_wait_impl(p, t);
    });
}

int frt_pipeline_submit(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out, long *ticket_out) {
    return guarded([&] {
        if (!p || !frames || !results || !ticket_out) raise(FRT_ERR_INVALID, "null argument");
        *ticket_out = pipeline_submit_impl(p, frames, n_frames, results, embeds_out);
    });
}

int frt_pipeline_submit_crops(frt_pipeline *p, const uint8_t *frames, int n_frames, frt_face_result *results, float *embeds_out, uint8_t *crops_out,
                              long *ticket_out) {
    return guarded([&] {
        if (!p || !frames || !results || !ticket_out) raise(FRT_ERR_INVALID, "null argument");
        *ticket_out = pipeline_submit_impl(p, frames, n_frames, results, embeds_out, false, crops_out);
    });
}

int frt_pipeline_wait(frt_pipeline *p, long ticket) {
    return guarded([&] {
        if (!p) raise(FRT_ERR_INVALID, "null argument");
        pipeline_wait_impl(p, ticket);
    });
}


}  // extern "C"
