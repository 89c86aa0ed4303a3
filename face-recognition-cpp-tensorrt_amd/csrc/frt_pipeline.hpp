// struct frt_pipeline: the batched, three-stage pipeline behind frt_pipeline_* (include/frt.h).  Internal header of libfrt.so.
#pragma once
#include "frt_detector.hpp"
#include "frt_embedder.hpp"
#include "frt_matcher.hpp"

struct frt_pipeline {
    frt_detector *det;
    frt_embedder *emb;
    frt_matcher *mat;
    int max_frames, max_faces, F_cap;
    hipStream_t stream = nullptr, own_stream = nullptr;
    // Three-stage software pipeline over consecutive calls: detector of call b+1 (det_stream), crop + recogniser of call b
    // (emb_stream / emb_stream2 alternately), match + pack of call b-1 (behind its recogniser pass on the same stream, i.e. beside
    // the other set's pass) - three stages with different bottlenecks (latency / MFMA+LDS / HBM) that overlap on the same CUs.  `stream` (the caller's) only joins.  Fork/join with events; boxes, embeddings and validity
    // flags of a call live in one of two slots so that a later stage of the previous call can still read them.
    hipStream_t det_stream = nullptr, emb_stream = nullptr, emb_stream2 = nullptr;
    bool dual_embed = true;   // recogniser passes of consecutive calls on two streams with two activation sets (FRT_PIPELINE_DUAL_EMBED=0: one)
    float *d_chw2 = nullptr;
    hipEvent_t ev_serial = nullptr;  // end of the last serial (profiled) call while overlap is on
    bool serial_pending = false;
    // calls in flight between the start of D and the end of M (2: 32.5k, 3: 33.6k, 4: 33.6k faces/s).  Six since pairing exists: paired calls finish
    // two at a time and one call late, so two pairs in the later stages + the detector a call or two ahead need six slots (with three the detector
    // of call b + 5 waited for the pair (b + 2, b + 3) and the recogniser passes ran one after the other: 0.84 instead of 0.74 ms per 4-frame call)
    static constexpr int NSLOT = 10;  // (groups of four: 2 * 4 + 2)
    hipEvent_t ev_det[NSLOT] = {}, ev_emb[NSLOT] = {}, ev_done[NSLOT] = {};
    float *slot_embeds[NSLOT] = {};
    int *slot_valid[NSLOT] = {};
    frt_bbox *slot_boxes[NSLOT] = {};
    int *slot_nout[NSLOT] = {};
    float *slot_landmarks[NSLOT] = {};
    bool align = false;  // optional: 5-point similarity warp instead of the reference's bbox crop + bicubic resize
    unsigned seq = 0;
    bool overlap = true;
    bool serial_call = false;  // this call only: every stage on the caller's stream (a synchronous call with nothing else in flight, see frt_pipeline_run)
    Arena arena;
    float *d_chw, *d_sim;
    int32_t *d_idx;
    std::mutex run_mu;               // serialises run(): stream selection, slot counters and the stage enqueue order are per-call state
    bool input_sync = false;         // frt_pipeline_set_input_sync: order every run_dev call behind the work queued on `stream` so far
    hipEvent_t ev_input = nullptr;   // ... recorded on `stream` at the call
    hipEvent_t ev_ready = nullptr;   // caller's "frames are ready" event of frt_pipeline_run_dev_after (borrowed, one call)

    // ---- asynchronous host boundary (frt_pipeline_submit / frt_pipeline_wait): NBUF staging sets so that the H2D copy of batch
    //      b+1 (copy_stream, the SDMA engine) and the D2H of batch b-1 run under the stages of batch b
    static constexpr int NBUF = 12;  // (4 until pairing: up to eleven batches between submit and wait)
    struct AsyncBuf {
        uint8_t *d_frames = nullptr;
        frt_face_result *d_results = nullptr;
        float *d_embeds = nullptr;
        uint8_t *d_crops = nullptr;  // u8 BGR 112x112 crops of the batch's faces (frt_pipeline_submit_crops)
        hipEvent_t ev_h2d = nullptr, ev_out = nullptr;
        long ticket = -1;  // ticket whose results ev_out guards; -1: never used
        std::atomic<bool> failed{false};  // the held stages of this ticket could not be queued (flush_pending / start_held): frt_pipeline_wait reports it
    };
    AsyncBuf abuf[NBUF];
    hipStream_t copy_stream = nullptr;
    int copy_prio = 0;
    hipEvent_t ev_frames = nullptr;  // set by submit for the next run(): the detector stream waits for it
    uint8_t *crops_req = nullptr;    // set by submit for the next run(): the crop kernel also writes the u8 crops there
    long next_ticket = 0;
    std::mutex async_mu;

    // ---- pairing (frt_pipeline_set_pairing; off by default).  A recogniser pass over 16 faces costs 0.59 ms, one over 32 faces 0.92 ms
    //      (profiles/r05z_small_batch_layers.txt: below ~ 64 faces a pass is a chain of launch latencies, not work), and one match call scans
    //      the gallery once whatever the number of queries.  With pairing on, the crop + recogniser + match stages of TWO consecutive calls
    //      run as one pass: a call's detector stage is queued at the call as always, its later stages wait for the next call (or for a
    //      flush: frt_pipeline_wait on its ticket, frt_pipeline_sync, any mode switch).  Nothing about a result changes except when it is
    //      ready - one call later - and which batch-size class of recogniser kernels produced it (the class of the two calls' faces together).
    //      A call is only ever deferred when it could be paired: both calls' face slots together must fit this pipeline's max_frames *
    //      max_faces and the recogniser's max_batch, i.e. create the pipeline for twice the frames a call carries.
    struct Sub {  // one frt_pipeline_submit ticket inside a call: `n` frames, where its results go, the staging set that carries its events
        AsyncBuf *ab = nullptr;
        frt_face_result *h_results = nullptr;
        float *h_embeds = nullptr;
        uint8_t *h_crops = nullptr;
        int n = 0;
        long ticket = -1;
    };
    static constexpr int MAXSUB = 4;
    struct CallRec {
        bool on = false;
        unsigned call = 0;
        int slot = 0, n = 0;
        const uint8_t *frames = nullptr;
        frt_face_result *results = nullptr;
        float *embeds = nullptr;
        uint8_t *crops = nullptr;
        // host side of frt_pipeline_submit: the downloads of this call's results follow its match stage, wherever that is queued.  One entry per
        // ticket: a call is the frames of up to MAXSUB consecutive submits when they were merged at the host boundary (Held, below)
        int nsub = 0;
        Sub sub[MAXSUB];
    };
    static constexpr int MAXG = 4;  // calls per recogniser pass at most
    CallRec pend[MAXG];  // the calls whose later stages are still to be queued (fewer than `group` of them)
    int npend = 0;
    CallRec host_req;  // set by submit for the next run(): staging set + host destinations
    // group: 0 off; 2 .. MAXG: ALWAYS wait for that many calls per recogniser pass (results up to group - 1 calls late);
    //        -1 (default, round 6) ADAPTIVE: a call's later stages are held back only while the recogniser is still busy with earlier calls -
    //        the pass could not start now anyway, so waiting for the next call costs a lone caller nothing - and go out together with the
    //        next call's (up to MAXG calls per pass while the backlog lasts).  A call that finds the recogniser idle is queued at once,
    //        exactly like group == 0.  frt_pipeline_wait on ANY ticket releases held calls once the recogniser has gone idle.
    //        Only calls that came through frt_pipeline_submit are held (their contract is the ticket); frt_pipeline_run_dev promises that
    //        the pipeline stream joins the results AT the call, so device-resident calls are held only on request (adaptive_dev).
    int group = -1;
    bool adaptive_dev = false;
    bool merge_submits = true;   // adaptive mode: merge held submits at the host boundary (Held, below)
    // ---- merging of consecutive submits at the host boundary (adaptive mode only, round 6).  A small call's DETECTOR stage is as much a chain of
    //      launch latencies as its recogniser pass (4 frames: 258 us of kernels, 32 frames: 809 - profiles/r06g_det_tables.txt).  A submit that
    //      finds the pipeline backed up (detector or recogniser still busy with earlier calls) is not queued at all: its frames are uploaded into its staging set and the call is
    //      HELD; the next submit's frames go into the same staging set behind them, and the held frames then run as ONE call (one detector pass,
    //      one recogniser pass, one match call; per-ticket result downloads).  Released by: the submit that fills it (MAXSUB tickets / the
    //      pipeline's capacity) or finds the detector idle, any submit that cannot join, frt_pipeline_wait on one of its tickets - or on any
    //      ticket once the detector has gone idle -, run_dev, sync, set_*, destroy.  A call that finds the detector idle is never held.
    struct Held {
        bool on = false;
        AsyncBuf *base = nullptr;  // the staging set that holds the frames / results / embeddings / crops of every ticket of the call
        int n = 0, nsub = 0;
        bool want_embeds = false, want_crops = false;
        Sub sub[MAXSUB];
    } held;
    long merged_calls = 0, merged_tickets = 0;
    std::string held_error;  // why the last held call could not be queued
    bool detector_busy() const { return det->busy && hipEventQuery(det->ev_busy) == hipErrorNotReady; }
    // "backed up": a stage of an earlier call is still running or queued.  (The detector alone is the wrong signal: under load the recogniser is
    // the bottleneck and the detector is often idle at the moment of a submit - single 4-frame calls then slip in between the merged ones:
    // measured 0.559 ms per 4-frame step against 0.524 without any merging.)
    bool backed_up() const { return detector_busy() || recogniser_busy(); }
    // tickets whose stages are queued and whose results have not left yet (held / pending ones are not counted: nothing of theirs is queued)
    int tickets_running() const {
        int n = 0;
        for (const AsyncBuf &b : abuf) {
            if (b.ticket < 0 || is_pending(b.ticket)) continue;
            bool h = false;
            for (int j = 0; held.on && j < held.nsub; ++j) h = h || held.sub[j].ticket == b.ticket;
            if (!h && hipEventQuery(b.ev_out) == hipErrorNotReady) ++n;
        }
        return n;
    }
    // Holding is only free while the GPU has enough queued work to stay busy until the held frames are released: a submit is held only when at
    // least HOLD_MIN tickets are running, and the held ones go out as soon as fewer are.  Measured with 4-frame calls (tools/proxy_only.py, ms
    // per call; pairing off 0.71 - 0.72 at every depth): without the threshold 2 / 3 / 4 calls in flight cost 0.98 / 0.86 / 0.74 - a caller that
    // keeps few calls in flight is latency-coupled to each of them; with HOLD_MIN = 5 and the release rule: <= 5 in flight as without pairing,
    // 6: 0.58, 7: 0.55, 8: 0.52, 11: 0.49 (profiles/r06_adaptive_hold_sweep.txt).
    static constexpr int HOLD_MIN = 5;
    unsigned epass = 0;  // recogniser passes queued so far (activation set / stream of the next one)
    long paired_passes = 0, single_passes = 0;
    // is a recogniser pass queued earlier still running (or waiting to run)?  Two event queries, ~ 1 us each
    bool recogniser_busy() const {
        for (int k = 0; k < 2; ++k)
            if (emb->busy[k] && hipEventQuery(emb->ev_busy[k]) == hipErrorNotReady) return true;
        return false;
    }
    void ensure_async() {
        if (copy_stream) return;
        // The upload stream sits in the stage streams' priority class (its own hardware-queue pool): as a normal-priority stream it is
        // dealt round-robin onto the four queues the CALLER's streams live on, and whenever it lands on the queue of the caller's joining
        // stream the next batch's upload sits behind the pending joins of the batches in flight (measured with RCCL's streams in the
        // process: 4-frame step 0.96 -> 1.69 ms, 32-frame step 3.28 -> 3.45 ms).  copy_prio: see frt_pipeline_create.
        HIPCHK(hipStreamCreateWithPriority(&copy_stream, hipStreamNonBlocking, copy_prio));
        const size_t F = (size_t)F_cap;
        for (AsyncBuf &b : abuf) {
            b.d_frames = arena.alloc<uint8_t>((size_t)max_frames * det->g.frame_h * det->g.frame_w * 3);
            b.d_results = arena.alloc<frt_face_result>(F);
            b.d_embeds = arena.alloc<float>(F * 512);
            b.d_crops = arena.alloc<uint8_t>(F * 112 * 112 * 3);
            HIPCHK(hipEventCreateWithFlags(&b.ev_h2d, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&b.ev_out, hipEventDisableTiming));
        }
    }

    // ---- hipGraph replay.  A step is ~150 dependent launches; eager dispatch costs 3.1 us per dependent kernel on this part,
    //      a graph replay 1.8 us (tools/ubench/launch_gap.hip).  Each call is two graphs - the detector part on det_stream, the
    //      rest on `stream` - so the cross-call overlap of the two streams survives; the fork/join events stay ordinary stream
    //      operations between the graph launches.  A part is keyed by everything baked into its nodes (buffers, batch, slot,
    //      mode, gallery generation); first sighting of a key runs eagerly (lazy one-time setup inside the launchers), the second
    //      is captured, later ones replay.  Off while the profiling hooks record events (frt_profile_enable).
    //      OPT-IN (FRT_PIPELINE_GRAPH=1 / frt_pipeline_set_graph(p, 1)): on the benchmark step the replay measured 4.41 ms against
    //      4.39 ms eager - the launches are queued far enough ahead that the per-dispatch cost hides behind the previous kernel.
    struct GraphKey {
        int part;
        const void *frames, *results, *embeds;
        int n, slot, align;
        unsigned gallery_gen;
        bool operator==(const GraphKey &o) const {
            return part == o.part && frames == o.frames && results == o.results && embeds == o.embeds && n == o.n && slot == o.slot && align == o.align &&
                   gallery_gen == o.gallery_gen;
        }
    };
    struct GraphEntry {
        GraphKey key;
        int seen = 0;
        hipGraphExec_t exec = nullptr;
        unsigned long used = 0;  // tick of the last sighting (least-recently-used eviction)
    };
    std::vector<GraphEntry> graphs;
    bool use_graphs = false;
    // a steady pipelined workload cycles through NSLOT keys of stage 0, up to 2 * NSLOT (slot, activation set) pairs of stage 1 and NSLOT of
    // stage 2: the cache holds them all (a smaller one evicted every key before it recurred - nothing was ever replayed)
    static constexpr size_t GRAPH_CAP = 4 * NSLOT + 8;
    unsigned long graph_tick = 0;
    long graphs_captured = 0, graphs_replayed = 0;
    void drop_graphs() {
        for (GraphEntry &e : graphs)
            if (e.exec) (void)hipGraphExecDestroy(e.exec);
        graphs.clear();
    }
    template <typename Body>
    void run_part(const GraphKey &key, hipStream_t st, Body body) {
        if (!use_graphs || g_prof_kind != 0 || serial_call) return body(st);
        GraphEntry *e = nullptr;
        for (GraphEntry &g : graphs)
            if (g.key == key) e = &g;
        if (!e) {
            if (graphs.size() >= GRAPH_CAP) {  // callers that never repeat their buffers: bounded memory - the least recently seen key goes
                size_t lru = 0;
                for (size_t i = 1; i < graphs.size(); ++i)
                    if (graphs[i].used < graphs[lru].used) lru = i;
                if (graphs[lru].exec) (void)hipGraphExecDestroy(graphs[lru].exec);
                graphs.erase(graphs.begin() + (long)lru);
            }
            graphs.push_back(GraphEntry{key, 0, nullptr, 0});
            e = &graphs.back();
        }
        e->used = ++graph_tick;
        if (e->exec) {
            HIPCHK(hipGraphLaunch(e->exec, st));
            ++graphs_replayed;
            return;
        }
        if (e->seen++ == 0) return body(st);
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        try {
            body(st);
        } catch (...) {
            (void)hipStreamEndCapture(st, &g);
            if (g) (void)hipGraphDestroy(g);
            throw;
        }
        HIPCHK(hipStreamEndCapture(st, &g));
        const hipError_t ie = hipGraphInstantiate(&e->exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) {
            e->exec = nullptr;
            HIPCHK(ie);
        }
        ++graphs_captured;
        HIPCHK(hipGraphLaunch(e->exec, st));
    }

    // ---- stream-overlap self-check.  The three-stage pipeline only overlaps when its stage streams (and the caller's joining stream)
    //      sit on different hardware queues: ROCm maps streams round-robin onto GPU_MAX_HW_QUEUES (4) queues per priority level, a
    //      queue is in-order, and one extra stream created before the pipeline has been seen to cost 7 % - 2.5x.
    //      Measured, not assumed: one 150 us single-wave spin kernel per stream, started together; `ratio` = elapsed / 150 us is ~1 when
    //      they run side by side and ~n when n streams share a queue.
    std::string warning;      // last self-check verdict ("" = fine); frt_pipeline_check_overlap returns it through frt_last_error
    float overlap_ratio = 0.f;
    float check_streams(const std::vector<hipStream_t> &sts, double us = 150.0) {
        std::vector<hipEvent_t> a(sts.size()), b(sts.size());
        for (size_t i = 0; i < sts.size(); ++i) {
            HIPCHK(hipEventCreate(&a[i]));
            HIPCHK(hipEventCreate(&b[i]));
            HIPCHK(hipStreamSynchronize(sts[i]));
        }
        for (size_t i = 0; i < sts.size(); ++i) launch_spin(5.0, sts[i]);  // first use of the kernel: code load off the clock
        for (size_t i = 0; i < sts.size(); ++i) HIPCHK(hipStreamSynchronize(sts[i]));
        float worst = 0.f;
        for (int rep = 0; rep < 3; ++rep) {  // best of three: a late host thread inflates a run, nothing deflates it
            for (size_t i = 0; i < sts.size(); ++i) {
                HIPCHK(hipEventRecord(a[i], sts[i]));
                launch_spin(us, sts[i]);
                HIPCHK(hipEventRecord(b[i], sts[i]));
            }
            for (size_t i = 0; i < sts.size(); ++i) HIPCHK(hipEventSynchronize(b[i]));
            float span = 0.f;  // first start -> last end
            for (size_t i = 0; i < sts.size(); ++i) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, a[0], b[i]));
                span = std::max(span, ms);
            }
            const float r = span * 1e3f / (float)us;
            worst = rep == 0 ? r : std::min(worst, r);
        }
        for (size_t i = 0; i < sts.size(); ++i) {
            (void)hipEventDestroy(a[i]);
            (void)hipEventDestroy(b[i]);
        }
        return worst;
    }
    // Does a wait that is PENDING on stream `j` hold up work on stream `x`?  That is what sharing a hardware queue means for this
    // pipeline: kernels of two streams multiplexed onto one queue may still run side by side, but a queue is in-order, so the caller's
    // stream - on which every call leaves "wait for the end of my match stage" - blocks whatever stream shares its queue until that
    // call has finished, and consecutive calls serialise.  Test: a 400 us probe kernel on `g`, an event behind it that `j` waits for,
    // then a 20 us probe on `x`: finished long before the gate opens (ratio << 1) or only behind it (>= 1).
    float blocked_by_wait(hipStream_t j, hipStream_t x, hipStream_t g) {
        hipEvent_t e0, gate, xb;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&gate));
        HIPCHK(hipEventCreate(&xb));
        for (hipStream_t st : {j, x, g}) HIPCHK(hipStreamSynchronize(st));
        float worst = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            HIPCHK(hipEventRecord(e0, g));
            launch_spin(400.0, g);
            HIPCHK(hipEventRecord(gate, g));
            HIPCHK(hipStreamWaitEvent(j, gate, 0));
            launch_spin(20.0, x);
            HIPCHK(hipEventRecord(xb, x));
            HIPCHK(hipEventSynchronize(xb));
            HIPCHK(hipStreamSynchronize(j));
            HIPCHK(hipStreamSynchronize(g));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, xb));
            const float r = ms / 0.4f;
            worst = rep == 0 ? r : std::min(worst, r);
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(gate);
        (void)hipEventDestroy(xb);
        return worst;
    }

    void self_check(bool with_caller) {
        std::vector<hipStream_t> sts = {det_stream, emb_stream, emb_stream2};
        std::vector<const char *> names = {"detector", "recogniser", "recogniser-2"};
        if (with_caller) {
            if (stream) {
                sts.push_back(stream);
                names.push_back("caller");
            }
            if (copy_stream) {
                sts.push_back(copy_stream);
                names.push_back("upload");
            }
        }
        overlap_ratio = check_streams({det_stream, emb_stream, emb_stream2});
        warning.clear();
        if (with_caller && stream) {  // the hazard proper: a pending join on the caller's stream must not hold up a pipeline stream
            std::string held;
            struct X {
                hipStream_t st;
                const char *name;
                hipStream_t gate_on;
            } xs[] = {{copy_stream, "upload", emb_stream2}, {det_stream, "detector", emb_stream2}, {emb_stream, "recogniser", emb_stream2},
                      {emb_stream2, "recogniser-2", emb_stream}};
            for (const X &x : xs) {
                if (!x.st) continue;
                const float r = blocked_by_wait(stream, x.st, x.gate_on);
                if (r > 0.8f) held += std::string(held.empty() ? "" : ", ") + x.name;
            }
            if (!held.empty()) {
                char buf[768];
                snprintf(buf, sizeof(buf),
                         "frt_pipeline: a wait pending on the caller's stream holds up the pipeline's %s stream(s) - they share a hardware queue, so "
                         "every call's final join blocks the next call and consecutive batches serialise (measured: 4-frame step 0.78 -> 1.65 ms).  "
                         "Hand the pipeline another stream (a newly created one lands on another queue) and check again; see INTEGRATION.md "
                         "'Streams and hardware queues'.",
                         held.c_str());
                warning = buf;
                overlap_ratio = std::max(overlap_ratio, 2.0f);
                if (!getenv("FRT_QUIET")) fprintf(stderr, "[libfrt] warning: %s\n", buf);
                return;
            }
        }
        if (overlap_ratio > 1.5f) {
            // which two?  pairwise probes (only on the failing path: 3 x 150 us per pair)
            std::string pairs;
            for (size_t i = 0; i < sts.size(); ++i)
                for (size_t j = i + 1; j < sts.size(); ++j)
                    if (check_streams({sts[i], sts[j]}) > 1.5f) pairs += std::string(pairs.empty() ? "" : ", ") + names[i] + " + " + names[j];
            char buf[768];
            snprintf(buf, sizeof(buf),
                     "frt_pipeline: %zu streams of the stage pipeline do not run side by side (150 us probe kernels took %.2fx as long together as "
                     "alone; sharing a hardware queue: %s): consecutive batches will not overlap.  Create the pipeline - and hand it the "
                     "caller's stream - before the process's other HIP streams (RCCL, codec, copy streams) are created or first used, keep "
                     "GPU_MAX_HW_QUEUES at its default 4, see INTEGRATION.md 'Streams and hardware queues'.",
                     sts.size(), overlap_ratio, pairs.empty() ? "?" : pairs.c_str());
            warning = buf;
            if (!getenv("FRT_QUIET")) fprintf(stderr, "[libfrt] warning: %s\n", buf);
        }
    }

    void ensure_stream() {
        if (!stream) {
            if (!own_stream) HIPCHK(hipStreamCreate(&own_stream));
            stream = own_stream;
        }
    }
    void run(const uint8_t *frames_dev, int n, frt_face_result *results_dev, float *embeds_dev) {
        ensure_stream();
        hipStream_t s = stream;
        const DetGeom &g = det->g;
        const int F = n * max_faces;
        // Three-stage software pipeline over consecutive calls (stage-profiling mode and overlap off: everything serially on `s`):
        //   D  detector of call b+1          (fp32 / split-fp16 MFMA + latency-bound stencils)
        //   E  crop + recogniser of call b   (fp16 MFMA / LDS bound)
        //   M  match + pack of call b-1      (HBM bound: streams the 1 GB fp16 shadow gallery)
        // The caller's stream only JOINS: it waits for M of this call, so everything the caller enqueues after the call sees the
        // results, exactly as if the call had run on that stream.  Boxes, embeddings and validity flags live in NSLOT slots.
        // Profiled calls (frt_profile_enable 1 or 2) run serially on `s`: HIP events around a launch only measure the kernel when
        // no other stream competes for the dispatch (with four streams in flight the bracketed time was 2.7x the kernel time).
        const bool pipe3 = overlap && g_prof_kind == 0 && !serial_call;
        // pairing: this call's later stages wait for the next call - or run together with the waiting call's
        const int gcap = std::min({group < 0 ? (int)MAXG : group, F_cap / F, emb->max_batch / F});  // calls of this size one pass can take
        const bool pairable = pipe3 && gcap >= 2 && (group > 0 || (group < 0 && (host_req.nsub || adaptive_dev)));
        if (npend && !(pairable && pend[0].n == n)) flush_pending();
        const unsigned call = seq++;
        const int slot = (int)(call % NSLOT);
        if (pipe3 && serial_pending) {  // a serial call used the shared detector / recogniser buffers on `s`: order the stages behind it
            HIPCHK(hipStreamWaitEvent(det_stream, ev_serial, 0));
            HIPCHK(hipStreamWaitEvent(emb_stream, ev_serial, 0));
            HIPCHK(hipStreamWaitEvent(emb_stream2, ev_serial, 0));
            serial_pending = false;
        }
        hipStream_t ds = pipe3 ? det_stream : s;
        if (pipe3 && call >= (unsigned)NSLOT) {
            // slot buffers are free again once M of the call NSLOT back is done.  NB the frames must be valid when the call is made:
            // making D wait for prior work on `s` would serialise the stages.
            HIPCHK(hipStreamWaitEvent(ds, ev_done[slot], 0));
        }
        if (ev_frames) {  // frt_pipeline_submit: the frames arrive on the copy stream
            HIPCHK(hipStreamWaitEvent(ds, ev_frames, 0));  // (crop + recogniser follow the detector through ev_det[slot])
            ev_frames = nullptr;
        }
        if (ev_ready) {  // frt_pipeline_run_dev_after: the caller's producer (upload / decode / resize on any stream) signals this event
            HIPCHK(hipStreamWaitEvent(ds, ev_ready, 0));
            ev_ready = nullptr;
        }
        if (input_sync && pipe3) {  // safe mode: everything queued on the caller's stream before this call happens-before the stages
            HIPCHK(hipEventRecord(ev_input, s));
            HIPCHK(hipStreamWaitEvent(ds, ev_input, 0));
        }
        CallRec cur = host_req;  // (staging set + host destinations when the call came through frt_pipeline_submit)
        host_req = CallRec{};
        cur.on = true;
        cur.call = call;
        cur.slot = slot;
        cur.n = n;
        cur.frames = frames_dev;
        cur.results = results_dev;
        cur.embeds = embeds_dev;
        cur.crops = crops_req;  // (one call only)
        crops_req = nullptr;
        const int akey = (align ? 1 : 0) | (cur.crops ? 2 : 0);
        run_part(GraphKey{0, frames_dev, nullptr, nullptr, n, slot, akey, 0u}, ds, [&](hipStream_t st) {
#ifdef FRT_TUNING
            // timing build: FRT_PIPE_ABLATE bit 0 = no detector network after the first calls (post-processing re-reads the old head outputs),
            // bit 1 = no recogniser network, bit 2 = no match: what each stage costs the pipelined step (profiles/r04/r04s_stage_ablation.txt)
            static const int pipe_abl = getenv("FRT_PIPE_ABLATE") ? atoi(getenv("FRT_PIPE_ABLATE")) : 0;
            if (!(pipe_abl & 1) || call < 8u)
#endif
            det->forward_frames(frames_dev, n, (size_t)g.frame_w * 3, (size_t)g.frame_w * g.frame_h * 3, st);
            det->postprocess(n, st, slot_boxes[slot], slot_nout[slot], slot_landmarks[slot]);  // straight into this call's slot
        });
        HIPCHK(hipEventRecord(det->ev_busy, ds));  // object-level detector calls wait for this (frt_detector::wait_idle)
        det->busy = true;
        if (pipe3) HIPCHK(hipEventRecord(ev_det[slot], ds));
        if (pairable) {
            // adaptive: hold this call back only while earlier recogniser passes are still in flight; fixed groups: always
            const bool hold = group > 0 || npend > 0 || (recogniser_busy() && (!cur.nsub || tickets_running() >= HOLD_MIN));
            if (hold) {
                pend[npend++] = cur;  // nothing else is queued for this call now (the caller's stream joins with the last partner's call)
                if (npend == gcap || (group < 0 && npend >= 2 && !recogniser_busy())) flush_pending();
                return;
            }
        }
        later_stages(&cur, 1, pipe3);
    }

    // the waiting calls' crop + recogniser + match: the group is complete, or the missing partners never came
    void flush_pending() {
        if (!npend) return;
        CallRec grp[MAXG];
        const int n = npend;
        for (int i = 0; i < n; ++i) grp[i] = pend[i];
        npend = 0;
        try {
            later_stages(grp, n, true);  // (a call is only ever deferred in the three-stream mode: its detector stage sits on det_stream)
        } catch (...) {
            // the held calls are lost; their tickets must not be answered from a staging set's STALE "results have left" event: mark them
            // failed and re-arm the event behind whatever did get queued, so that frt_pipeline_wait returns - with the error
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < grp[i].nsub; ++j) {
                    grp[i].sub[j].ab->failed = true;
                    (void)hipEventRecord(grp[i].sub[j].ab->ev_out, stream);
                }
            throw;
        }
    }
    bool is_pending(long ticket) const {
        for (int i = 0; i < npend; ++i)
            for (int j = 0; j < pend[i].nsub; ++j)
                if (pend[i].sub[j].ticket == ticket) return true;
        return false;
    }

    // E and M of one call, or of up to MAXG consecutive calls as ONE recogniser pass and ONE match call (pairing)
    void later_stages(const CallRec *c, int nc, bool pipe3) {
        hipStream_t s = stream;
        const DetGeom &g = det->g;
        int Fc[MAXG] = {}, Ftot = 0;
        for (int i = 0; i < nc; ++i) {
            Fc[i] = c[i].n * max_faces;
            Ftot += Fc[i];
        }
        (nc >= 2 ? paired_passes : single_passes) += 1;
        const int eset = (pipe3 && dual_embed && Ftot <= emb->max_batch) ? (int)(epass++ & 1u) : 0;  // activation set / stream of this recogniser pass
        hipStream_t es = pipe3 ? (eset ? emb_stream2 : emb_stream) : s;
        // match + pack follow the recogniser pass on ITS stream (they overlap the other set's pass and the next detector pass): a stream
        // of their own measured 0.6 % slower and is one more stream competing for the four hardware queues
        hipStream_t ms = es;
        float *chw = eset ? d_chw2 : d_chw;
        // embeddings and validity flags of the pass: the first call's slot (the calls of a group fit one slot together: run() checked)
        float *emb_slot = slot_embeds[c[0].slot];
        int *valid = slot_valid[c[0].slot];
        for (int i = 0; i < nc; ++i) {
            if (pipe3 && c[i].call >= (unsigned)NSLOT) HIPCHK(hipStreamWaitEvent(es, ev_done[c[i].slot], 0));
            if (pipe3) HIPCHK(hipStreamWaitEvent(es, ev_det[c[i].slot], 0));
        }
        const bool have_gallery = mat && mat->N > 0;
        const unsigned gen = mat ? mat->generation : 0u;
        const int akey = (align ? 1 : 0) | (c[0].crops ? 2 : 0);
        // (a pass on activation set k follows the previous pass on the same set: ordered by its stream)
        auto stage_e = [&](hipStream_t st) {
            int f_off = 0;
            for (int i = 0; i < nc; ++i) {
                const int sl = c[i].slot;
                if (align) {
                    ProfScope ps(2, "align_faces", (double)Fc[i] * 112 * 112 * 3, st);
                    launch_align_faces(c[i].frames, g.frame_h, g.frame_w, (size_t)g.frame_w * 3, (size_t)g.frame_w * g.frame_h * 3, slot_landmarks[sl],
                                       slot_nout[sl], max_faces, Fc[i], 0, c[i].crops, chw + (size_t)f_off * 3 * 112 * 112, valid + f_off, st);
                } else {
                    ProfScope ps(2, "crop_faces", (double)Fc[i] * 112 * 112 * 3, st);
                    launch_crop_faces(c[i].frames, g.frame_h, g.frame_w, (size_t)g.frame_w * 3, (size_t)g.frame_w * g.frame_h * 3, slot_boxes[sl],
                                      slot_nout[sl], max_faces, Fc[i], 0, 112, 112, c[i].crops, chw + (size_t)f_off * 3 * 112 * 112, valid + f_off, st);
                }
                f_off += Fc[i];
            }
#ifdef FRT_TUNING
            static const int pipe_abl = getenv("FRT_PIPE_ABLATE") ? atoi(getenv("FRT_PIPE_ABLATE")) : 0;
            if (!(pipe_abl & 2) || c[0].call < 8u)
#endif
            for (int f0 = 0; f0 < Ftot; f0 += emb->max_batch) {
                const int nf = std::min(emb->max_batch, Ftot - f0);
                emb->forward_set(eset, chw + (size_t)f0 * 3 * 112 * 112, nf, valid + f0, emb_slot + (size_t)f0 * 512, st);
            }
        };
        // (the fp32 pass is never captured: its single activation set is handed from pass to pass through the host-tracked event f32.done,
        //  which must be a real record on every pass - and a graph captured in one precision must not be replayed in the other)
        if (nc == 1 && !emb->fp32_mode) run_part(GraphKey{1, c[0].frames, nullptr, nullptr, c[0].n, c[0].slot, akey, (unsigned)eset}, es, stage_e);
        else stage_e(es);
        HIPCHK(hipEventRecord(emb->ev_busy[eset], es));
        emb->busy[eset] = true;
        if (pipe3) {
            HIPCHK(hipEventRecord(ev_emb[c[0].slot], es));
            HIPCHK(hipStreamWaitEvent(ms, ev_emb[c[0].slot], 0));
        }
        // consecutive calls' match stages sit on DIFFERENT streams (their recogniser passes') but share the matcher's scratch and this
        // pipeline's d_idx / d_sim: each one starts behind the previous one's end (they rarely meet: 0.3 ms every 3.3 ms, half a
        // period apart - which is exactly why an unordered pair showed up as one failing equality test in several hundred)
        // (the serial branch too: an object-level frt_matcher_top1_dev / topk_dev on another stream shares d_partial / the pair lists with this stage)
        if (mat && mat->busy) HIPCHK(hipStreamWaitEvent(ms, mat->ev_busy, 0));
        auto stage_m = [&](hipStream_t st) {
#ifdef FRT_TUNING
            static const int pipe_abl = getenv("FRT_PIPE_ABLATE") ? atoi(getenv("FRT_PIPE_ABLATE")) : 0;
            if (!(pipe_abl & 4) || c[0].call < 8u)
#endif
            if (have_gallery) mat->top1_dev(emb_slot, Ftot, d_idx, d_sim, st);
            int f_off = 0;
            for (int i = 0; i < nc; ++i) {
                const int sl = c[i].slot;
                {
                    ProfScope ps(2, "pack_results", (double)Fc[i], st);
                    // one launch per ticket of the call (merged submits): a ticket's records count ITS frames from zero
                    const int np = c[i].nsub > 1 ? c[i].nsub : 1;
                    for (int j = 0, fr = 0; j < np; ++j) {
                        const int nf = c[i].nsub > 1 ? c[i].sub[j].n : c[i].n, o = fr * max_faces;
                        launch_pack_results(slot_boxes[sl] + o, slot_nout[sl] + fr, valid + f_off + o, have_gallery ? d_idx + f_off + o : nullptr,
                                            have_gallery ? d_sim + f_off + o : nullptr, max_faces, nf * max_faces, c[i].results + o, st);
                        fr += nf;
                    }
                }
                if (c[i].embeds)
                    HIPCHK(hipMemcpyAsync(c[i].embeds, emb_slot + (size_t)f_off * 512, sizeof(float) * 512 * Fc[i], hipMemcpyDeviceToDevice, st));
                f_off += Fc[i];
            }
        };
        if (nc == 1) run_part(GraphKey{2, nullptr, c[0].results, c[0].embeds, c[0].n, c[0].slot, akey, gen}, ms, stage_m);
        else stage_m(ms);
        if (mat) {
            HIPCHK(hipEventRecord(mat->ev_busy, ms));
            mat->busy = true;
        }
        if (pipe3) {
            for (int i = 0; i < nc; ++i) HIPCHK(hipEventRecord(ev_done[c[i].slot], ms));
            HIPCHK(hipStreamWaitEvent(s, ev_done[c[0].slot], 0));  // the caller's stream joins here
        } else if (overlap) {
            HIPCHK(hipEventRecord(ev_serial, s));
            serial_pending = true;
        }
        // calls that came through frt_pipeline_submit: their downloads follow the join
        for (int i = 0; i < nc; ++i) {
            size_t o = 0;  // face slots in front of this ticket inside the call's device blocks (c[i].results / .embeds / .crops)
            for (int j = 0; j < c[i].nsub; ++j) {
                const Sub &t = c[i].sub[j];
                const size_t nf = (size_t)t.n * max_faces;
                HIPCHK(hipMemcpyAsync(t.h_results, c[i].results + o, sizeof(frt_face_result) * nf, hipMemcpyDeviceToHost, s));
                if (t.h_embeds) HIPCHK(hipMemcpyAsync(t.h_embeds, c[i].embeds + o * 512, sizeof(float) * 512 * nf, hipMemcpyDeviceToHost, s));
                if (t.h_crops) HIPCHK(hipMemcpyAsync(t.h_crops, c[i].crops + o * 112 * 112 * 3, nf * 112 * 112 * 3, hipMemcpyDeviceToHost, s));
                o += nf;
            }
            // "results have left" only behind the LAST download of the call: the first ticket's staging set carries every ticket's data
            for (int j = 0; j < c[i].nsub; ++j) HIPCHK(hipEventRecord(c[i].sub[j].ab->ev_out, s));
        }
    }
};

