"""Batched device-resident pipeline (detect -> crop -> embed -> match) vs the oracle run stage by stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pipeline_end_to_end(frt, orc, synth, blobs):
    from oracle import match, nets
    dpath, dsd = blobs("det")
    rpath, rsd = blobs("ir")
    B, K, H, W = 4, 4, 640, 640
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    frames = synth.make_frames(B, H, W)
    # oracle, stage by stage
    oboxes, oemb = [], []
    for f in range(B):
        loc, conf = nets.retinaface_forward(dsd, orc.det_preprocess(frames[f], H, W)[None])
        b = orc.postprocess(loc[0], conf[0], W, H, W, H, 0.4, 0.6, K)
        oboxes.append(b)
        oemb.append(nets.arcface_forward(rsd, orc.face_normalize(orc.crop_faces(frames[f], b))))
    oemb = np.concatenate(oemb)
    # enrol the oracle's embeddings inside a 10k gallery (config 2 of BASELINE.json: 10k x 512)
    gal = synth.make_gallery(10000)
    slots = np.arange(len(oemb)) * 613 + 7
    gal[slots] = oemb
    rec.initKnownEmbeds(len(gal))
    rec.addEmbeddings([str(i) for i in range(len(gal))], gal)
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    res, emb = pipe.run(frames)
    assert len(res) == B * K and res["valid"].all()
    k = 0
    for f in range(B):
        for j in range(K):
            r, ob = res[f * K + j], oboxes[f][j]
            assert r["frame"] == f
            same_box = all(r[c] == ob[c] for c in ("x1", "y1", "x2", "y2"))
            assert all(abs(int(r[c]) - int(ob[c])) <= 1 for c in ("x1", "y1", "x2", "y2"))
            cos = float((emb[f * K + j] * oemb[k]).sum())
            if same_box:
                assert cos > 1 - 1e-4, cos
            assert r["match_idx"] == slots[k], (f, j, r, slots[k])  # identical top-1 IDs
            assert r["match_sim"] > 0.99
            k += 1
    # device results == the un-fused API sequence findFace -> forward -> matchTop1
    b0 = det.findFace(frames[0])
    e0 = rec.forward(frames[0], b0)
    # (4 faces alone take the small-batch kernels for most layers, the pipeline's whole batch the strip kernels: fp16 roundings flip)
    assert np.array_equal(b0["x1"], res["x1"][:K]) and np.abs(e0 - emb[:K]).max() < 1e-3 and (e0 * emb[:K]).sum(1).min() > 1 - 1e-5
    oi, osim = match.top1(emb, gal)
    assert np.array_equal(oi, res["match_idx"]) and np.abs(osim - res["match_sim"]).max() < 1e-5
    # fewer frames than capacity, and an empty gallery
    res2, _ = pipe.run(frames[:2])
    assert len(res2) == 2 * K and np.array_equal(res2["match_idx"], res["match_idx"][:2 * K])
    pipe.close()
    det.close()
    rec.close()


def test_two_stream_overlap_keeps_batches_apart(frt, synth, blobs):
    """Back-to-back asynchronous calls (detector of call b+1 overlaps embed/match of call b) == one call at a time."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 2, 4, 320, 320
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(3000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    batches = [synth.make_frames(B, H, W, start=10 * i) for i in range(5)]
    want = [pipe.run(b)[0].copy() for b in batches]          # serial reference (each call synchronises)
    assert sum(int(w["valid"].sum()) for w in want) > 0
    d_frames = [torch.from_numpy(b).cuda() for b in batches]
    d_res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in batches]
    pipe.set_stream(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for rep in range(2):                                       # second round re-uses both box slots
        for f, r in zip(d_frames, d_res):
            pipe.run_dev(f.data_ptr(), B, r.data_ptr(), None)
        torch.cuda.synchronize()
        for r, w in zip(d_res, want):
            got = np.frombuffer(r.cpu().numpy().tobytes(), frt.RESULT_DTYPE)
            for k in ("x1", "y1", "x2", "y2", "frame", "match_idx", "valid"):
                assert np.array_equal(got[k], w[k]), (rep, k)
            assert np.abs(got["match_sim"] - w["match_sim"]).max() < 1e-6
    pipe.set_stream(None)
    pipe.close()
    det.close()
    rec.close()


def test_async_host_boundary_matches_synchronous_calls(frt, synth, blobs):
    """frt_pipeline_submit / frt_pipeline_wait with more batches than staging sets == frt_pipeline_run one call at a time."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 2, 4, 320, 320
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(3000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    n_batches = 11                                             # > 2 rounds of the 4 staging sets, partial last batch below
    batches = [synth.make_frames(B, H, W, start=7 * i) for i in range(n_batches)]
    batches[-1] = batches[-1][:1]
    want = [tuple(a.copy() for a in pipe.run(b)) for b in batches]
    pinned = [torch.from_numpy(b).pin_memory() for b in batches]
    res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in batches]
    emb = [torch.zeros(B * K, 512).pin_memory() for _ in batches]
    tickets = []
    for i in range(n_batches):
        r = res[i].numpy().view(frt.RESULT_DTYPE)
        tickets.append(pipe.submit(pinned[i].numpy(), r, emb[i].numpy() if i % 2 == 0 else None))
    assert tickets == list(range(tickets[0], tickets[0] + n_batches))
    for i in reversed(range(n_batches)):                       # waiting out of order is allowed
        pipe.wait(tickets[i])
    for i in range(n_batches):
        n = len(batches[i]) * K
        got = res[i].numpy().view(frt.RESULT_DTYPE)[:n]
        w_res, w_emb = want[i]
        for k in ("x1", "y1", "x2", "y2", "frame", "match_idx", "valid"):
            assert np.array_equal(got[k], w_res[k]), (i, k)
        assert np.abs(got["match_sim"] - w_res["match_sim"]).max() < 1e-6
        if i % 2 == 0:
            assert np.abs(emb[i].numpy()[:n] - w_emb).max() < 1e-6
    with pytest.raises(frt.FrtError):
        pipe.wait(tickets[-1] + 1)
    pipe.close()
    det.close()
    rec.close()


def test_gallery_reload_between_pipelined_calls(frt, synth, blobs):
    """/reload while calls are in flight (app.cpp:354-365 re-runs initKnownEmbeds + addEmbedding + initMatMul): calls submitted
    before the reload answer from the old gallery, calls after it from the new one (different size, so the scratch moves)."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 2, 4, 320, 320
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    g_old, g_new = synth.make_gallery(3000), synth.make_gallery(70000)[::-1].copy()
    batches = [synth.make_frames(B, H, W, start=5 * i) for i in range(4)]
    pipe = frt.Pipeline(det, rec, B)
    want = {}
    for tag, gal in (("old", g_old), ("new", g_new)):
        rec.setGallery(gal)
        rec.initMatMul()
        want[tag] = [pipe.run(b)[0].copy() for b in batches]
    assert any(not np.array_equal(a["match_idx"], b["match_idx"]) for a, b in zip(want["old"], want["new"]))
    d_frames = [torch.from_numpy(b).cuda() for b in batches]
    d_res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(8)]
    pipe.set_stream(torch.cuda.current_stream().cuda_stream)
    rec.setGallery(g_old)
    rec.initMatMul()
    for i in range(4):
        pipe.run_dev(d_frames[i].data_ptr(), B, d_res[i].data_ptr(), None)
    rec.setGallery(g_new)      # no explicit synchronisation by the caller
    rec.initMatMul()
    for i in range(4):
        pipe.run_dev(d_frames[i].data_ptr(), B, d_res[4 + i].data_ptr(), None)
    torch.cuda.synchronize()
    for i in range(8):
        got = np.frombuffer(d_res[i].cpu().numpy().tobytes(), frt.RESULT_DTYPE)
        w = want["old" if i < 4 else "new"][i % 4]
        for k in ("x1", "y1", "x2", "y2", "match_idx", "valid"):
            assert np.array_equal(got[k], w[k]), (i, k)
        assert np.abs(got["match_sim"] - w["match_sim"]).max() < 1e-6
    pipe.set_stream(None)
    pipe.close()
    det.close()
    rec.close()


def test_graph_replay_matches_eager(frt, synth, blobs):
    """Opt-in hipGraph replay: same results as eager launches, also after the frame contents / gallery change."""
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 2, 4, 640, 640
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    gal = synth.make_gallery(2048)
    rec.setGallery(gal)
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    fa, fb = synth.make_frames(B, H, W), synth.make_frames(B, H, W, start=7)
    ra, ea = pipe.run(fa)
    rb, eb = pipe.run(fb)
    pipe.set_graph(True)
    for i in range(7):  # both box slots: eager sighting, capture, then replays
        r, e = pipe.run(fa if i % 3 else fb)
        want_r, want_e = (ra, ea) if i % 3 else (rb, eb)
        assert np.array_equal(r, want_r) and np.array_equal(e, want_e), i
    # a new gallery invalidates the captured match part
    gal2 = synth.make_gallery(4096, seed=11)
    rec.setGallery(gal2)
    rec.initMatMul()
    pipe.set_graph(False)
    rc, _ = pipe.run(fa)
    pipe.set_graph(True)
    for i in range(5):
        r, _ = pipe.run(fa)
        assert np.array_equal(r, rc), i
    assert not np.array_equal(rc["match_idx"], ra["match_idx"])
    # the pipelined paths (run_dev / submit: stage streams, ten box slots, two activation sets): a key recurs every ten calls - first round eager,
    # second captured, later ones replayed; the cache must hold all 30 keys of the three stages (it used to drop everything at 16)
    import torch
    d_f = [torch.from_numpy(fa).cuda(), torch.from_numpy(fb).cuda()]
    d_r = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(2)]
    st = torch.cuda.Stream()
    pipe.set_stream(st.cuda_stream)
    pipe.set_graph(False)
    want = []
    for k in range(2):
        pipe.run_dev(d_f[k].data_ptr(), B, d_r[k].data_ptr(), None)
        pipe.sync()
        st.synchronize()
        want.append(d_r[k].cpu().numpy().copy())
    assert np.array_equal(want[0].view(frt.RESULT_DTYPE), rc)
    pipe.set_graph(True)
    c0, r0 = pipe.graph_stats()
    for i in range(50):
        d_r[i & 1].zero_()
        torch.cuda.synchronize()
        pipe.run_dev(d_f[i & 1].data_ptr(), B, d_r[i & 1].data_ptr(), None)
        if i % 10 == 9 or i >= 40:
            pipe.sync()
            st.synchronize()
            assert np.array_equal(d_r[i & 1].cpu().numpy(), want[i & 1]), i
    pipe.sync()
    c1, r1 = pipe.graph_stats()
    assert c1 - c0 >= 30 and r1 - r0 >= 60, (c1 - c0, r1 - r0)   # 3 stages x 10 slots captured in calls 10-19, replayed from call 20 on
    pipe.close()
    det.close()
    rec.close()


def test_repeated_runs_are_bit_identical_at_the_benchmark_size(frt, synth, blobs):
    """32 frames x K = 4 (the bench.py step), 12 back-to-back runs with the two-stream overlap on: every run must return the same
    bytes.  The persistent / double-buffered kernels (conv64, split 3x3, conv_dw) would show hand-over races here as flaky bits."""
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 32, 4, 640, 640
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(50000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    frames = synth.make_frames(B, H, W)
    r0, e0 = pipe.run(frames)
    assert r0["valid"].sum() >= B * K // 2
    for i in range(12):
        r, e = pipe.run(frames)
        assert np.array_equal(r, r0) and np.array_equal(e, e0), i
    pipe.close()
    det.close()
    rec.close()


def test_ir_se_batches_in_flight_equal_one_at_a_time(frt, synth, blobs):
    """IR-SE-50 at the benchmark size with three batches in flight: the SE tail runs inside conv2's epilogue, where the workgroups of
    a face hand their channel sums over through device-scope stores and a flag - while a second recogniser pass, the detector and the
    match compete for the same CUs.  Every batch must return the bytes the one-call-at-a-time run returned, and 128 faces in one pass
    must embed like the same faces in passes of 8 (other strip shapes, the stand-alone SE kernels for some units)."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir_se")
    B, K, H, W = 32, 4, 640, 640
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(50000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    batches = [synth.make_frames(B, H, W, start=40 * i) for i in range(2)]
    want = [tuple(a.copy() for a in pipe.run(b)) for b in batches]
    n_sub = 9
    pinned = [torch.from_numpy(b).pin_memory() for b in batches]
    res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(n_sub)]
    emb = [torch.zeros(B * K, 512).pin_memory() for _ in range(n_sub)]
    tickets = [pipe.submit(pinned[i & 1].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()) for i in range(3)]
    for i in range(3, n_sub):  # three in flight from here on
        pipe.wait(tickets[i - 3])
        tickets.append(pipe.submit(pinned[i & 1].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()))
    for t in tickets[-3:]:
        pipe.wait(t)
    for i in range(n_sub):
        w_res, w_emb = want[i & 1]
        assert np.array_equal(res[i].numpy().view(frt.RESULT_DTYPE), w_res), i
        assert np.array_equal(emb[i].numpy(), w_emb), i
    pipe.close()
    det.close()
    big = rec
    small = frt.ArcFaceIR50(rpath, maxBatchSize=8)
    x = np.random.default_rng(5).standard_normal((128, 3, 112, 112)).astype(np.float32) * 0.5
    e = big.doInference(x)
    for f0 in (0, 56, 120):
        es = small.doInference(x[f0:f0 + 8])
        # (not bit-identical: other strip shapes add the pooled sums in another order, and an fp16 rounding flip behind a gate is 1e-6 in cosine)
        assert (e[f0:f0 + 8] * es).sum(1).min() > 1 - 1e-5 and np.abs(e[f0:f0 + 8] - es).max() < 5e-4, f0
    big.close()
    small.close()


@pytest.mark.timeout(300)  # (unordered match stages corrupt the candidate list and the re-rank kernel then spins for minutes)
def test_match_stages_of_consecutive_calls_are_ordered(frt, synth, blobs):
    """Consecutive calls run their match + pack behind their recogniser pass on two different streams but share the matcher's scratch:
    with one frame per call and a large gallery the match is as long as everything else of a call, so two unordered match stages
    would overlap all the time.  40 calls, three in flight, two alternating frames: every call must return what it returns alone."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 1, 4, 320, 320
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(400000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    batches = [synth.make_frames(B, H, W, start=11 * i) for i in range(2)]
    want = [tuple(a.copy() for a in pipe.run(b)) for b in batches]
    assert not np.array_equal(want[0][0]["match_idx"], want[1][0]["match_idx"])  # the two frames must not match the same rows
    n_sub = 40
    pinned = [torch.from_numpy(b).pin_memory() for b in batches]
    res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(n_sub)]
    tickets = []
    for i in range(n_sub):
        if i >= 3:
            pipe.wait(tickets[i - 3])
        tickets.append(pipe.submit(pinned[i & 1].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), None))
    for t in tickets[-3:]:
        pipe.wait(t)
    for i in range(n_sub):
        assert np.array_equal(res[i].numpy().view(frt.RESULT_DTYPE), want[i & 1][0]), i
    pipe.close()
    det.close()
    rec.close()


def test_pipeline_run_from_several_threads(frt, synth, blobs):
    """frt_pipeline_run is the entry point a multithreaded server (src/app.cpp:367) would call concurrently.  Every call
    takes its own staging set, so four threads hammering it must each get exactly the single-threaded answer for their frames."""
    import threading
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 2, 4, 320, 320
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(3000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    batches = [synth.make_frames(B, H, W, start=3 * i) for i in range(4)]
    want = [tuple(a.copy() for a in pipe.run(b)) for b in batches]
    errors = []

    def worker(i):
        try:
            for _ in range(25):
                r, e = pipe.run(batches[i])
                if not (np.array_equal(r, want[i][0]) and np.array_equal(e, want[i][1])):
                    errors.append(i)
                    return
        except Exception as ex:  # noqa: BLE001
            errors.append((i, repr(ex)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    pipe.close()
    det.close()
    rec.close()


def test_object_level_calls_wait_for_pipeline_stages_in_flight(frt, synth, blobs):
    """after frt_pipeline_run_dev returns, its stages still run on the pipeline's streams and share the detector's
    candidate buffers / the recogniser's activations / the matcher's scratch with the object-level entry points.  Calling those
    immediately (no synchronisation by the caller) must give the quiet-device answers, and the pipelined results must survive."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 8, 4, 640, 640
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    gal = synth.make_gallery(40000)
    rec.setGallery(gal)
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    frames = synth.make_frames(B, H, W)
    other = synth.make_frames(1, H, W, start=77)[0]
    want_res, _ = pipe.run(frames)
    want_boxes = det.findFace(other)
    want_emb = rec.forward(other, want_boxes)
    q = gal[[3, 30003]] + 0.01
    want_idx, want_sim = rec.matmul.top1(q)
    d_frames = torch.from_numpy(frames).cuda()
    d_res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(6)]
    pipe.set_stream(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(6):
        pipe.run_dev(d_frames.data_ptr(), B, d_res[i].data_ptr(), None)
        if i % 2 == 0:
            boxes = det.findFace(other)              # detector stage of call i may still be running
            assert np.array_equal(boxes, want_boxes), i
        if i % 3 == 1:
            emb = rec.forward(other, want_boxes)     # recogniser pass in flight on either activation set
            assert np.array_equal(emb, want_emb), i
        idx, sim = rec.matmul.top1(q)                # match stage in flight on the shared scratch
        assert np.array_equal(idx, want_idx) and np.array_equal(sim, want_sim), i
    torch.cuda.synchronize()
    for i in range(6):
        got = np.frombuffer(d_res[i].cpu().numpy().tobytes(), frt.RESULT_DTYPE)
        assert np.array_equal(got, want_res), i
    pipe.set_stream(None)
    pipe.close()
    det.close()
    rec.close()


def test_run_dev_orders_behind_the_callers_producer(frt, synth, blobs):
    """frt_pipeline_run_dev_after (caller's event) and frt_pipeline_set_input_sync (event on the pipeline stream): frames that are
    still being uploaded when the call is made must be the ones the stages see."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 8, 4, 640, 640
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(3000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    batches = [synth.make_frames(B, H, W, start=11 * i) for i in range(4)]
    want = [pipe.run(b)[0].copy() for b in batches]
    pinned = [torch.from_numpy(b).pin_memory() for b in batches]
    filler = torch.zeros(256 << 20, dtype=torch.uint8).pin_memory()   # a long copy in front of each upload keeps it late
    d_fill = torch.empty_like(filler, device="cuda")
    d_frames = [torch.zeros_like(p, device="cuda") for p in pinned]
    d_res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in batches]
    main = torch.cuda.current_stream()
    pipe.set_stream(main.cuda_stream)
    up = torch.cuda.Stream()
    # (a) the caller's own event on a separate upload stream: calls keep overlapping
    evs = []
    for i in range(4):
        with torch.cuda.stream(up):
            d_fill.copy_(filler, non_blocking=True)
            d_frames[i].copy_(pinned[i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(up)
        evs.append(ev)
        pipe.run_dev(d_frames[i].data_ptr(), B, d_res[i].data_ptr(), None, ready_event=ev.cuda_event)
    torch.cuda.synchronize()
    for i in range(4):
        assert np.array_equal(np.frombuffer(d_res[i].cpu().numpy().tobytes(), frt.RESULT_DTYPE), want[i]), i
    # (b) safe mode: upload on the pipeline stream itself
    pipe.set_input_sync(True)
    for t in d_frames + d_res:
        t.zero_()
    torch.cuda.synchronize()
    for i in range(4):
        d_fill.copy_(filler, non_blocking=True)
        d_frames[i].copy_(pinned[i], non_blocking=True)
        pipe.run_dev(d_frames[i].data_ptr(), B, d_res[i].data_ptr(), None)
    torch.cuda.synchronize()
    for i in range(4):
        assert np.array_equal(np.frombuffer(d_res[i].cpu().numpy().tobytes(), frt.RESULT_DTYPE), want[i]), i
    pipe.set_input_sync(False)
    pipe.set_stream(None)
    pipe.close()
    det.close()
    rec.close()


def test_stream_overlap_self_check_reports_a_clean_pipeline(frt, synth, blobs):
    """frt_pipeline_check_overlap (round 3): on a freshly created pipeline the three stage streams run their probe kernels side by side
    (ratio ~ 1) and a wait pending on the caller's stream holds up none of them - no warning; the call leaves the pipeline usable."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 2, 4, 160, 224
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(2000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    frames = synth.make_frames(B, H, W)
    want, _ = pipe.run(frames)
    st = torch.cuda.Stream()
    pipe.set_stream(st.cuda_stream)
    ratio, warning = pipe.check_overlap()
    assert 0.5 < ratio < 1.5 and warning == "", (ratio, warning)
    got, _ = pipe.run(frames)
    assert np.array_equal(got, want)
    pipe.set_stream(None)
    pipe.close()
    det.close()
    rec.close()


def test_pairing_of_consecutive_calls_changes_nothing_but_the_pass_count(frt, synth, blobs):
    """frt_pipeline_set_pairing: crop + recogniser + match of two consecutive calls as ONE pass.  Four frames x four face slots per call = 16
    faces, 32 per paired pass: the two sizes take different tile shapes of the recogniser kernels, so embeddings agree to fp16 rounding (cosine
    >= 1 - 1e-5), everything else - boxes, validity, the matched rows - is identical to the unpaired pipeline; the submit / wait contract holds
    (waiting on the ticket of a call that is still waiting for a partner flushes it), and so does the device-resident path."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 4, 4, 320, 320

    def same(got, want):  # records: everything exact but the similarity, which follows the embedding
        got, want = np.asarray(got).view(frt.RESULT_DTYPE), np.asarray(want).view(frt.RESULT_DTYPE)
        return all(np.array_equal(got[k], want[k]) for k in ("x1", "y1", "x2", "y2", "frame", "match_idx", "valid")) and \
            float(np.abs(got["match_sim"] - want["match_sim"]).max()) < 2e-4

    def same_emb(got, want):
        n = np.linalg.norm(want, axis=1) > 0.5
        return np.array_equal(got[~n], want[~n]) and float((got[n] * want[n]).sum(1).min(initial=1.0)) > 1 - 1e-5 and float(np.abs(got - want).max()) < 2e-3
    det = frt.RetinaFace(dpath, W, H, (3, H, W), 4 * B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=4 * B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(3000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, 4 * B)                       # room for four calls' face slots: the condition for groups of up to four
    pipe.set_pairing(0)                                        # (the default is the adaptive mode, tested below on its own)
    n_batches = 7                                              # odd: the last call finds no partner
    batches = [synth.make_frames(B, H, W, start=5 * i) for i in range(n_batches)]
    pinned = [torch.from_numpy(b).pin_memory() for b in batches]

    def through_submit(order):
        res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in batches]
        emb = [torch.zeros(B * K, 512).pin_memory() for _ in batches]
        tickets = [pipe.submit(pinned[i].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()) for i in range(n_batches)]
        for i in order:
            pipe.wait(tickets[i])
        return [r.numpy().view(frt.RESULT_DTYPE).copy() for r in res], [e.numpy().copy() for e in emb]

    # every face gets its own gallery row (similarity ~ 1, far above the next row), so that "the same matched rows" does not hang on the last
    # bits of an embedding
    _r0, emb0 = through_submit(range(n_batches))
    gal = synth.make_gallery(3000)
    all_emb = np.concatenate(emb0)
    ok = np.linalg.norm(all_emb, axis=1) > 0.5
    slots = (np.arange(len(all_emb)) * 23 + 5)
    gal[slots[ok]] = all_emb[ok]
    rec.setGallery(gal)
    rec.initMatMul()
    want_res, want_emb = through_submit(range(n_batches))
    assert sum(int(w["valid"].sum()) for w in want_res) > 0
    for i, w in enumerate(want_res):
        v = w["valid"] != 0
        assert np.array_equal(w["match_idx"][v], slots[i * B * K:(i + 1) * B * K][v]) and (w["match_sim"][v] > 0.99).all()
    p0, s0 = pipe.pairing_stats()
    assert p0 == 0 and s0 == 2 * n_batches
    pipe.set_pairing(True)
    for order in (range(n_batches), reversed(range(n_batches))):
        got_res, got_emb = through_submit(list(order))
        for i in range(n_batches):
            assert same(got_res[i], want_res[i]), i
            assert same_emb(got_emb[i], want_emb[i]), i
    p1, s1 = pipe.pairing_stats()
    assert p1 - p0 == 2 * (n_batches // 2) and s1 - s0 == 2       # per round: three pairs and the odd call out
    # a call waited for before its partner arrives runs alone
    r = torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
    t = pipe.submit(pinned[0].numpy(), r.numpy().view(frt.RESULT_DTYPE), None)
    pipe.wait(t)
    assert same(r.numpy().view(frt.RESULT_DTYPE), want_res[0])
    assert pipe.pairing_stats() == (p1, s1 + 1)
    # synchronous calls never wait for a partner; a call with too many frames for a pair runs as always
    res_sync, emb_sync = pipe.run(batches[1])
    assert same(res_sync, want_res[1]) and same_emb(emb_sync, want_emb[1])
    big = np.ascontiguousarray(np.concatenate([batches[2], batches[3], batches[4]]))   # 12 frames: two of these do not fit the 16-frame pipeline
    r_big = np.zeros(3 * B * K, frt.RESULT_DTYPE)
    pb0 = pipe.pairing_stats()
    t = pipe.submit(big, r_big, None)
    assert pipe.pairing_stats() == (pb0[0], pb0[1] + 1)            # queued at the call, nothing deferred
    pipe.wait(t)
    for part, src in enumerate((2, 3, 4)):                        # every third of the 12-frame call against the batch it was built from
        seg = r_big[part * B * K:(part + 1) * B * K].copy()
        assert np.array_equal(seg["frame"], want_res[src]["frame"] + part * B), part
        seg["frame"] -= part * B                                    # (frame indices count within the call)
        assert same(seg, want_res[src]), part
    # groups of four: (0, 1, 2, 3) share a pass, (4, 5, 6) are flushed as three when the last ticket is waited for
    pipe.set_pairing(4)
    pg, sg = pipe.pairing_stats()
    got_res, got_emb = through_submit(range(n_batches))
    for i in range(n_batches):
        assert same(got_res[i], want_res[i]), i
        assert same_emb(got_emb[i], want_emb[i]), i
    assert pipe.pairing_stats() == (pg + 2, sg)
    pipe.set_pairing(2)
    # device-resident path: results are complete after the next call's join or after sync
    d_frames = [torch.from_numpy(b).cuda() for b in batches]
    d_res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in batches]
    d_emb = [torch.zeros(B * K, 512, device="cuda") for _ in batches]
    pipe.set_stream(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pa, sa = pipe.pairing_stats()
    for f, rr, ee in zip(d_frames, d_res, d_emb):
        pipe.run_dev(f.data_ptr(), B, rr.data_ptr(), ee.data_ptr())
    pipe.sync()
    torch.cuda.synchronize()
    for i in range(n_batches):
        assert same(d_res[i].cpu().numpy(), want_res[i]), i
        assert same_emb(d_emb[i].cpu().numpy(), want_emb[i]), i
    pb, sb = pipe.pairing_stats()
    assert (pb - pa, sb - sa) == (n_batches // 2, 1)
    pipe.set_pairing(False)
    pipe.set_stream(None)
    pipe.close()
    det.close()
    rec.close()


def test_adaptive_pairing_holds_calls_only_behind_a_busy_recogniser(frt, synth, blobs):
    """The default mode (frt_pipeline_set_pairing(p, -1)): a submit() call is held back - its frames merged with the next submits' into one call, or
    its recogniser pass shared with the next call's - only while at least five earlier tickets are still running.  (a) a lone caller - submit,
    wait, submit, wait - is never held; (b) calls submitted back to back are, with the unpaired pipeline's boxes / rows and embeddings to fp16
    rounding; (c) run_dev calls are never held in this mode: their
    results are joined on the pipeline stream AT the call, no frt_pipeline_sync needed; (d) -2 extends the mode to run_dev (results at sync)."""
    import torch
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 4, 4, 320, 320

    def same(got, want):
        got, want = np.asarray(got).view(frt.RESULT_DTYPE), np.asarray(want).view(frt.RESULT_DTYPE)
        return all(np.array_equal(got[k], want[k]) for k in ("x1", "y1", "x2", "y2", "frame", "match_idx", "valid")) and \
            float(np.abs(got["match_sim"] - want["match_sim"]).max()) < 2e-4

    def same_emb(got, want):
        n = np.linalg.norm(want, axis=1) > 0.5
        return np.array_equal(got[~n], want[~n]) and float((got[n] * want[n]).sum(1).min(initial=1.0)) > 1 - 1e-5
    det = frt.RetinaFace(dpath, W, H, (3, H, W), 4 * B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=4 * B * K, maxFacesPerScene=K)
    n_batches = 11                                             # (a call is only ever held while at least five tickets are running)
    batches = [synth.make_frames(B, H, W, start=11 * i) for i in range(n_batches)]
    pinned = [torch.from_numpy(b).pin_memory() for b in batches]
    rec.setGallery(synth.make_gallery(4000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, 4 * B)

    def new_out():
        return ([torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in batches], [torch.zeros(B * K, 512).pin_memory() for _ in batches])
    # reference: pairing off, one call at a time; every face then gets its own gallery row (see the fixed-group test above)
    pipe.set_pairing(0)
    res, emb = new_out()
    for i in range(n_batches):
        pipe.wait(pipe.submit(pinned[i].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()))
    gal = synth.make_gallery(4000)
    all_emb = np.concatenate([e.numpy() for e in emb])
    ok = np.linalg.norm(all_emb, axis=1) > 0.5
    gal[(np.arange(len(all_emb)) * 19 + 3)[ok]] = all_emb[ok]
    rec.setGallery(gal)
    rec.initMatMul()
    res, emb = new_out()
    for i in range(n_batches):
        pipe.wait(pipe.submit(pinned[i].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()))
    want_res, want_emb = [r.numpy().view(frt.RESULT_DTYPE).copy() for r in res], [e.numpy().copy() for e in emb]
    assert sum(int(w["valid"].sum()) for w in want_res) > 0
    pipe.set_pairing(-1)                                       # the default
    # (a) lone caller: detector and recogniser are idle at every call
    p0, s0 = pipe.pairing_stats()
    assert pipe.merge_stats() == (0, 0)
    res, emb = new_out()
    for i in range(n_batches):
        pipe.wait(pipe.submit(pinned[i].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()))
        assert np.array_equal(res[i].numpy().view(frt.RESULT_DTYPE), want_res[i]) and np.array_equal(emb[i].numpy(), want_emb[i]), i   # the very same pass
    p1, s1 = pipe.pairing_stats()
    assert p1 == p0 and s1 - s0 == n_batches and pipe.merge_stats() == (0, 0), (p1 - p0, s1 - s0)
    # (b) back to back: later calls find the recogniser busy and share passes; waiting in submit order and in reverse
    for order in (list(range(n_batches)), list(reversed(range(n_batches)))):
        res, emb = new_out()
        tickets = [pipe.submit(pinned[i].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()) for i in range(n_batches)]
        for i in order:
            pipe.wait(tickets[i])
        for i in range(n_batches):
            assert same(res[i].numpy(), want_res[i]), i
            assert same_emb(emb[i].numpy(), want_emb[i]), i
    # tickets of one merged call that want different things: every other submit without embeddings
    res, emb = new_out()
    tickets = [pipe.submit(pinned[i].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy() if i % 2 == 0 else None) for i in range(n_batches)]
    for t in tickets:
        pipe.wait(t)
    for i in range(n_batches):
        assert same(res[i].numpy(), want_res[i]), i
        if i % 2 == 0:
            assert same_emb(emb[i].numpy(), want_emb[i]), i
        else:
            assert not emb[i].numpy().any(), i                       # nobody wrote where nothing was asked for
    p2, s2 = pipe.pairing_stats()
    mc, mt = pipe.merge_stats()
    # calls shared passes - as whole calls merged at the host boundary (detector busy at the submit) and / or at the recogniser stage
    assert (p2 - p1) + mc >= 2, (p2 - p1, s2 - s1, mc, mt)
    assert mt >= 2 * mc
    # (c) run_dev is not held: complete after a synchronisation of the pipeline's stream alone
    d_frames = [torch.from_numpy(b).cuda() for b in batches]
    d_res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in batches]
    st = torch.cuda.Stream()
    pipe.set_stream(st.cuda_stream)
    torch.cuda.synchronize()
    pc, sc = pipe.pairing_stats()
    for f, rr in zip(d_frames, d_res):
        pipe.run_dev(f.data_ptr(), B, rr.data_ptr(), None)
    st.synchronize()
    assert pipe.pairing_stats() == (pc, sc + n_batches)
    for i in range(n_batches):
        assert same(d_res[i].cpu().numpy(), want_res[i]), i
    # (d) -2: device-resident calls are held too; complete after frt_pipeline_sync
    pipe.set_pairing(-2)
    for rr in d_res:
        rr.zero_()
    torch.cuda.synchronize()
    pd, sd = pipe.pairing_stats()
    for f, rr in zip(d_frames, d_res):
        pipe.run_dev(f.data_ptr(), B, rr.data_ptr(), None)
    pipe.sync()
    st.synchronize()
    pe, se = pipe.pairing_stats()
    assert pe - pd >= 1, (pe - pd, se - sd)
    for i in range(n_batches):
        assert same(d_res[i].cpu().numpy(), want_res[i]), i
    pipe.close()
    det.close()
    rec.close()
