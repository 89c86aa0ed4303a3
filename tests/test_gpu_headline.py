"""End-to-end parity at the configurations BASELINE.json names, against the oracle run stage by stage (round-1 VERDICT, parity
items 2-4): the headline config (B = 32 frames, K = 4, 640x640, 1M-row gallery) through the full three-slot / dual-activation-set
pipeline with batches in flight, and 1080p frames letterboxed to the 640x640 detector input (the `if` branch of
src/retinaface.cpp:112-116 and :177-181)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COS_TOL = 1e-4  # north_star: embeddings cosine-equal within 1e-4


def oracle_frame(orc, dsd, rsd, frame, in_h, in_w, K):
    from oracle import nets
    fh, fw = frame.shape[:2]
    loc, conf = nets.retinaface_forward(dsd, orc.det_preprocess(frame, in_h, in_w)[None])
    boxes = orc.postprocess(loc[0], conf[0], in_w, in_h, fw, fh, 0.4, 0.6, K)
    emb = nets.arcface_forward(rsd, orc.face_normalize(orc.crop_faces(frame, boxes)))
    return boxes, emb


def check_faces(res, emb, f, K, oboxes, oemb, slots):
    """records of frame f vs the oracle's boxes / embeddings / planted gallery rows"""
    exact = 0
    for j in range(len(oboxes)):
        r, ob = res[f * K + j], oboxes[j]
        assert r["valid"] and r["frame"] == f
        d = [abs(int(r[c]) - int(ob[c])) for c in ("x1", "y1", "x2", "y2")]
        assert max(d) <= 1, (f, j, r, ob)  # census-backed bound, see DESIGN.md section 4 / profiles/r02/r02_box_census.json
        same_box = max(d) == 0
        exact += same_box
        cos = float((emb[f * K + j] * oemb[j]).sum())
        if same_box:
            assert cos > 1 - COS_TOL, (f, j, cos)
        assert r["match_idx"] == slots[j], (f, j, r, slots[j])  # identical top-1 IDs
        assert abs(r["match_sim"] - cos) < 1e-5
    return exact


def test_headline_config_against_the_oracle(frt, orc, synth, blobs):
    import torch
    dpath, dsd = blobs("det")
    rpath, rsd = blobs("ir")
    B, K, H, W, N = 32, 4, 640, 640, 1_000_000
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    batch_a, batch_b = synth.make_frames(B, H, W), synth.make_frames(B, H, W, start=64)
    picked = {"a": (0, 9, 18, 31), "b": (5, 27)}  # frames the oracle also runs on (first / last / middle of the batch)
    gal = synth.make_gallery(N)
    want, slot = {}, 1234
    for tag, frames in (("a", batch_a), ("b", batch_b)):
        for f in picked[tag]:
            boxes, emb = oracle_frame(orc, dsd, rsd, frames[f], H, W, K)
            assert len(boxes) == K
            slots = slot + 7919 * np.arange(K)
            gal[slots] = emb  # plant the oracle's embeddings in the 1M gallery
            want[(tag, f)] = (boxes, emb, slots)
            slot += 40009
    rec.setGallery(gal)
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    # six batches through submit/wait with three in flight: slots, both activation sets and the staging ring all cycle
    order = ["a", "b", "a", "a", "b", "a"]
    pinned = {"a": torch.from_numpy(batch_a).pin_memory(), "b": torch.from_numpy(batch_b).pin_memory()}
    res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in order]
    emb = [torch.zeros(B * K, 512).pin_memory() for _ in order]
    tickets = []
    for i, tag in enumerate(order):
        if len(tickets) >= 3:
            pipe.wait(tickets[i - 3])
        tickets.append(pipe.submit(pinned[tag].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()))
    for t in tickets:
        pipe.wait(t)
    exact = total = 0
    for i, tag in enumerate(order):
        r, e = res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()
        assert r["valid"].all()
        for f in picked[tag]:
            boxes, oemb, slots = want[(tag, f)]
            exact += check_faces(r, e, f, K, boxes, oemb, slots)
            total += K
        if i >= 2:  # the same batch gives the same bytes wherever it sat in the pipeline
            j = order.index(tag)
            assert np.array_equal(r, res[j].numpy().view(frt.RESULT_DTYPE)) and np.array_equal(e, emb[j].numpy()), i
    assert exact >= total - 2, (exact, total)
    pipe.close()
    det.close()
    rec.close()


def test_1080p_frames_end_to_end_against_the_oracle(frt, orc, synth, blobs):
    """BASELINE config 4's frames: 1920x1080 -> scale_h 0.593 > scale_w 0.333 -> w = 640, h = 360, y-offset 140; boxes are mapped back
    to the full-resolution frame and the crops are cut from it."""
    dpath, dsd = blobs("det")
    rpath, rsd = blobs("ir")
    B, K, H, W, FH, FW, N = 4, 4, 640, 640, 1080, 1920, 20000
    det = frt.RetinaFace(dpath, FW, FH, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, FW, FH, maxBatchSize=B * K, maxFacesPerScene=K)
    frames = synth.make_frames(B, FH, FW, start=3)
    gal = synth.make_gallery(N)
    want = []
    for f in range(B):
        boxes, emb = oracle_frame(orc, dsd, rsd, frames[f], H, W, K)
        slots = 100 + 1000 * f + 37 * np.arange(len(boxes))
        gal[slots] = emb
        want.append((boxes, emb, slots))
    assert sum(len(w[0]) for w in want) >= B  # the synthetic detector finds faces in letterboxed frames too
    rec.setGallery(gal)
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    res, emb = pipe.run(frames)
    exact = total = 0
    for f in range(B):
        boxes, oemb, slots = want[f]
        assert int(res["valid"][f * K:(f + 1) * K].sum()) == len(boxes)
        exact += check_faces(res, emb, f, K, boxes, oemb, slots)
        total += len(boxes)
    assert exact >= total - 2, (exact, total)
    # the un-fused call sequence of src/app.cpp:304-310 on the same frames
    b0 = det.findFace(frames[0])
    assert np.array_equal(b0["x1"], res["x1"][:len(b0)]) and np.array_equal(b0["y2"], res["y2"][:len(b0)])
    pipe.close()
    det.close()
    rec.close()


def test_ir_se_through_the_pipeline_against_the_oracle(frt, orc, synth, blobs):
    """IR-SE-50 (the network north_star names) through frt_pipeline_submit / wait with three batches in flight - every slot, both
    activation sets, the fused SE epilogue with its cross-workgroup hand-over under co-running passes - against the fp32 oracle run
    stage by stage, planted gallery rows; then the same batches with the stand-alone SE tail (frt_embedder_set_se_fused(e, 0)): same boxes / ids, embeddings equal to 1e-5 in cosine."""
    import torch
    dpath, dsd = blobs("det")
    rpath, rsd = blobs("ir_se")
    B, K, H, W, N = 8, 4, 640, 640, 200_000
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    batch_a, batch_b = synth.make_frames(B, H, W, start=200), synth.make_frames(B, H, W, start=300)
    picked = {"a": (0, 7), "b": (3,)}
    gal = synth.make_gallery(N)
    want, slot = {}, 777
    for tag, frames in (("a", batch_a), ("b", batch_b)):
        for f in picked[tag]:
            boxes, emb = oracle_frame(orc, dsd, rsd, frames[f], H, W, K)
            assert len(boxes) == K
            slots = slot + 4099 * np.arange(K)
            gal[slots] = emb
            want[(tag, f)] = (boxes, emb, slots)
            slot += 50021
    rec.setGallery(gal)
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    order = ["a", "b", "a", "b", "a", "a", "b", "a"]  # >= 8 batches: every slot / activation set / staging set is reused
    pinned = {"a": torch.from_numpy(batch_a).pin_memory(), "b": torch.from_numpy(batch_b).pin_memory()}

    def run_all():
        res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in order]
        emb = [torch.zeros(B * K, 512).pin_memory() for _ in order]
        tickets = []
        for i, tag in enumerate(order):
            if len(tickets) >= 3:
                pipe.wait(tickets[i - 3])
            tickets.append(pipe.submit(pinned[tag].numpy(), res[i].numpy().view(frt.RESULT_DTYPE), emb[i].numpy()))
        for t in tickets:
            pipe.wait(t)
        return [r.numpy().view(frt.RESULT_DTYPE).copy() for r in res], [e.numpy().copy() for e in emb]

    res, emb = run_all()
    exact = total = 0
    for i, tag in enumerate(order):
        assert res[i]["valid"].all()
        for f in picked[tag]:
            boxes, oemb, slots = want[(tag, f)]
            exact += check_faces(res[i], emb[i], f, K, boxes, oemb, slots)
            total += K
        j = order.index(tag)
        assert np.array_equal(res[i], res[j]) and np.array_equal(emb[i], emb[j]), i  # same bytes wherever the batch sat in the pipeline
    assert exact >= total - 2, (exact, total)
    rec.setSeFused(False)  # stand-alone pool + gate + apply launches: the same arithmetic up to the order in which the pooled sums are added
    res2, emb2 = run_all()
    for i in range(len(order)):
        for c in ("x1", "y1", "x2", "y2", "score", "frame", "match_idx", "valid"):
            assert np.array_equal(res2[i][c], res[i][c]), (i, c)
        assert np.abs(res2[i]["match_sim"] - res[i]["match_sim"]).max() < 5e-3  # |de| <= sqrt(2 * 1e-5) for embeddings within 1e-5 in cosine
        assert ((emb2[i] * emb[i]).sum(1) > 1 - 1e-5).all(), i                   # (measured 2e-6: different summation order of the pooled means)
        assert np.array_equal(res2[i], res2[order.index(order[i])]) and np.array_equal(emb2[i], emb2[order.index(order[i])]), i
    rec.setSeFused(True)
    pipe.close()
    det.close()
    rec.close()


def test_config0_one_jpeg_ir_se_two_face_gallery(frt, orc, synth, blobs):
    """BASELINE configs[0]: a single 640x640 jpg -> RetinaFace-mnet0.25 -> ArcFace IR-SE50 -> 2-face gallery, through the reference's own
    call sequence (src/app.cpp:296-310: imdecode, findFace, forward, featureMatching, getOutputs) vs the oracle on the same bytes."""
    import io
    Image = pytest.importorskip("PIL.Image")
    from oracle import match, nets
    dpath, dsd = blobs("det")
    rpath, rsd = blobs("ir_se")
    K, H, W = 4, 640, 640
    b = io.BytesIO()
    Image.fromarray(synth.make_frame(4242, H, W)[..., ::-1]).save(b, "JPEG", quality=90, subsampling=2)
    jpg = b.getvalue()
    # oracle side: libjpeg(-turbo) decode (what cv::imdecode wraps), then the CPU restatement stage by stage
    oframe = np.ascontiguousarray(np.array(Image.open(io.BytesIO(jpg)))[..., ::-1])
    oboxes, oemb = oracle_frame(orc, dsd, rsd, oframe, H, W, K)
    assert len(oboxes) >= 2
    gallery = np.stack([oemb[1], oemb[0]])  # the "2-face gallery": two enrolled identities
    names = ["bob", "alice"]
    # product side
    codec = frt.JpegCodec(max_images=1, max_width=W, max_height=H)
    frame = codec.decode(jpg)
    assert np.array_equal(frame, oframe)
    det = frt.RetinaFace(dpath, W, H, (3, H, W), 1, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=1, maxFacesPerScene=K)  # rec_maxBatchSize 1: the reference's default (config.json:18)
    rec.initKnownEmbeds(2)
    for n, e in zip(names, gallery):
        rec.addEmbedding(n, e)
    rec.initMatMul()
    boxes = det.findFace(frame)
    assert len(boxes) == len(oboxes)
    for c in ("x1", "y1", "x2", "y2"):
        assert np.array_equal(boxes[c], oboxes[c]), c
    emb = rec.forward(frame, boxes)
    cos = (emb * oemb).sum(1)
    assert cos.min() > 1 - COS_TOL, cos
    sims = rec.featureMatching()                    # the full [F, 2] matrix, as MatMul::calculate returns it
    got_names, got_sims = rec.getOutputs(sims)
    osim = match.similarity(emb, gallery)
    assert sims.shape == (len(boxes), 2) and np.abs(sims - osim).max() < 1e-5
    oi, _ = match.top1(oemb, gallery)
    assert got_names == [names[i] for i in oi]
    assert got_names[0] == "alice" and got_names[1] == "bob" and got_sims[0] > 0.9999 and got_sims[1] > 0.9999
    assert rec.matchTop1()[0] == got_names
    codec.close()
    det.close()
    rec.close()


def test_bench_line_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` (the driver's command, short): ONE JSON line with the contract's fields, the roofline
    object of the dominant kernel (live HIP-event duration, counter traffic from profiles/) and the bounded cpu_baseline leg."""
    import json
    import subprocess
    import sys

    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2"], capture_output=True, text=True,
                         timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "faces/sec" and "workload" in d["config"]
    assert d["value"] > 5000 and abs(d["value"] - d["config"]["faces_per_step"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and 0.05 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 1e7
    assert "conv_patchc_kernel<7" in r["kernel"] and r["launches"] == 27  # the 14x14 body convs on compact strips (the 7 at 28x28 stay on conv_patch_kernel)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "faces/sec" and c["sample"]


@pytest.mark.parametrize("mode", ["ir", "ir_se"])
def test_configs1_batch1_fp32_one_workload(frt, orc, synth, blobs, mode):
    """BASELINE configs[1] as ONE workload in the precision it names: 640x640, batch 1, 10k x 512 gallery, recogniser in fp32 mode
    (setPrecision(True)), through the reference's own call sequence findFace -> forward -> featureMatching -> getOutputs
    (src/app.cpp:304-310) AND through Pipeline.run, against the oracle run stage by stage: boxes by the census rule (every coordinate
    within one pixel, at most one of a frame's differing), 1 - cos <= 1e-6 wherever the crop is the oracle's, identical top-1 rows,
    |delta sim| < 1e-5.  Then back to the default precision: same boxes / rows, embeddings within north_star's 1e-4."""
    dpath, dsd = blobs("det")
    rpath, rsd = blobs(mode)
    K, H, W, N = 4, 640, 640, 10_000
    det = frt.RetinaFace(dpath, W, H, (3, H, W), 1, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=K, maxFacesPerScene=K)
    rec.setPrecision(True)
    frames = synth.make_frames(3, H, W, start=500)
    gal = synth.make_gallery(N)
    want = []
    for f in range(len(frames)):
        boxes, emb = oracle_frame(orc, dsd, rsd, frames[f], H, W, K)
        assert len(boxes) == K
        slots = 11 + 3001 * f + 701 * np.arange(K)
        gal[slots] = emb
        want.append((boxes, emb, slots))
    rec.initKnownEmbeds(N)
    rec.addEmbeddings([str(i) for i in range(N)], gal)
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, 1)
    off_total = 0
    fp32_embeds = []
    for f in range(len(frames)):
        oboxes, oemb, slots = want[f]
        # (1) the reference's call sequence, one frame per call
        boxes = det.findFace(frames[f])
        assert len(boxes) == K
        d = np.stack([np.abs(boxes[c].astype(np.int64) - oboxes[c].astype(np.int64)) for c in ("x1", "y1", "x2", "y2")], 1)
        assert d.max() <= 1 and (d != 0).sum() <= 1, (f, boxes, oboxes)
        off_total += int((d != 0).sum())
        emb = rec.forward(frames[f], boxes)
        sims = rec.featureMatching()
        names, best = rec.getOutputs(sims)
        for j in range(K):
            cos = float((emb[j].astype(np.float64) * oemb[j]).sum())
            if d[j].max() == 0:
                assert cos > 1 - 1e-6, (f, j, 1 - cos)
            assert names[j] == str(slots[j]), (f, j, names[j], slots[j])  # identical top-1 IDs
            assert abs(best[j] - cos) < 1e-5, (f, j, best[j], cos)
        # (2) the same frame through the fused pipeline: same boxes, same rows, the same embeddings to fp32 rounding
        res, pemb = pipe.run(frames[f][None])
        assert res["valid"].all()
        for c in ("x1", "y1", "x2", "y2"):
            assert np.array_equal(res[c], boxes[c]), (f, c)
        assert np.array_equal(res["match_idx"], slots), (f, res["match_idx"], slots)
        assert ((pemb.astype(np.float64) * emb).sum(1) > 1 - 1e-6).all()
        assert np.abs(res["match_sim"] - np.asarray(best, np.float32)).max() < 1e-5
        fp32_embeds.append(pemb.copy())
    assert off_total <= 1, off_total
    # back on the default (fp16 MFMA) path: the mode is a switch, not a rebuild - and it is the less accurate of the two
    rec.setPrecision(False)
    for f in range(len(frames)):
        res, pemb = pipe.run(frames[f][None])
        assert np.array_equal(res["match_idx"], want[f][2])
        assert ((pemb.astype(np.float64) * fp32_embeds[f]).sum(1) > 1 - COS_TOL).all()
        assert not np.array_equal(pemb, fp32_embeds[f])
    pipe.close()
    det.close()
    rec.close()
