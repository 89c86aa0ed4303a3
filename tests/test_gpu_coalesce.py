"""Request coalescing (frt_coalescer_*, include/frt/coalesce.h): concurrent one-frame requests - the reference's request shape,
src/app.cpp:293-352, served .multithreaded(), :367 - share pipeline batches; every caller gets exactly its own frame's answer."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd")
pytestmark = pytest.mark.gpu


def _objects(frt, blobs, frames_cap, H=320, W=320, K=4):
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    det = frt.RetinaFace(dpath, W, H, (3, H, W), frames_cap, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, (3, 112, 112), 512, frames_cap * K, K, 0.65)
    return det, rec, dpath, rpath


def test_concurrent_requests_get_their_own_answers(frt, synth, blobs):
    H = W = 320
    T, PER = 8, 6
    det, rec, _, _ = _objects(frt, blobs, 8, H, W)
    frames = synth.make_frames(T * PER, H, W, start=500)
    # the lone answers: one frame per call through the object-level entry points
    want = []
    planted = []
    for f in frames:
        boxes = det.findFace(f)
        emb = rec.forward(f, boxes) if len(boxes) else np.zeros((0, 512), np.float32)
        want.append((boxes, emb))
        planted.extend(emb)
    assert sum(len(b) for b, _ in want) >= T * PER
    gal = synth.make_gallery(3000)
    planted = np.array(planted, np.float32)
    gal[1000:1000 + len(planted)] = planted
    rec.setGallery(gal)
    rec.initMatMul()
    co = frt.Coalescer(det, rec, 8, window_us=300)
    got = [None] * len(frames)
    errs = []

    def worker(t):
        try:
            for i in range(t * PER, (t + 1) * PER):
                got[i] = co.infer(frames[i], want_crops=True)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    batches, carried = co.stats()
    assert carried == len(frames) and batches < carried, (batches, carried)   # requests really travelled together
    row = 1000
    for i, (boxes, emb) in enumerate(want):
        res, e, crops = got[i]
        assert len(res) == len(boxes)
        for k in ("x1", "y1", "x2", "y2", "score"):
            assert np.array_equal(res[k], boxes[k]), (i, k)          # the detector's boxes do not depend on the batch
        assert (res["valid"] == 1).all() and (res["frame"] == 0).all()
        if len(boxes):
            cos = (e * emb).sum(1)
            assert cos.min() > 1 - 1e-5, (i, cos)                     # recogniser kernels are chosen by batch size: same embedding up to fp16 roundings
            assert np.array_equal(res["match_idx"], np.arange(row, row + len(boxes))), (i, res["match_idx"])
            assert np.abs(res["match_sim"] - 1).max() < 1e-4
            assert np.array_equal(crops, frt.getCroppedFaces(frames[i], boxes)), i   # what CroppedFace.face holds (src/arcface.cpp:3-17)
        row += len(boxes)
    # a lone request on an idle coalescer is answered too (after at most the window)
    res, e = co.infer(frames[0])
    assert len(res) == len(want[0][0])
    # wrong frame size: rejected, not queued
    with pytest.raises(frt.FrtError):
        co.infer(np.zeros((H // 2, W, 3), np.uint8))
    co.close()
    det.close()
    rec.close()


def _build(tmp, src, name):
    exe = os.path.join(tmp, name)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", src),
                           "-o", exe, os.path.join(PKG, "libfrt.so"), "-lpthread", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shells_coalesced_equal_plain_per_call(frt, synth, blobs, tmp_path):
    """tests/cpp/coalesce_test.cpp: 6 threads on ONE detector + ONE recogniser, src/app.cpp:304-310 verbatim; plain, then after
    recognizer.coalesceWith(detector): identical boxes / names / crops / input tensors per call, embeddings and similarities within 1e-5."""
    H = W = 320
    det, rec, dpath, rpath = _objects(frt, blobs, 1, H, W)
    frames = synth.make_frames(5, H, W, start=900)
    embs = [rec.forward(f, det.findFace(f)) for f in frames]
    gal = synth.make_gallery(2000)
    planted = np.concatenate([e for e in embs if len(e)])
    gal[300:300 + len(planted)] = planted
    det.close()
    rec.close()
    frames.tofile(str(tmp_path / "frames.bin"))
    gal.tofile(str(tmp_path / "gal.bin"))
    exe = _build(str(tmp_path), "coalesce_test.cpp", "coalesce_test")
    out = subprocess.run([exe, dpath, rpath, str(tmp_path / "frames.bin"), "5", str(H), str(W), str(tmp_path / "gal.bin"), "2000", "6", "8"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "coalesce ok" in out.stdout, out.stdout + out.stderr


def test_env_opt_in_links_the_two_shells(frt, synth, blobs, tmp_path):
    """FRT_COALESCE=<frames>:<window> in the environment: the unmodified call sequence (dropin_bench, shared objects constructed with the
    reference's maxBatchSize 1 / 4) coalesces - no coalesceWith() call in the program."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import json

    import dropin_bench as db
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    frames = synth.make_frames(3, 640, 640)
    frames.tofile(str(tmp_path / "frames.bin"))
    exe = db.build(str(tmp_path))
    env = dict(os.environ, FRT_COALESCE="8:200")
    out = subprocess.run([exe, dpath, rpath, str(tmp_path / "frames.bin"), "3", "640", "640", "50000", "6", "5", "0", "shared", "0"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(next(l for l in out.stdout.splitlines() if l.startswith("{")))
    assert r["fastpath_mismatches"] == 0 and r["objects"] == "shared"
    assert r["coalesced_frames"] >= 3 * 6 * 5 and r["coalesced_batches"] < r["coalesced_frames"], r
    for k in ("featureMatching_getOutputs", "featureMatching_getOutputs_no_matrix", "matchTop1"):
        assert r[k]["frames"] == 30 and r[k]["faces"] >= 30, r
