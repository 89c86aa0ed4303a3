"""Optional 5-point alignment mode on the GPU vs oracle/align.py (raw landmark head pinned by the reference-module golden;
decode / similarity / warp are "parity unpinned" - the reference has no such path)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _x(frames):
    return np.ascontiguousarray((frames.astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))


@pytest.mark.parametrize("tag,hw", [("96x160", (96, 160)), ("288x320", (288, 320))])
def test_landmark_head_matches_oracle_and_golden(frt, synth, blobs, tag, hw):
    from oracle import nets
    h, w = hw
    path, sd = blobs("det_ldm")
    det = frt.RetinaFace(path, w, h, (3, h, w), 2, 4)
    assert det.hasLandmarks
    x = _x(synth.make_frames(2, h, w))
    loc, conf, ldm = det.doInferenceLandmarks(x)
    oloc, oconf, oldm = nets.retinaface_forward(sd, x)
    assert np.abs(ldm - oldm).max() < 2e-4 and np.abs(loc - oloc).max() < 2e-4 and np.abs(conf - oconf).max() < 2e-5
    g = np.load(os.path.join(GOLDEN, "retinaface_mnet_ldm.npz"))
    assert np.abs(ldm[:, ::3] - g["ldm_" + tag]).max() < 2e-4
    # the trimmed blob gives bit-identical loc/conf (the extra head is a pure addition) and refuses the landmark calls
    tpath, _ = blobs("det")
    tdet = frt.RetinaFace(tpath, w, h, (3, h, w), 2, 4)
    assert not tdet.hasLandmarks
    tloc, tconf = tdet.doInference(x)
    assert np.array_equal(tloc, loc) and np.array_equal(tconf, conf)
    with pytest.raises(frt.FrtError):
        tdet.doInferenceLandmarks(x)
    with pytest.raises(frt.FrtError):
        tdet.findFaceLandmarks(np.zeros((h, w, 3), np.uint8))
    det.close()
    tdet.close()


@pytest.mark.parametrize("geom", [(640, 640, 640, 640), (320, 288, 640, 480)])
def test_find_face_landmarks(frt, orc, synth, blobs, geom):
    from oracle import align
    in_w, in_h, fw, fh = geom
    path, sd = blobs("det_ldm")
    det = frt.RetinaFace(path, fw, fh, (3, in_h, in_w), 1, 4, 0.4, 0.6)
    checked = 0
    for seed in range(4):
        frame = synth.make_frames(1, fh, fw, start=seed)[0]
        boxes, lms = det.findFaceLandmarks(frame)
        assert np.array_equal(boxes, det.findFace(frame))  # boxes are those of the reference path
        if not len(boxes):
            continue
        # oracle side: GPU head outputs -> reference postprocessing (with anchor indices) -> landmark decode
        loc, conf, ldm = det.doInferenceLandmarks(det.preprocess(frame))
        oboxes, cand, cidx = orc.postprocess(loc[0], conf[0], in_w, in_h, fw, fh, 0.4, 0.6, 4, return_candidates=True)
        assert np.array_equal(oboxes, boxes)
        kept = []
        for b in oboxes:
            hit = [int(cidx[i]) for i in range(len(cand)) if cand[i] == b]
            kept.append(min(hit))
        olm = align.decode_landmarks(ldm[0], np.array(kept), in_h, in_w, fh, fw)
        assert lms.shape == olm.shape and np.abs(lms - olm).max() < 1e-3, np.abs(lms - olm).max()
        checked += len(boxes)
    assert checked > 0
    det.close()


def test_align_faces_matches_oracle(frt, synth):
    from oracle import align
    frame = synth.make_frames(1, 480, 640)[0]
    t = align.ARC_TEMPLATE.astype(np.float64).reshape(5, 2)
    r = np.random.default_rng(5)
    lms = []
    for theta, s, shift in ((0.0, 1.0, (40, 30)), (0.4, 1.8, (500, -20)), (-0.9, 0.6, (300, 200)), (2.8, 3.1, (320, 240)), (0.1, 0.25, (10, 400))):
        A = s * np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
        lms.append(t @ A.T + shift + r.normal(0, 1.0, (5, 2)))
    lms = np.array(lms, np.float32)
    crops = frt.alignFaces(frame, lms)
    ocrops, valid = align.align_faces(frame, lms)
    assert valid.all()
    d = np.abs(crops.astype(int) - ocrops.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())  # same float32 operation order: essentially bit-equal
    # integer-shifted template = plain copy
    assert np.array_equal(frt.alignFaces(frame, (t + (40, 30)).astype(np.float32)[None])[0], frame[30:142, 40:152])
    # degenerate landmarks fail loudly
    with pytest.raises(frt.FrtError):
        frt.alignFaces(frame, np.full((1, 5, 2), 9.0, np.float32))
    assert len(frt.alignFaces(frame, np.zeros((0, 5, 2), np.float32))) == 0


def test_forward_aligned_and_pipeline_align_mode(frt, orc, synth, blobs):
    from oracle import align, match, nets
    dpath, dsd = blobs("det_ldm")
    rpath, rsd = blobs("ir")
    B, K, H, W = 2, 4, 640, 640
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    frames = synth.make_frames(B, H, W)
    boxes, lms, oemb = [], [], []
    for f in range(B):
        b, lm = det.findFaceLandmarks(frames[f])
        assert len(b) == K
        crops, valid = align.align_faces(frames[f], lm)
        assert valid.all()
        e = rec.forwardAligned(frames[f], b, lm)
        assert np.abs(np.stack([c["face"] for c in rec.croppedFaces]).astype(int) - crops.astype(int)).max() <= 1
        oe = nets.arcface_forward(rsd, orc.face_normalize(crops))
        assert ((e * oe).sum(1) > 1 - 1e-4).all(), (e * oe).sum(1)
        # aligned crops differ from the reference's bbox crops (otherwise this test would prove nothing)
        assert ((rec.forward(frames[f], b) * e).sum(1) < 0.999).all()
        boxes.append(b); lms.append(lm); oemb.append(e)
    oemb = np.concatenate(oemb)
    gal = synth.make_gallery(4096)
    slots = np.arange(len(oemb)) * 311 + 5
    gal[slots] = oemb
    rec.setGallery(gal)
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    res_crop, emb_crop = pipe.run(frames)
    pipe.set_align(True)
    res, emb = pipe.run(frames)
    pipe.sync()
    assert res["valid"].all() and np.array_equal(res["x1"], res_crop["x1"])
    # 8 faces in one pass vs 4 + 4: other kernels for some layers (kernels_arc_small.hip), fp16 roundings of activations flip
    assert np.abs(emb - oemb).max() < 1e-3 and (emb * oemb).sum(1).min() > 1 - 1e-5
    assert np.array_equal(res["match_idx"], slots) and (res["match_sim"] > 0.999).all()
    assert not np.array_equal(res_crop["match_idx"], slots)
    oi, _ = match.top1(emb, gal)
    assert np.array_equal(oi, res["match_idx"])
    pipe.set_align(False)
    res3, emb3 = pipe.run(frames)
    assert np.array_equal(res3["match_idx"], res_crop["match_idx"]) and np.array_equal(emb3, emb_crop)
    pipe.close()
    # a pipeline over the trimmed detector refuses the mode
    tpath, _ = blobs("det")
    tdet = frt.RetinaFace(tpath, W, H, (3, H, W), B, K)
    tp = frt.Pipeline(tdet, rec, B)
    with pytest.raises(frt.FrtError):
        tp.set_align(True)
    tp.close(); tdet.close(); det.close(); rec.close()
