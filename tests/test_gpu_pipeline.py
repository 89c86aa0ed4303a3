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
    assert np.array_equal(b0["x1"], res["x1"][:K]) and np.abs(e0 - emb[:K]).max() < 1e-5
    oi, osim = match.top1(emb, gal)
    assert np.array_equal(oi, res["match_idx"]) and np.abs(osim - res["match_sim"]).max() < 1e-5
    # fewer frames than capacity, and an empty gallery
    res2, _ = pipe.run(frames[:2])
    assert len(res2) == 2 * K and np.array_equal(res2["match_idx"], res["match_idx"][:2 * K])
    pipe.close()
    det.close()
    rec.close()
