"""RetinaFace-mnet0.25 HIP forward vs the fp32 oracle (and the reference-module goldens), then findFace end to end."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

LOC_TOL = 2e-4   # fp32 network, BN folded on the host, different summation order: observed ~1e-5
CONF_TOL = 2e-5


@pytest.mark.parametrize("tag,hw", [("96x160", (96, 160)), ("288x320", (288, 320)), ("640", (640, 640))])
def test_network_matches_oracle_and_reference_goldens(frt, synth, blobs, tag, hw):
    from oracle import nets
    h, w = hw
    path, sd = blobs("det")
    det = frt.RetinaFace(path, w, h, (3, h, w), 2, 4)
    fr = synth.make_frames(2, h, w)
    x = np.ascontiguousarray((fr.astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))
    loc, conf = det.doInference(x)
    oloc, oconf = nets.retinaface_forward(sd, x)
    assert loc.shape == oloc.shape and conf.shape == oconf.shape
    assert np.abs(loc - oloc).max() < LOC_TOL, np.abs(loc - oloc).max()
    assert np.abs(conf - oconf).max() < CONF_TOL, np.abs(conf - oconf).max()
    g = np.load(os.path.join(GOLDEN, "retinaface_mnet.npz"))
    step = int(g["step_" + tag])
    assert np.abs(loc[:, ::step] - g["loc_" + tag]).max() < LOC_TOL
    assert np.abs(conf[:, ::step] - g["conf_" + tag]).max() < CONF_TOL
    det.close()


def test_non_multiple_of_32_input(frt, synth, blobs):
    from oracle import nets
    h, w = 100, 172  # ceil feature maps 13x22, 7x11, 4x6; nearest upsample 4->7->13 is not an exact 2x
    path, sd = blobs("det")
    det = frt.RetinaFace(path, w, h, (3, h, w), 1, 4)
    x = np.ascontiguousarray((synth.make_frames(1, h, w).astype(np.float32) - 110).transpose(0, 3, 1, 2))
    loc, conf = det.doInference(x)
    oloc, oconf = nets.retinaface_forward(sd, x)
    assert loc.shape == oloc.shape
    assert np.abs(loc - oloc).max() < LOC_TOL and np.abs(conf - oconf).max() < CONF_TOL
    det.close()


@pytest.mark.parametrize("geom", [(640, 640, 640, 640), (320, 288, 640, 480)])
def test_find_face_boxes_match_oracle_pipeline(frt, orc, synth, blobs, geom):
    from oracle import nets
    in_w, in_h, fw, fh = geom
    path, sd = blobs("det")
    det = frt.RetinaFace(path, fw, fh, (3, in_h, in_w), 4, 4, 0.4, 0.6)
    frames = synth.make_frames(4, fh, fw)
    got = det.findFaceBatch(frames)
    exact = total = 0
    for f in range(4):
        x = orc.det_preprocess(frames[f], in_h, in_w)
        oloc, oconf = nets.retinaface_forward(sd, x[None])
        want = orc.postprocess(oloc[0], oconf[0], in_w, in_h, fw, fh, 0.4, 0.6, 4)
        assert len(got[f]) == len(want) == 4
        single = det.findFace(frames[f])
        for k in ("x1", "y1", "x2", "y2", "score"):
            assert np.array_equal(single[k], got[f][k])  # batch of 1 == batched path
        for k in ("x1", "y1", "x2", "y2"):
            d = np.abs(got[f][k] - want[k])
            assert d.max() <= 1, (f, k, got[f], want)  # int truncation of a float that differs in the last bits
            exact += int((d == 0).sum())
            total += d.size
        assert np.abs(got[f]["score"] - want["score"]).max() < CONF_TOL
    assert exact >= total - 2, (exact, total)
    det.close()


@pytest.mark.parametrize("hw", [(101, 173), (96, 160), (320, 320)])
def test_fused_u8_first_conv_equals_the_staged_path(frt, synth, blobs, hw):
    """findFace on a frame whose size equals the network input runs the fused u8 first conv (dword loads in the interior,
    byte loads on an odd right edge, 4-pixel conv_dw rows when W % 4 == 0); preprocess -> doInference -> postprocessing runs the
    separate preprocess kernel and the generic first conv.  Same arithmetic, same summation order: identical boxes and scores."""
    h, w = hw
    path, _ = blobs("det")
    det = frt.RetinaFace(path, w, h, (3, h, w), 1, 8, 0.4, 0.02)
    for seed in range(3):
        frame = synth.make_frames(1, h, w, start=seed)[0]
        fused = det.findFace(frame)
        loc, conf = det.doInference(det.preprocess(frame)[None])
        staged = det.postprocessing(loc[0], conf[0])
        assert len(fused) == len(staged) > 0
        for k in ("x1", "y1", "x2", "y2", "score"):
            assert np.array_equal(fused[k], staged[k]), (hw, seed, k)
    det.close()


def test_batch_of_32_equals_frame_by_frame(frt, synth, blobs):
    """Size-independent property at the benchmark batch: every frame of a 32-frame call gets exactly the boxes (and head outputs)
    it gets alone - the persistent tile walks / batch indexing of the matrix-core kernels must not leak across frames."""
    path, _ = blobs("det")
    h = w = 640
    det32 = frt.RetinaFace(path, w, h, (3, h, w), 32, 4)
    det1 = frt.RetinaFace(path, w, h, (3, h, w), 1, 4)
    frames = synth.make_frames(32, h, w)
    batch = det32.findFaceBatch(frames)
    for i in (0, 1, 13, 30, 31):
        alone = det1.findFace(frames[i])
        assert np.array_equal(batch[i], alone), i
    x = np.ascontiguousarray((frames[:32].astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))
    loc32, conf32 = det32.doInference(x)
    for i in (0, 17, 31):
        loc1, conf1 = det1.doInference(x[i:i + 1])
        assert np.array_equal(loc32[i], loc1[0]) and np.array_equal(conf32[i], conf1[0]), i
    det32.close()
    det1.close()


@pytest.mark.parametrize("geo,frames,max_differing", [("640x640->640x640", 256, 0), ("640x480->320x288", 256, 1)])
def test_box_census(frt, blobs, geo, frames, max_differing):
    """The box-flip census (tools/box_census.py) as a test: findFace on the GPU against the fp32 oracle +
    oracle/postproc.c on 256 synthetic frames per geometry, 1 024 boxes / 4 096 coordinates each.  Expected (profiles/r02/r02_box_census.json,
    512 frames): no differing coordinate at 640x640, ONE at 320x288 (an `int` truncation of a float that differs in its last bits,
    src/retinaface.cpp:171-187) - never a different box count, never more than one detector-input pixel."""
    import importlib.util
    import os

    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("box_census", os.path.join(ROOT, "tools", "box_census.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)
    dpath, dsd = blobs("det")
    r = bc.census(frt, dsd, dpath, bc.GEOMETRIES[geo], frames)
    assert r["boxes"] >= frames * 3 and r["frames_with_different_box_count"] == 0, r
    assert r["coordinates_differing"] <= max_differing and r["max_abs_diff_px"] <= 2, r
    assert r["max_abs_score_diff"] < 1e-5, r


def test_batch_size_classes_of_the_conv_dw_kernels_agree_bit_for_bit(frt, synth, blobs):
    """Round 4: at 640x640 the 64->64 (80x80), 128->128 (40x40) and 256->256 (20x20) conv_dw blocks run on dwpw_wave_kernel from 3 / 12 / 12 frames per
    call upwards and on dwpw_mfma_kernel below (kernels_det_wave.hip: same arithmetic operation for operation).  16 frames in one call must give
    the head outputs of the same frames in calls of 1, 2 and 4, bit for bit - and the oracle's within the usual tolerance."""
    from oracle import nets
    path, sd = blobs("det")
    fr = np.concatenate([synth.make_frames(4, 640, 640), synth.make_frames(4, 640, 640)[:, ::-1], synth.make_frames(4, 640, 640)[:, :, ::-1],
                         synth.make_frames(4, 640, 640)[:, ::-1, ::-1]])
    x = np.ascontiguousarray((fr.astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))
    big = frt.RetinaFace(path, 640, 640, (3, 640, 640), 16, 4)
    loc16, conf16 = big.doInference(x)
    big.close()
    small = frt.RetinaFace(path, 640, 640, (3, 640, 640), 2, 4)
    for i in range(0, 16, 2):
        loc2, conf2 = small.doInference(x[i:i + 2])
        assert np.array_equal(loc2, loc16[i:i + 2]) and np.array_equal(conf2, conf16[i:i + 2]), i
    small.close()
    # the 32 -> 32 block and the 16-channel SSH convs have their own kernels for 1, 2 - 5 and more frames per call (kernels_det.hip: output-channel
    # tiles over grid.y, weights through LDS): calls of 1 and of 4 frames as well
    for nb in (1, 4):
        det = frt.RetinaFace(path, 640, 640, (3, 640, 640), nb, 4)
        for i in range(0, 8, nb):
            locn, confn = det.doInference(x[i:i + nb])
            assert np.array_equal(locn, loc16[i:i + nb]) and np.array_equal(confn, conf16[i:i + nb]), (nb, i)
        det.close()
    oloc, oconf = nets.retinaface_forward(sd, x[12:14])
    assert np.abs(loc16[12:14] - oloc).max() < LOC_TOL and np.abs(conf16[12:14] - oconf).max() < CONF_TOL


@pytest.mark.parametrize("hw", [(256, 320), (192, 192), (320, 448)])
def test_fused_stem_equals_the_staged_path_on_other_identity_geometries(frt, synth, blobs, hw):
    """Round 4: from two frames per call whose size equals the network input, the first conv and the first two conv_dw blocks run as ONE kernel
    (kernels_det_stem.hip: 8x8 output tiles, an interior and a ring launch).  Same arithmetic as the three kernels: four frames through findFaceBatch
    must give exactly what preprocess -> doInference -> postprocessing gives frame by frame (separate preprocess kernel, generic first conv,
    stand-alone conv_dw kernels) - on non-square maps and on maps whose ring is most of the tiles."""
    h, w = hw
    path, _ = blobs("det")
    det = frt.RetinaFace(path, w, h, (3, h, w), 4, 8, 0.4, 0.02)
    frames = synth.make_frames(4, h, w)
    fused = det.findFaceBatch(frames)
    for i in range(4):
        loc, conf = det.doInference(det.preprocess(frames[i])[None])
        staged = det.postprocessing(loc[0], conf[0])
        assert len(fused[i]) == len(staged) > 0
        for k in ("x1", "y1", "x2", "y2", "score"):
            assert np.array_equal(fused[i][k], staged[k]), (hw, i, k)
    det.close()
