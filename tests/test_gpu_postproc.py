"""Device decode + NMS vs oracle/postproc.c on identical raw head outputs: exact integer equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GEOMS = [(640, 640, 640, 640), (320, 288, 640, 480), (640, 640, 1920, 1080), (640, 640, 480, 640), (160, 96, 200, 100)]


def heads(A, seed, frac=0.03):
    r = np.random.Generator(np.random.PCG64(seed))
    loc = r.normal(0, 1.0, (A, 4)).astype(np.float32)
    c1 = r.random(A).astype(np.float32)
    c1[r.random(A) > frac] *= 0.5
    c1[:8] = np.float32(0.6)      # exactly at threshold -> dropped (strict >)
    c1[8:12] = np.float32(0.75)   # ties -> lower anchor index first
    return loc, np.stack([1 - c1, c1], 1).astype(np.float32)


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("kmax", [4, 64])
def test_postprocess_exact_integer_parity(frt, orc, blobs, geom, kmax):
    in_w, in_h, fw, fh = geom
    path, _ = blobs("det")
    det = frt.RetinaFace(path, fw, fh, (3, in_h, in_w), 1, kmax, 0.4, 0.6)
    assert det.numAnchors == orc.anchors(in_w, in_h).shape[0]
    for seed in range(3):
        loc, conf = heads(det.numAnchors, 1000 * seed + in_w)
        got = det.postprocessing(loc, conf)
        want = orc.postprocess(loc, conf, in_w, in_h, fw, fh, 0.4, 0.6, kmax)
        assert len(got) == len(want) and len(got) >= 4
        for k in ("x1", "y1", "x2", "y2", "score"):
            assert np.array_equal(got[k], want[k]), (k, got, want)
    det.close()


def test_postprocess_edge_cases(frt, orc, blobs):
    path, _ = blobs("det")
    det = frt.RetinaFace(path, 640, 640, (3, 640, 640), 1, 4, 0.4, 0.6)
    A = det.numAnchors
    loc = np.zeros((A, 4), np.float32)
    conf = np.zeros((A, 2), np.float32)
    assert len(det.postprocessing(loc, conf)) == 0                       # nothing above threshold
    conf[:, 1] = 0.9                                                       # every anchor a candidate (n = 16 800)
    loc[:] = np.random.Generator(np.random.PCG64(5)).normal(0, 0.5, (A, 4))
    conf[:, 1] += np.linspace(0, 0.05, A, dtype=np.float32)
    got = det.postprocessing(loc, conf)
    want = orc.postprocess(loc, conf, 640, 640, 640, 640, 0.4, 0.6, 4)
    assert len(got) == 4
    for k in ("x1", "y1", "x2", "y2", "score"):
        assert np.array_equal(got[k], want[k])
    conf[:, 1] = np.float32("nan")                                         # NaN scores never pass '>'
    assert len(det.postprocessing(loc, conf)) == 0
    det.close()
