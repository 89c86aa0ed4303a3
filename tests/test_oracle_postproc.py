"""oracle/postproc.c: known answers recorded from the reference + an independent NumPy restatement."""
import numpy as np
import pytest


def np_anchors(w, h):
    out = []
    for k, step in enumerate((8.0, 16.0, 32.0)):
        fh, fw = int(np.ceil(np.float32(h) / np.float32(step))), int(np.ceil(np.float32(w) / np.float32(step)))
        for i in range(fh):
            for j in range(fw):
                for ms in ((10, 20), (32, 64), (128, 256))[k]:
                    out.append([np.float32((j + 0.5) * step / w), np.float32((i + 0.5) * step / h), np.float32(ms * 1.0 / w), np.float32(ms * 1.0 / h)])
    return np.array(out, np.float32)


def np_postprocess(loc, conf, in_w, in_h, fw, fh, nms_thr, thr, kmax):
    """Independent restatement of retinaface.cpp:154-271 with explicit float32/float64 typing."""
    f32, f64 = np.float32, np.float64
    anc = np_anchors(in_w, in_h)
    sh, sw = f32(in_h) / f32(fh), f32(in_w) / f32(fw)
    cands = []
    for i in range(len(anc)):
        s = conf[i, 1]
        if not (s > f32(thr)):
            continue
        cx = f32(f64(anc[i, 0]) + f64(loc[i, 0]) * 0.1 * f64(anc[i, 2]))
        cy = f32(f64(anc[i, 1]) + f64(loc[i, 1]) * 0.1 * f64(anc[i, 3]))
        sx = f32(f64(anc[i, 2]) * np.exp(f64(loc[i, 2]) * 0.2))
        sy = f32(f64(anc[i, 3]) * np.exp(f64(loc[i, 3]) * 0.2))
        y1 = int(f32(f32(cx - f32(sx / f32(2))) * f32(in_w)))
        x1 = int(f32(f32(cy - f32(sy / f32(2))) * f32(in_h)))
        y2 = int(f32(f32(cx + f32(sx / f32(2))) * f32(in_w)))
        x2 = int(f32(f32(cy + f32(sy / f32(2))) * f32(in_h)))
        if sh > sw:
            off = f32(f32(f32(in_h) - f32(sw * f32(fh))) / f32(2))
            y1, y2 = int(f32(f32(y1) / sw)), int(f32(f32(y2) / sw))
            x1, x2 = int(f32(f32(f32(x1) - off) / sw)), int(f32(f32(f32(x2) - off) / sw))
        else:
            off = f32(f32(f32(in_w) - f32(sh * f32(fw))) / f32(2))
            y1, y2 = int(f32(f32(f32(y1) - off) / sh)), int(f32(f32(f32(y2) - off) / sh))
            x1, x2 = int(f32(f32(x1) / sh)), int(f32(f32(x2) / sh))
        clip = lambda v, hi: max(min(v, hi), 0)
        cands.append((clip(x1, fh - 1), clip(y1, fw - 1), clip(x2, fh - 1), clip(y2, fw - 1), s, i))
    cands.sort(key=lambda c: (-c[4], c[5]))
    keep = []
    alive = [True] * len(cands)
    for i, a in enumerate(cands):
        if not alive[i]:
            continue
        keep.append(a)
        area_a = f32((a[2] - a[0] + 1) * (a[3] - a[1] + 1))
        for j in range(i + 1, len(cands)):
            if not alive[j]:
                continue
            b = cands[j]
            w = max(f32(0), f32(f32(min(a[2], b[2])) - f32(max(a[0], b[0])) + f32(1)))
            h = max(f32(0), f32(f32(min(a[3], b[3])) - f32(max(a[1], b[1])) + f32(1)))
            inter = f32(w * h)
            area_b = f32((b[2] - b[0] + 1) * (b[3] - b[1] + 1))
            if f32(inter / f32(f32(area_a + area_b) - inter)) >= f32(nms_thr):
                alive[j] = False
    return keep[:kmax]


def test_anchor_known_answers(orc):
    a = orc.anchors(640, 640)
    assert a.shape == (16800, 4)  # A = 16 800 at 640x640
    assert np.array_equal(a[0], np.array([0.00625, 0.00625, 0.015625, 0.015625], np.float32))  # recorded from the reference
    assert np.array_equal(a[-1], np.array([0.975, 0.975, 0.4, 0.4], np.float32))
    assert orc.anchors(320, 288).shape == (3780, 4)  # default config 288x320 (app/config.json:8)
    for w, h in ((640, 640), (320, 288), (100, 70)):
        assert np.array_equal(orc.anchors(w, h), np_anchors(w, h))


GEOMS = [  # (in_w, in_h, frame_w, frame_h)   SURVEY §8(c) G2
    (640, 640, 640, 640),    # identity; scale_h == scale_w -> else branch
    (320, 288, 640, 480),    # config.json default: scale_h 0.6 > scale_w 0.5 -> if branch, y offset 24
    (640, 640, 1920, 1080),  # 1080p: if branch, offset 140
    (640, 640, 480, 640),    # portrait: else branch, x offset 80
]


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("seed", [0, 1])
def test_postprocess_matches_independent_numpy_restatement(orc, geom, seed):
    in_w, in_h, fw, fh = geom
    r = np.random.Generator(np.random.PCG64(100 + seed))
    A = orc.anchors(in_w, in_h).shape[0]
    loc = r.normal(0, 1.0, (A, 4)).astype(np.float32)
    c1 = r.random(A).astype(np.float32)
    c1[r.random(A) < 0.97] *= 0.5  # ~3 % above 0.6
    c1[:8] = np.float32(0.6)       # exactly at the threshold: strict '>' must drop them
    c1[8:12] = np.float32(0.75)    # score ties: lower anchor index first
    conf = np.stack([1 - c1, c1], 1).astype(np.float32)
    got = orc.postprocess(loc, conf, in_w, in_h, fw, fh, 0.4, 0.6, 50)
    want = np_postprocess(loc, conf, in_w, in_h, fw, fh, 0.4, 0.6, 50)
    assert len(got) == len(want) and len(got) > 4
    for g, w in zip(got, want):
        assert (g["x1"], g["y1"], g["x2"], g["y2"]) == w[:4] and g["score"] == w[4]
    assert np.all(np.diff(got["score"]) <= 0)
    assert got["x1"].min() >= 0 and got["x2"].max() <= fh - 1 and got["y2"].max() <= fw - 1


def test_postprocess_cap_is_applied_after_nms_and_empty_input(orc):
    A = 16800
    loc = np.zeros((A, 4), np.float32)
    conf = np.zeros((A, 2), np.float32)
    assert len(orc.postprocess(loc, conf, 640, 640, 640, 640)) == 0
    # many identical boxes on one anchor cell all collapse to one; distinct cells survive
    conf[:, 1] = 0.0
    conf[0:2, 1] = [0.9, 0.8]          # same cell, IoU high -> one survivor
    conf[5000, 1] = 0.7
    conf[12800, 1] = 0.95              # a stride-16 anchor somewhere else
    out = orc.postprocess(loc, conf, 640, 640, 640, 640, max_faces=2)
    assert len(out) == 2 and out["score"].tolist() == pytest.approx([0.95, 0.9])
