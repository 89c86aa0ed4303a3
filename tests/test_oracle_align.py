"""Oracle of the optional 5-point alignment mode (oracle/align.py, "parity unpinned" except for the raw landmark head)."""
import os

import numpy as np

from conftest import GOLDEN


def test_landmark_head_matches_reference_module(synth):
    """The raw LandmarkHead output IS pinned: golden from the unmodified conversion/retina/models/retinaface.py."""
    from oracle import nets
    g = np.load(os.path.join(GOLDEN, "retinaface_mnet_ldm.npz"))
    sd = synth.retinaface_state(int(g["seed"]), landmarks=True)
    trimmed = synth.retinaface_state(int(g["seed"]))
    assert all(np.array_equal(sd[k], v) for k, v in trimmed.items())  # the extra head changes nothing else
    for tag, (h, w) in (("96x160", (96, 160)), ("288x320", (288, 320))):
        fr = synth.make_frames(2, h, w)
        x = np.ascontiguousarray((fr.astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))
        loc, conf, ldm = nets.retinaface_forward(sd, x)
        assert ldm.shape == (2, loc.shape[1], 10)
        assert np.abs(ldm[:, ::3] - g["ldm_" + tag]).max() < 2e-5
        assert np.abs(loc[:, ::3] - g["loc_" + tag]).max() < 2e-5
        assert np.abs(conf[:, ::3] - g["conf_" + tag]).max() < 2e-6


def test_decode_zero_offsets_give_anchor_centres(orc):
    from oracle import align
    in_h, in_w, fh, fw = 288, 320, 480, 640  # scale_h 0.6 > scale_w 0.5: rows are letterboxed
    anc = orc.anchors(in_w, in_h)
    idx = np.array([0, 1, 777, len(anc) - 1])
    for a in idx:
        assert np.allclose(align.anchor_geometry(int(a), in_h, in_w), anc[a], rtol=0, atol=1e-7)
    pts = align.decode_landmarks(np.zeros((len(anc), 10), np.float32), idx, in_h, in_w, fh, fw)
    pad = (in_h - 0.5 * fh) / 2
    for n, a in enumerate(idx):
        ex, ey = anc[a, 0] * in_w / 0.5, (anc[a, 1] * in_h - pad) / 0.5
        assert np.allclose(pts[n], [[ex, ey]] * 5, atol=1e-3)
    # one unit of raw output moves a point by 0.1 anchor sizes (in network pixels), i.e. 0.1*size/scale frame pixels
    raw = np.zeros((len(anc), 10), np.float32)
    raw[777, 4:6] = (1.0, -2.0)
    p2 = align.decode_landmarks(raw, np.array([777]), in_h, in_w, fh, fw)[0]
    p0 = pts[2]
    assert np.allclose(p2[2] - p0[2], [0.1 * anc[777, 2] * in_w / 0.5, -0.2 * anc[777, 3] * in_h / 0.5], atol=1e-3)
    assert np.allclose(np.delete(p2, 2, 0), np.delete(p0, 2, 0))


def _rot(theta, s):
    return s * np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])


def test_similarity_recovers_a_known_transform():
    from oracle import align
    t = align.ARC_TEMPLATE.astype(np.float64).reshape(5, 2)
    for theta, s, shift in ((0.0, 1.0, (0, 0)), (0.3, 2.5, (100, 40)), (-1.1, 0.7, (5, 300))):
        A = _rot(theta, s)
        lm = t @ A.T + np.array(shift)                    # landmarks = the template moved into the frame
        M = align.similarity_matrix(lm)                    # must map them back onto the template
        assert np.allclose(lm @ M[:, :2].T + M[:, 2], t, atol=1e-9)
        ok, ia, ib, tx, ty = align.similarity_inverse(lm)
        assert ok
        Minv = np.linalg.inv(np.vstack([M, [0, 0, 1]]))
        assert np.allclose([[ia, ib], [-ib, ia]], Minv[:2, :2], atol=1e-5)
        assert np.allclose(Minv[:2, :2] @ (np.array([10.0, 20.0]) - np.array([tx, ty])), Minv[:2, :2] @ [10, 20] + Minv[:2, 2], atol=2e-3)
    # noisy landmarks: the fit is the least-squares optimum (perturbing any parameter increases the residual)
    r = np.random.default_rng(0)
    lm = t @ _rot(0.2, 1.7).T + (50, 60) + r.normal(0, 1.5, (5, 2))
    M = align.similarity_matrix(lm)
    res = lambda M_: ((lm @ M_[:, :2].T + M_[:, 2] - t) ** 2).sum()
    base = res(M)
    for d in (1e-3, -1e-3):
        for k in range(4):
            a, b, tx, ty = M[0, 0], M[1, 0], M[0, 2], M[1, 2]
            v = [a, b, tx, ty]
            v[k] += d
            assert res(np.array([[v[0], -v[1], v[2]], [v[1], v[0], v[3]]])) > base


def test_align_identity_and_independent_bilinear(synth):
    from scipy import ndimage
    from oracle import align
    frame = synth.make_frames(1, 240, 320)[0]
    t = align.ARC_TEMPLATE.reshape(5, 2)
    # landmarks exactly on the template shifted by an integer offset: the warp is a plain 112x112 copy
    crops, valid = align.align_faces(frame, (t + np.array([40, 30], np.float32))[None])
    assert valid[0] == 1
    assert np.abs(crops[0].astype(int) - frame[30:142, 40:152].astype(int)).max() <= 1
    # rotated/scaled, partly outside the frame: compare with scipy's order-1 map_coordinates (constant 0 border)
    lm = (t.astype(np.float64) @ _rot(0.4, 1.8).T + (180, -20)).astype(np.float32)
    crops, valid = align.align_faces(frame, lm[None])
    Minv = np.linalg.inv(np.vstack([align.similarity_matrix(lm), [0, 0, 1]]))
    oy, ox = np.meshgrid(np.arange(112.0), np.arange(112.0), indexing="ij")
    sx = Minv[0, 0] * ox + Minv[0, 1] * oy + Minv[0, 2]
    sy = Minv[1, 0] * ox + Minv[1, 1] * oy + Minv[1, 2]
    ref = np.stack([ndimage.map_coordinates(frame[..., c].astype(np.float64), [sy, sx], order=1, mode="grid-constant", cval=0.0) for c in range(3)], -1)
    diff = np.abs(crops[0].astype(np.float64) - ref)
    assert (diff > 1.0).mean() < 0.002 and diff.max() < 3.0   # float32 vs float64 coordinates flip a few roundings
    assert (crops[0] == 0).all(-1).any() and (crops[0] > 0).any()  # the zero border really is in view


def test_align_degenerate_landmarks():
    from oracle import align
    frame = np.full((64, 64, 3), 200, np.uint8)
    crops, valid = align.align_faces(frame, np.full((1, 5, 2), 17.0, np.float32))
    assert valid[0] == 0 and not crops.any()
