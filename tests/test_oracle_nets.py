"""oracle/nets.py against the golden vectors produced by the reference's own nn.Modules (tests/golden/make_golden.py)."""
import os

import numpy as np

from conftest import GOLDEN, face_input


def test_arcface_ir50_matches_reference_module(synth):
    from oracle import nets
    g = np.load(os.path.join(GOLDEN, "arcface_ir.npz"))
    sd = synth.arcface_state(int(g["seed"]), "ir", calib=synth.load_calibration("ir"))
    x = face_input(synth.make_faces(int(g["n_faces"])))
    emb, blocks, _ = nets.arcface_forward(sd, x, return_blocks=True)
    assert emb.shape == (8, 512)
    assert np.abs(emb - g["embeddings"]).max() < 2e-6
    assert np.allclose((emb.astype(np.float64) ** 2).sum(1), 1, atol=1e-5)
    assert len(blocks) == 25  # input layer + 24 units
    for b, idx, val in zip(blocks, g["block_idx"], g["block_val"]):
        assert np.abs(b[0].reshape(-1)[idx] - val).max() <= 1e-4 * max(1.0, np.abs(val).max())
    # different synthetic faces are not collapsed onto one direction (a meaningful cosine test downstream)
    off = (emb @ emb.T)[~np.eye(8, dtype=bool)]
    assert np.abs(off).max() < 0.3


def test_arcface_ir_se50_matches_reference_module(synth):
    from oracle import nets
    g = np.load(os.path.join(GOLDEN, "arcface_ir_se.npz"))
    sd = synth.arcface_state(int(g["seed"]), "ir_se", calib=synth.load_calibration("ir_se"))
    emb = nets.arcface_forward(sd, face_input(synth.make_faces(int(g["n_faces"]))))
    assert np.abs(emb - g["embeddings"]).max() < 2e-6


def test_retinaface_matches_reference_module(synth):
    from oracle import nets
    g = np.load(os.path.join(GOLDEN, "retinaface_mnet.npz"))
    sd = synth.retinaface_state(int(g["seed"]))
    for tag, (h, w) in (("96x160", (96, 160)), ("288x320", (288, 320)), ("640", (640, 640))):
        fr = synth.make_frames(2, h, w)
        x = np.ascontiguousarray((fr.astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))
        loc, conf = nets.retinaface_forward(sd, x)
        step = int(g["step_" + tag])
        assert loc.shape[1] == 2 * sum(-(-h // s) * -(-w // s) for s in (8, 16, 32))
        assert np.abs(loc[:, ::step] - g["loc_" + tag]).max() < 2e-5
        assert np.abs(conf[:, ::step] - g["conf_" + tag]).max() < 2e-6
        assert np.allclose(conf.sum(-1), 1, atol=1e-6)
        assert np.array_equal((conf[..., 1] > 0.6).sum(1), g["npass_" + tag])
