"""Deterministic synthetic weights, frames and galleries.

The reference ships no weights (``/root/reference/.gitignore:7`` ignores ``weight/``) and there is no network, so every
test and benchmark runs on *synthetic* parameters of the exact architectures the reference converts
(``conversion/retina/models/retinaface_trim.py`` with ``cfg_mnet``; ``conversion/arcface/model_irse.py`` ``IR_50`` /
``IR_SE_50``).  Names and shapes follow the PyTorch ``state_dict`` of those modules so that the same dictionary can be
loaded into the reference ``nn.Module`` (golden-vector generation, ``tests/golden/make_golden.py``), into the oracle
(``oracle/nets.py``) and - via :mod:`weights_io` - into ``libfrt.so``.

Scales are chosen so a random network behaves like a trained one where it matters for testing: activations stay O(1)
(fp16-safe through 50 layers), a few hundred of the 16 800 anchors pass the 0.6 score threshold, decoded boxes are
sane, and embeddings of different faces are not collapsed onto one direction.
"""
from collections import OrderedDict

import numpy as np

BN_EPS = 1e-5


def _rng(seed, tag):
    # independent stream per tensor: stable under re-ordering / adding tensors
    h = np.uint64(1469598103934665603)
    for ch in tag.encode():
        h = np.uint64((int(h) ^ ch) * 1099511628211 & 0xFFFFFFFFFFFFFFFF)
    return np.random.Generator(np.random.PCG64([int(seed), int(h)]))


def _conv(sd, seed, name, cout, cin_g, k, gain=2.0, scale=1.0):
    fan_in = cin_g * k * k
    w = _rng(seed, name).standard_normal((cout, cin_g, k, k)).astype(np.float32)
    sd[name] = (w * np.float32(np.sqrt(gain / fan_in) * scale)).astype(np.float32)


def _bn(sd, seed, prefix, c, gamma=1.0, gamma_jit=0.1, beta_std=0.1, mean_std=0.1):
    r = _rng(seed, prefix)
    sd[prefix + ".weight"] = (gamma * (1.0 + gamma_jit * (2 * r.random(c) - 1))).astype(np.float32)
    sd[prefix + ".bias"] = (beta_std * r.standard_normal(c)).astype(np.float32)
    sd[prefix + ".running_mean"] = (mean_std * r.standard_normal(c)).astype(np.float32)
    sd[prefix + ".running_var"] = (0.8 + 0.4 * r.random(c)).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------------
# RetinaFace mobilenet0.25 (trimmed: no landmark head) - names as in retinaface_trim.RetinaFace(cfg_mnet).state_dict()
# ----------------------------------------------------------------------------------------------------------------------
MNET_STAGES = OrderedDict(
    stage1=[("bn", 3, 8, 2), ("dw", 8, 16, 1), ("dw", 16, 32, 2), ("dw", 32, 32, 1), ("dw", 32, 64, 2), ("dw", 64, 64, 1)],
    stage2=[("dw", 64, 128, 2)] + [("dw", 128, 128, 1)] * 5,
    stage3=[("dw", 128, 256, 2), ("dw", 256, 256, 1)],
)


# per-level face-logit offsets, calibrated once on make_frame(0..3) so that ~0.5 % / 2 % / 5 % of the stride-8/16/32
# anchors exceed det_threshold_bbox = 0.6 (a few hundred candidates per frame, spread over the three levels)
CLASS_BIAS = (-4.90, -2.65, -2.48)


# public ArcFace 112x112 alignment template (x, y) x 5: eyes, nose, mouth corners
ARC_TEMPLATE = (38.2946, 51.6963, 73.5318, 51.5014, 56.0252, 71.7366, 41.5493, 92.3655, 70.7299, 92.2041)


def retinaface_state(seed=1, class_bias=CLASS_BIAS, class_scale=1.0, bbox_scale=3.0, landmarks=False):
    """state_dict of ``RetinaFace(cfg_mnet, 'test')`` (``conversion/retina/models/retinaface_trim.py:48-127``).

    ``class_bias`` is added to the face-class logit so that only O(1 %) of anchors pass ``det_threshold_bbox``.
    ``landmarks=True`` additionally emits the LandmarkHead of ``retinaface.py:37-46`` (optional D1 mode).
    """
    sd = OrderedDict()
    for st, layers in MNET_STAGES.items():
        for i, (kind, cin, cout, _s) in enumerate(layers):
            p = "body.%s.%d" % (st, i)
            if kind == "bn":
                # input is mean-subtracted 8-bit pixels (|x| up to ~128): bring activations to O(1)
                _conv(sd, seed, p + ".0.weight", cout, cin, 3, scale=1.0 / 48.0)
                _bn(sd, seed, p + ".1", cout)
            else:
                _conv(sd, seed, p + ".0.weight", cin, 1, 3)
                _bn(sd, seed, p + ".1", cin)
                _conv(sd, seed, p + ".3.weight", cout, cin, 1)
                _bn(sd, seed, p + ".4", cout)
    for i, cin in enumerate((64, 128, 256)):
        _conv(sd, seed, "fpn.output%d.0.weight" % (i + 1), 64, cin, 1)
        _bn(sd, seed, "fpn.output%d.1" % (i + 1), 64)
    for m in ("merge1", "merge2"):
        _conv(sd, seed, "fpn.%s.0.weight" % m, 64, 64, 3)
        _bn(sd, seed, "fpn.%s.1" % m, 64)
    for s in (1, 2, 3):
        for nm, cin, cout in (("conv3X3", 64, 32), ("conv5X5_1", 64, 16), ("conv5X5_2", 16, 16), ("conv7X7_2", 16, 16),
                              ("conv7x7_3", 16, 16)):
            _conv(sd, seed, "ssh%d.%s.0.weight" % (s, nm), cout, cin, 3)
            _bn(sd, seed, "ssh%d.%s.1" % (s, nm), cout)
    for i in range(3):
        n = "ClassHead.%d.conv1x1" % i
        _conv(sd, seed, n + ".weight", 4, 64, 1, gain=1.0, scale=class_scale * (1.0, 1.25, 2.0)[i])
        b = 0.1 * _rng(seed, n + ".bias").standard_normal(4)
        b[1::2] += class_bias[i] if hasattr(class_bias, "__len__") else class_bias  # channel 2l+1 = face logit of anchor l
        sd[n + ".bias"] = b.astype(np.float32)
    for i in range(3):
        n = "BboxHead.%d.conv1x1" % i
        _conv(sd, seed, n + ".weight", 8, 64, 1, gain=1.0, scale=bbox_scale * (1.0, 1.25, 2.0)[i])
        sd[n + ".bias"] = (0.1 * _rng(seed, n + ".bias").standard_normal(8)).astype(np.float32)
    if landmarks:
        for i in range(3):
            n = "LandmarkHead.%d.conv1x1" % i
            _conv(sd, seed, n + ".weight", 20, 64, 1, gain=1.0, scale=0.6)
            # bias = a face-like 5-point layout spanning about one anchor (point = centre + pre * 0.1 * anchor size), so that
            # the similarity fit of the alignment mode is well conditioned; the random weights perturb it per anchor
            tmpl = (np.array(ARC_TEMPLATE, np.float64).reshape(5, 2) - 56.0) / 112.0 * 10.0
            b = np.tile(tmpl.reshape(-1), 2) + 0.1 * _rng(seed, n + ".bias").standard_normal(20)
            sd[n + ".bias"] = b.astype(np.float32)
    return sd


# ----------------------------------------------------------------------------------------------------------------------
# ArcFace IR-50 / IR-SE-50 - names as in model_irse.Backbone([112,112], 50, mode).state_dict()
# ----------------------------------------------------------------------------------------------------------------------
def ir_units(num_layers=50):
    """(in_channel, depth, stride) per unit; ``conversion/arcface/model_irse.py:97-125``."""
    cfg = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}[num_layers]
    units = []
    for (cin, depth), n in zip(((64, 64), (64, 128), (128, 256), (256, 512)), cfg):
        units.append((cin, depth, 2))
        units += [(depth, depth, 1)] * (n - 1)
    return units


def arcface_state(seed=2, mode="ir", num_layers=50, calib=None):
    """state_dict of ``Backbone([112,112], 50, mode)`` (``conversion/arcface/model_irse.py:128-173``).

    ``calib`` = optional (mean[512], var[512]) for the final ``BatchNorm1d`` running statistics (see
    :func:`load_calibration`): centres the 512 features like a trained network's would be.
    """
    assert mode in ("ir", "ir_se")
    sd = OrderedDict()
    _conv(sd, seed, "input_layer.0.weight", 64, 3, 3)
    _bn(sd, seed, "input_layer.1", 64)
    sd["input_layer.2.weight"] = (0.25 + 0.1 * (2 * _rng(seed, "input_layer.2").random(64) - 1)).astype(np.float32)
    for i, (cin, depth, stride) in enumerate(ir_units(num_layers)):
        p = "body.%d" % i
        if cin != depth:
            _conv(sd, seed, p + ".shortcut_layer.0.weight", depth, cin, 1, gain=1.0)
            _bn(sd, seed, p + ".shortcut_layer.1", depth)
        _bn(sd, seed, p + ".res_layer.0", cin)
        _conv(sd, seed, p + ".res_layer.1.weight", depth, cin, 3)
        sd[p + ".res_layer.2.weight"] = (0.25 + 0.1 * (2 * _rng(seed, p + ".prelu").random(depth) - 1)).astype(np.float32)
        _conv(sd, seed, p + ".res_layer.3.weight", depth, depth, 3, gain=1.0)
        # damp the residual branch so the stream does not grow over 24 units (fp16 range, like a trained net)
        _bn(sd, seed, p + ".res_layer.4", depth, gamma=0.5)
        if mode == "ir_se":
            _conv(sd, seed, p + ".res_layer.5.fc1.weight", depth // 16, depth, 1)
            _conv(sd, seed, p + ".res_layer.5.fc2.weight", depth, depth // 16, 1, gain=1.0)
    _bn(sd, seed, "output_layer.0", 512)
    r = _rng(seed, "output_layer.3")
    sd["output_layer.3.weight"] = (r.standard_normal((512, 512 * 7 * 7)) * np.sqrt(1.0 / (512 * 49))).astype(np.float32)
    sd["output_layer.3.bias"] = (0.05 * r.standard_normal(512)).astype(np.float32)
    _bn(sd, seed, "output_layer.4", 512)
    if calib is not None:
        sd["output_layer.4.running_mean"] = np.asarray(calib[0], np.float32)
        sd["output_layer.4.running_var"] = np.asarray(calib[1], np.float32)
    return sd


def load_calibration(mode="ir"):
    """BatchNorm1d running statistics measured once over synthetic faces (``tests/golden/make_golden.py --calib``)."""
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_calib_%s.npz" % mode)
    if not os.path.exists(path):
        return None
    z = np.load(path)
    return z["mean"], z["var"]


# ----------------------------------------------------------------------------------------------------------------------
# frames / faces / gallery
# ----------------------------------------------------------------------------------------------------------------------
def _box_blur(a, k):
    """Separable box blur with edge replication along axes 0 and 1 (k odd)."""
    if k <= 1:
        return a
    r = k // 2
    for ax in (0, 1):
        pad = [(0, 0)] * a.ndim
        pad[ax] = (r, r)
        p = np.pad(a, pad, mode="edge")
        c = np.cumsum(p, axis=ax, dtype=np.float64)
        z = np.zeros_like(np.take(c, [0], axis=ax))
        c = np.concatenate([z, c], axis=ax)
        n = a.shape[ax]
        hi = np.take(c, range(k, k + n), axis=ax)
        lo = np.take(c, range(0, n), axis=ax)
        a = (hi - lo) / k
    return a


def make_frame(idx, rows=640, cols=640, seed=0xFACE0000):
    """u8 BGR HWC frame: low-passed uniform noise (SURVEY §8(d) 'Synthetic inputs'), contrast-stretched."""
    r = np.random.Generator(np.random.PCG64([int(seed), int(idx)]))
    coarse = r.random((rows // 8 + 2, cols // 8 + 2, 3))
    up = np.kron(coarse, np.ones((8, 8, 1)))[:rows, :cols]
    sm = _box_blur(up, 9)
    fine = _box_blur(r.random((rows, cols, 3)), 3)
    img = 0.75 * (sm - 0.5) * 2.2 + 0.25 * (fine - 0.5) * 2.0
    out = np.clip(127.5 + 127.5 * img, 0, 255)
    return np.ascontiguousarray(np.rint(out).astype(np.uint8))


def make_frames(n, rows=640, cols=640, seed=0xFACE0000, start=0):
    return np.stack([make_frame(start + i, rows, cols, seed) for i in range(n)])


def make_faces(n, seed=0xFACE1000):
    """u8 BGR 112x112 crops (smooth noise) for recogniser-only tests."""
    return np.stack([make_frame(i, 112, 112, seed) for i in range(n)])


def make_gallery(n, d=512, seed=3, chunk=1 << 16):
    """N x D float32, rows i.i.d. N(0,1) then L2-normalised."""
    out = np.empty((n, d), np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        g = np.random.Generator(np.random.PCG64([int(seed), s])).standard_normal((e - s, d)).astype(np.float32)
        g /= np.sqrt((g.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
        out[s:e] = g
    return out


def make_queries(gallery, idx, noise=0.01, seed=4):
    """Gallery rows at ``idx`` + N(0, noise^2) re-normalised: known answers with large margins."""
    r = np.random.Generator(np.random.PCG64(int(seed)))
    q = gallery[np.asarray(idx)] + noise * r.standard_normal((len(idx), gallery.shape[1])).astype(np.float32)
    q /= np.sqrt((q.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    return q.astype(np.float32)
