#!/usr/bin/env python3
"""Generates tests/golden/jpeg_vectors.npz: small JPEG streams written by PIL's libjpeg-turbo together with PIL's own decode of them, and
small BGR images together with PIL's encode of them (quality 95 / 75, 4:2:0 - what cv::imencode does by default, src/app.cpp:328).
These are DATA produced by the third-party library the reference's OpenCV wraps (the reference itself holds no JPEG fixtures); the
tests compare oracle/jpegops.py, the host entropy coder and the device kernels against them without needing PIL at run time.

    python tests/golden/make_jpeg_golden.py
"""
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

frt = entry.load_pkg()
s = frt.synth


def noisy(seed, h, w):
    img = s.make_frame(seed, max(h, 8), max(w, 8))[:h, :w]
    return np.clip(img.astype(int) + np.random.default_rng(seed).integers(-25, 25, img.shape), 0, 255).astype(np.uint8)


def enc(img_bgr, **kw):
    b = io.BytesIO()
    Image.fromarray(img_bgr[..., ::-1]).save(b, "JPEG", **kw)
    return b.getvalue()


def dec(data):
    im = Image.open(io.BytesIO(data))
    a = np.array(im)
    return np.ascontiguousarray(a[..., ::-1]) if a.ndim == 3 else np.stack([a] * 3, -1)


out = {}
cases = [("444_q95", 40, 56, dict(quality=95, subsampling=0)), ("422_q80", 37, 53, dict(quality=80, subsampling=1)),
         ("420_q95", 64, 48, dict(quality=95, subsampling=2)), ("420_odd_q60", 33, 71, dict(quality=60, subsampling=2)),
         ("420_rst", 48, 80, dict(quality=90, subsampling=2, restart_marker_blocks=2)), ("420_tiny", 3, 5, dict(quality=95, subsampling=2)),
         ("420_q100", 24, 24, dict(quality=100, subsampling=2))]
for i, (name, h, w, kw) in enumerate(cases):
    data = enc(noisy(10 + i, h, w), **kw)
    out["dec_%s_jpg" % name] = np.frombuffer(data, np.uint8)
    out["dec_%s_bgr" % name] = dec(data)
b = io.BytesIO()
Image.fromarray(noisy(30, 50, 34)[..., 0]).save(b, "JPEG", quality=85)
out["dec_gray_q85_jpg"] = np.frombuffer(b.getvalue(), np.uint8)
out["dec_gray_q85_bgr"] = dec(b.getvalue())
# progressive streams (cv::imdecode accepts them, src/app.cpp:296): libjpeg's default scan script - DC first, spectral bands,
# successive-approximation refinement passes - for colour (4:4:4, 4:2:0, odd size, restart intervals) and grey
for name, h, w, kw in [("prog_444_q85", 40, 40, dict(quality=85, subsampling=0)), ("prog_420_odd_q60", 33, 71, dict(quality=60, subsampling=2)),
                       ("prog_422_q80", 37, 53, dict(quality=80, subsampling=1)), ("prog_420_rst", 48, 80, dict(quality=90, subsampling=2, restart_marker_blocks=2))]:
    data = enc(noisy(60 + h, h, w), progressive=True, **kw)
    out["dec_%s_jpg" % name] = np.frombuffer(data, np.uint8)
    out["dec_%s_bgr" % name] = dec(data)
b = io.BytesIO()
Image.fromarray(noisy(31, 40, 40)[..., 0]).save(b, "JPEG", quality=85, progressive=True)
out["dec_prog_gray_q85_jpg"] = np.frombuffer(b.getvalue(), np.uint8)
out["dec_prog_gray_q85_bgr"] = dec(b.getvalue())
# an arithmetic-coded frame header (SOF9 in place of SOF0): must be refused, not mis-decoded
arith = bytearray(out["dec_420_q95_jpg"].tobytes())
arith[arith.index(b"\xff\xc0") + 1] = 0xC9
out["unsupported_arithmetic_jpg"] = np.frombuffer(bytes(arith), np.uint8)
for i, (h, w, q) in enumerate([(112, 112, 95), (112, 112, 75), (37, 53, 95), (16, 31, 95)]):
    img = noisy(40 + i, h, w)
    out["enc_%d_bgr" % i] = img
    out["enc_%d_q" % i] = np.int32(q)
    out["enc_%d_jpg" % i] = np.frombuffer(enc(img, quality=q, subsampling=2), np.uint8)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_vectors.npz"), **out)
print("wrote", len(out), "arrays")
