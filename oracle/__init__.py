"""ORACLE package - test infrastructure only.

CPU restatements of the reference's hot path: ``nets.py`` (both networks, fp32 torch-CPU),
``postproc.c`` (anchors / decode / NMS), ``imgops.c`` (OpenCV letterbox, bicubic crop, normalisation), ``match.py``
(``MatMul::calculate`` + ``getOutputs``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product (``libfrt.so``) never does and has no CPU fallback.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Bbox(ctypes.Structure):
    _fields_ = [("x1", ctypes.c_int32), ("y1", ctypes.c_int32), ("x2", ctypes.c_int32), ("y2", ctypes.c_int32), ("score", ctypes.c_float)]


BBOX_DTYPE = np.dtype([("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4"), ("score", "<f4")])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_anchor_count.restype = ctypes.c_int
        _LIB.orc_make_anchors.restype = ctypes.c_int
        _LIB.orc_postprocess.restype = ctypes.c_int
        _LIB.orc_crop_face.restype = ctypes.c_int
    return _LIB


def _p(a, t=ctypes.c_void_p):
    return a.ctypes.data_as(t)


def anchors(w, h):
    """``create_anchor_retinaface`` (retinaface.cpp:210-240) -> float32 [A,4] (cx, cy, sx, sy)."""
    L = lib()
    n = L.orc_anchor_count(int(w), int(h))
    out = np.empty((n, 4), np.float32)
    L.orc_make_anchors(int(w), int(h), _p(out))
    return out


def postprocess(loc, conf, in_w, in_h, frame_w, frame_h, nms_thr=0.4, bbox_thr=0.6, max_faces=4, return_candidates=False):
    """``RetinaFace::postprocessing`` (retinaface.cpp:154-208) for one frame -> structured array of Bbox."""
    L = lib()
    loc = np.ascontiguousarray(loc, np.float32).reshape(-1, 4)
    conf = np.ascontiguousarray(conf, np.float32).reshape(-1, 2)
    A = L.orc_anchor_count(int(in_w), int(in_h))
    assert loc.shape[0] == A and conf.shape[0] == A, (loc.shape, A)
    out = np.zeros(max(max_faces, 1), BBOX_DTYPE)
    cand = np.zeros(A, BBOX_DTYPE)
    cidx = np.zeros(A, np.int32)
    nc = ctypes.c_int(0)
    n = L.orc_postprocess(_p(loc), _p(conf), int(in_w), int(in_h), int(frame_w), int(frame_h), ctypes.c_float(nms_thr),
                          ctypes.c_float(bbox_thr), int(max_faces), _p(out), _p(cand), _p(cidx), ctypes.byref(nc))
    if return_candidates:
        return out[:n].copy(), cand[:nc.value].copy(), cidx[:nc.value].copy()
    return out[:n].copy()


def det_preprocess(frame, in_h, in_w):
    """``RetinaFace::preprocess`` (retinaface.cpp:106-136): u8 BGR HWC -> float32 [3,in_h,in_w]."""
    frame = np.ascontiguousarray(frame, np.uint8)
    fh, fw, _ = frame.shape
    out = np.empty((3, in_h, in_w), np.float32)
    lib().orc_det_preprocess(_p(frame), fh, fw, ctypes.c_size_t(fw * 3), int(in_h), int(in_w), _p(out))
    return out


def resize_linear(img, dh, dw):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dh, dw, 3), np.uint8)
    lib().orc_resize_linear_u8c3(_p(img), img.shape[0], img.shape[1], ctypes.c_size_t(img.shape[1] * 3), _p(out), dh, dw, ctypes.c_size_t(dw * 3))
    return out


def resize_cubic(img, dh, dw):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dh, dw, 3), np.uint8)
    lib().orc_resize_cubic_u8c3(_p(img), img.shape[0], img.shape[1], ctypes.c_size_t(img.shape[1] * 3), _p(out), dh, dw, ctypes.c_size_t(dw * 3))
    return out


def crop_faces(frame, boxes, out_h=112, out_w=112):
    """``getCroppedFaces`` (arcface.cpp:3-17) -> u8 [F,out_h,out_w,3] BGR; raises on an empty ROI like OpenCV would."""
    frame = np.ascontiguousarray(frame, np.uint8)
    fh, fw, _ = frame.shape
    out = np.empty((len(boxes), out_h, out_w, 3), np.uint8)
    for i, b in enumerate(boxes):
        rc = lib().orc_crop_face(_p(frame), fh, fw, ctypes.c_size_t(fw * 3), int(b["x1"]), int(b["y1"]), int(b["x2"]), int(b["y2"]),
                                 int(out_h), int(out_w), _p(out[i]))
        if rc != 0:
            raise ValueError("empty ROI for box %d" % i)
    return out


def face_normalize(crops):
    """``preprocessFaces`` (arcface.cpp:116-129): u8 BGR [F,h,w,3] -> float32 planar RGB [F,3,h,w]."""
    crops = np.ascontiguousarray(crops, np.uint8)
    f, h, w, _ = crops.shape
    out = np.empty((f, 3, h, w), np.float32)
    for i in range(f):
        lib().orc_face_normalize(_p(crops[i]), h, w, _p(out[i]))
    return out
