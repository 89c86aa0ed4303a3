"""C-ABI surface: the library loads, exports every symbol include/frt.h declares, and reports errors the reference's way.
No compute calls here (no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "frt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(frt_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(frt):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(frt.lib, s), "libfrt.so does not export %s" % s
        assert s in frt.ABI, "python binding misses %s" % s


def test_struct_layout_matches_reference_bbox(frt):
    # struct Bbox {int x1,y1,x2,y2; float score;}  (src/common.h:13-16): 20 bytes, no padding
    assert frt.BBOX_DTYPE.itemsize == 20
    assert [frt.BBOX_DTYPE.fields[k][1] for k in ("x1", "y1", "x2", "y2", "score")] == [0, 4, 8, 12, 16]


def test_missing_engine_file_is_reported_like_the_reference(frt):
    h = ctypes.c_void_p()
    rc = frt.lib.frt_detector_create(b"/nonexistent/retina.frtw", 640, 480, 3, 288, 320, 1, 4, 0.4, 0.6, 0, ctypes.byref(h))
    assert rc == frt.FRT_ERR_NOT_FOUND and not h
    assert frt.lib.frt_last_error() == b"Cant find engine file"  # retinaface.cpp:53
    rc = frt.lib.frt_embedder_create(b"/nonexistent/arc.frtw", 3, 112, 112, 512, 1, 0, ctypes.byref(h))
    assert rc == frt.FRT_ERR_NOT_FOUND
    assert frt.lib.frt_last_error() == b"Cant find engine file"  # arcface.cpp:67


def test_wrong_blob_kind_and_bad_shapes(frt, blobs, tmp_path):
    path, _ = blobs("det")
    h = ctypes.c_void_p()
    assert frt.lib.frt_embedder_create(path.encode(), 3, 112, 112, 512, 1, 0, ctypes.byref(h)) == frt.FRT_ERR_FORMAT
    assert frt.lib.frt_detector_create(path.encode(), 640, 480, 1, 288, 320, 1, 4, 0.4, 0.6, 0, ctypes.byref(h)) == frt.FRT_ERR_INVALID
    bad = tmp_path / "bad.frtw"
    bad.write_bytes(b"not a blob at all")
    assert frt.lib.frt_detector_create(str(bad).encode(), 640, 480, 3, 288, 320, 1, 4, 0.4, 0.6, 0, ctypes.byref(h)) == frt.FRT_ERR_FORMAT


def test_null_arguments_are_rejected(frt):
    assert frt.lib.frt_matcher_init(None, None, 1, 512) == frt.FRT_ERR_INVALID
    assert frt.lib.frt_detector_find_faces(None, None, 1, 1, 3, None, None) == frt.FRT_ERR_INVALID
    assert frt.lib.frt_pipeline_sync(None) == frt.FRT_ERR_INVALID
    assert frt.lib.frt_detector_num_anchors(None) == 0


def test_weight_blob_roundtrip(frt, blobs):
    path, sd = blobs("det")
    kind, back = frt.weights_io.read_blob(path)
    assert kind == 1 and list(back) == [k for k in sd]
    for k in sd:
        assert np.array_equal(back[k], sd[k]), k


def test_merge_top1_prefers_higher_sim_then_lower_index(frt):
    ia = np.array([5, 9, -1, 3, 7], np.int32)
    sa = np.array([0.5, 0.7, 0.0, 0.2, 0.9], np.float32)
    ib = np.array([100, 2, 8, -1, 1], np.int32)
    sb = np.array([0.6, 0.7, 0.1, 0.0, 0.9], np.float32)
    io, so = frt.merge_top1(ia, sa, ib, sb)
    assert io.tolist() == [100, 2, 8, 3, 1]
    assert np.allclose(so, [0.6, 0.7, 0.1, 0.2, 0.9])


def test_synthetic_generators_are_deterministic(synth):
    a, b = synth.make_frame(3), synth.make_frame(3)
    assert a.dtype == np.uint8 and a.shape == (640, 640, 3) and np.array_equal(a, b)
    assert not np.array_equal(a, synth.make_frame(4))
    g = synth.make_gallery(1000)
    assert np.allclose((g.astype(np.float64) ** 2).sum(1), 1, atol=1e-6)
    assert np.array_equal(g, synth.make_gallery(1000))
    sd1, sd2 = synth.retinaface_state(1), synth.retinaface_state(1)
    assert all(np.array_equal(sd1[k], sd2[k]) for k in sd1)
    assert sum(v.size for k, v in sd1.items() if "running" not in k) == 422708 - 0  # params incl. BN affine


def test_pth_exporter_strips_prefixes_and_unwraps(frt, tmp_path):
    """weights_io.export_pth (the torch2trt.py replacement): `module.` prefixes and a wrapping {'state_dict': ...} are removed like
    conversion/retina/torch2trt.py:41-61 does; the blob round-trips bit-exactly."""
    import torch
    sd = {"module.body.stage1.0.0.weight": torch.randn(8, 3, 3, 3), "module.body.stage1.0.1.running_var": torch.rand(8) + 0.5,
          "fpn.output1.0.weight": torch.randn(64, 64, 1, 1)}
    for wrap in (False, True):
        pth = tmp_path / ("w%d.pth" % wrap)
        torch.save({"state_dict": sd} if wrap else sd, pth)
        out = tmp_path / ("w%d.frtw" % wrap)
        frt.weights_io.export_pth(str(pth), str(out), frt.weights_io.KIND_RETINAFACE_MNET025)
        kind, back = frt.weights_io.read_blob(str(out))
        assert kind == frt.weights_io.KIND_RETINAFACE_MNET025
        assert set(back) == {"body.stage1.0.0.weight", "body.stage1.0.1.running_var", "fpn.output1.0.weight"}
        for k, v in sd.items():
            assert np.array_equal(back[k.replace("module.", "")], v.numpy())
