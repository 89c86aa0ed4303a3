"""The C++ drop-in shells (include/frt/*.h) compile as C++11 with g++ and behave like the reference classes."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd")


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cpp") / "dropin_demo")
    # TWO translation units that both include frt/arcface.h, like the reference's app.cpp (src/app.cpp:3-6) and db.cpp (via
    # src/db.h:8): a non-inline `int ArcFaceIR50::classCount` in the header would fail here with "multiple definition"
    tmp = os.path.dirname(exe)
    objs = []
    for src in ("dropin_demo.cpp", "dropin_db.cpp"):
        obj = os.path.join(tmp, src[:-4] + ".o")
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                               os.path.join(ROOT, "tests", "cpp", src), "-o", obj])
        objs.append(obj)
    sqlite = next(p for p in ("/lib/x86_64-linux-gnu/libsqlite3.so.0", "/usr/lib/x86_64-linux-gnu/libsqlite3.so.0") if os.path.exists(p))
    subprocess.check_call(["g++", "-o", exe] + objs + [os.path.join(PKG, "libfrt.so"), sqlite, "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_no_escape_hatch_macro_left():
    """The header needs no per-TU opt-out macro any more (round-1 VERDICT, boundary item 5)."""
    for root, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "FRT_ARCFACE_NO_STATIC_DEFINITION" not in open(os.path.join(root, f)).read()


def test_shells_compile_and_report_missing_engine(demo):
    out = subprocess.run([demo, "--selftest"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "selftest ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_inference_call_sequence_matches_python_binding(demo, frt, synth, blobs, tmp_path):
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    H, W, N = 160, 224, 300
    det = frt.RetinaFace(dpath, W, H, (3, H, W), 1, 4, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H)
    for seed in range(16):  # first synthetic frame of this size with at least one detection
        frame = synth.make_frame(seed, H, W)
        boxes = det.findFace(frame)
        if len(boxes):
            break
    emb = rec.forward(frame, boxes)
    gal = synth.make_gallery(N)
    gal[100:100 + len(emb)] = emb
    (tmp_path / "frame.bin").write_bytes(frame.tobytes())
    (tmp_path / "gal.bin").write_bytes(gal.tobytes())
    # the same gallery as the reference's SQLite database (src/db.cpp:58-65), read by the second TU row by row
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gallery_sqlite
    gallery_sqlite.write_gallery(str(tmp_path / "gal.db"), [str(i) for i in range(N)], gal)
    for gfile in ("gal.bin", "gal.db"):
        out = subprocess.run([demo, dpath, rpath, str(tmp_path / "frame.bin"), str(H), str(W), str(tmp_path / gfile), str(N)],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = [l.split() for l in out.stdout.splitlines() if l and l[0].isdigit()]
        assert len(lines) == len(boxes) > 0
        for i, l in enumerate(lines):
            assert [int(v) for v in l[:4]] == [int(boxes[i][k]) for k in ("x1", "y1", "x2", "y2")]
            assert abs(float(l[4]) - float(boxes[i]["score"])) < 1e-6
            assert int(l[5]) == 100 + i and float(l[6]) > 0.9999
    det.close()
    rec.close()
