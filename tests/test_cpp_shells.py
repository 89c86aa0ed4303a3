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
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "dropin_demo.cpp"), "-o", exe, os.path.join(PKG, "libfrt.so"),
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


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
    out = subprocess.run([demo, dpath, rpath, str(tmp_path / "frame.bin"), str(H), str(W), str(tmp_path / "gal.bin"), str(N)],
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
