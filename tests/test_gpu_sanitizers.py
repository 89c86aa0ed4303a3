"""ASan + UBSan over libfrt's host code WITH a device behind it: the pipeline's slot / staging / ticket bookkeeping, the object
mutexes and stage events and the streaming gallery loader under four concurrent caller threads (tests/cpp/pipeline_stress.cpp).
libfrt_asan.so = the three host translation units of libfrt compiled with -fsanitize=address,undefined (`make ASAN=1`, also done by
__graft_entry__.build()); device code is untouched."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
PKG = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd")


def test_pipeline_host_code_under_asan_ubsan(frt, synth, blobs, tmp_path):
    lib = os.path.join(PKG, "libfrt_asan.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-j8", "ASAN=1", "-C", os.path.join(PKG, "csrc")])
    exe = str(tmp_path / "pipeline_stress")
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "pipeline_stress.cpp"), "-o", exe, lib,
                           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, H, W, N = 4, 320, 320, 40000
    frames = np.concatenate([synth.make_frames(B, H, W), synth.make_frames(B, H, W, start=50)])
    (tmp_path / "frames.bin").write_bytes(frames.tobytes())
    (tmp_path / "gal.bin").write_bytes(synth.make_gallery(N).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="protect_shadow_gap=0:detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, dpath, rpath, str(tmp_path / "frames.bin"), str(B), str(H), str(W), str(tmp_path / "gal.bin"), str(N)],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "stress ok" in out.stdout, out.stdout[-2000:] + out.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-6000:]
