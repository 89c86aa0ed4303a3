"""ASan + UBSan run of libfrt's pure-host input parsers (JPEG codec host half, FRTW weight blobs) on valid, truncated and randomly
damaged inputs.  The device-facing host code (pipeline / ticket queue) needs a GPU and is exercised by tests/test_gpu_pipeline.py's
multi-threaded tests instead."""
import os
import subprocess

import numpy as np

from conftest import GOLDEN, ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
CSRC = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd", "csrc")


def test_host_parsers_under_asan_ubsan(frt, synth, tmp_path):
    exe = str(tmp_path / "sanitize_host")
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "cpp", "sanitize_host.cpp"), os.path.join(CSRC, "frt_jpeg.cpp"), "-o", exe])
    vec = np.load(os.path.join(GOLDEN, "jpeg_vectors.npz"))
    jpgs = []
    for k in vec.files:
        if k.endswith("_jpg"):
            p = tmp_path / (k + ".jpg")
            p.write_bytes(vec[k].tobytes())
            jpgs.append(str(p))
    blob = frt.write_weights(str(tmp_path / "det.frtw"), synth.retinaface_state(1), 1)
    out = subprocess.run([exe] + jpgs + ["--", blob], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0 and "sanitize ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
