"""Run only the recogniser (F faces, a few passes) - a short target for rocprofv3 kernel traces / PMC passes."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

frt = entry.load_pkg()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = frt.synth
tmp = tempfile.mkdtemp()
MODE = os.environ.get("FRT_PROF_MODE", "ir")
path = frt.write_weights(os.path.join(tmp, "rec.frtw"), s.arcface_state(2, MODE, calib=s.load_calibration(MODE)), 2 if MODE == "ir" else 3)
rec = frt.ArcFaceIR50(path, maxBatchSize=F)
if os.environ.get("FRT_PROF_FP32"):
    rec.setPrecision(True)
x = np.random.default_rng(0).standard_normal((F, 3, 112, 112)).astype(np.float32) * 0.5
for _ in range(reps):
    e = rec.doInference(x)
print("ok", e.shape, float(np.abs(e).sum()))
