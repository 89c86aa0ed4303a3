"""Small-batch recogniser path (kernels_arc_small.hip) against the strip kernels and the fp32 oracle: same faces, batch 1 / 4 / 5 / 8 / 64."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

frt = entry.load_pkg()
from oracle import nets  # noqa: E402

s = frt.synth
tmp = tempfile.mkdtemp()
for mode in ("ir", "ir_se"):
    sd = s.arcface_state(2, mode, calib=s.load_calibration(mode))
    path = frt.write_weights(os.path.join(tmp, mode + ".frtw"), sd, 2 if mode == "ir" else 3)
    x = np.random.default_rng(0).standard_normal((8, 3, 112, 112)).astype(np.float32) * 0.5
    ref = nets.arcface_forward(sd, x)
    big = frt.ArcFaceIR50(path, maxBatchSize=64)
    e64 = big.doInference(np.concatenate([x] * 8))[:8]  # 64 faces: the strip kernels
    print(mode, "strip kernels vs oracle: 1-cos max", float((1 - (e64 * ref).sum(1)).max()))
    for mb in (1, 4, 5, 8):
        rec = frt.ArcFaceIR50(path, maxBatchSize=mb)
        e = rec.doInference(x)
        print(mode, "maxBatch", mb, "vs oracle 1-cos max %.3g" % float((1 - (e * ref).sum(1)).max()),
              "| vs strip kernels 1-cos max %.3g, |d| max %.3g" % (float((1 - (e * e64).sum(1)).max()), float(np.abs(e - e64).max())))
        rec.close()
    big.close()
