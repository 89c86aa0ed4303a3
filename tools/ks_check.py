"""Medium-batch recogniser path (kernels_arc_ks.hip) against the fp32 oracle and the strip kernels: 12 / 16 / 24 / 32 / 40 / 48 faces."""
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
sizes = [int(v) for v in sys.argv[1:]] or [12, 16, 24, 32, 40, 48]
for mode in ("ir", "ir_se"):
    sd = s.arcface_state(2, mode, calib=s.load_calibration(mode))
    path = frt.write_weights(os.path.join(tmp, mode + ".frtw"), sd, 2 if mode == "ir" else 3)
    x = np.random.default_rng(0).standard_normal((max(sizes), 3, 112, 112)).astype(np.float32) * 0.5
    ref = nets.arcface_forward(sd, x[:8])
    big = frt.ArcFaceIR50(path, maxBatchSize=64)
    e64 = big.doInference(np.concatenate([x[:8]] * 8))[:8]
    big.close()
    print(mode, "strip kernels (64) vs oracle: 1-cos max %.3g" % float((1 - (e64 * ref).sum(1)).max()), flush=True)
    for F in sizes:
        rec = frt.ArcFaceIR50(path, maxBatchSize=F)
        e = rec.doInference(x[:F])
        e2 = rec.doInference(np.concatenate([x[F - 5:F], x[:F - 5]]))  # the same faces at other positions of the batch: bit for bit
        rec.close()
        print(mode, "faces", F, "vs oracle 1-cos max %.3g" % float((1 - (e[:8] * ref).sum(1)).max()),
              "| vs strip 1-cos max %.3g |d| max %.3g" % (float((1 - (e[:8] * e64).sum(1)).max()), float(np.abs(e[:8] - e64).max())),
              "| position independent:", bool(np.array_equal(e2[5:], e[:F - 5]) and np.array_equal(e2[:5], e[F - 5:])),
              "finite", bool(np.isfinite(e).all()), flush=True)
