"""Detector with dwpw_wave_kernel (kernels_det_wave.hip) against the same detector with dwpw_mfma_kernel: head outputs bit for bit, and both
against the fp32 oracle.  Needs the tuning build (make TUNING=1: FRT_DWPW_WAVE is a tuning switch); runs itself once per setting."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import __graft_entry__ as entry
    frt = entry.load_pkg()
    s = frt.synth
    out, B = sys.argv[2], int(sys.argv[3])
    tmp = tempfile.mkdtemp()
    sd = s.retinaface_state(1)
    path = frt.write_weights(os.path.join(tmp, "det.frtw"), sd, 1)
    det = frt.RetinaFace(path, 640, 640, (3, 640, 640), B, 4)
    fr = np.concatenate([s.make_frames(4, 640, 640)] * ((B + 3) // 4))[:B]
    fr = np.ascontiguousarray(fr[:, ::-1][np.arange(B) % 2 == 0].repeat(2, 0)[:B]) if B > 4 else fr
    x = np.ascontiguousarray((fr.astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))
    loc, conf = det.doInference(x)
    np.savez(out, loc=loc, conf=conf, x=x[:2])
    sys.exit(0)

tmp = tempfile.mkdtemp()
# optional: another tuning switch and its two values, e.g.  FRT_DWPW_PIX 0 7
SWITCH, OFF, ON = (sys.argv[1:4] + ["FRT_DWPW_WAVE", "0", "1"][len(sys.argv[1:4]):]) if len(sys.argv) > 1 else ("FRT_DWPW_WAVE", "0", "1")
lib = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd", "libfrt_tuning.so")
res = {}
for B in (3, 32):
    for mode in ("0", "1"):
        env = dict(os.environ, FRT_LIB=lib, FRT_DWPW_WAVE_ANYB="1")  # (ANYB: also below the batch thresholds)
        env[SWITCH] = ON if mode == "1" else OFF
        out = os.path.join(tmp, "o%s_%d.npz" % (mode, B))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out, str(B)], env=env, check=True)
        res[mode] = np.load(out)
    a, b = res["0"], res["1"]
    print("B=%d: loc identical %s conf identical %s | max |d| loc %.3g conf %.3g" % (
        B, np.array_equal(a["loc"], b["loc"]), np.array_equal(a["conf"], b["conf"]),
        float(np.abs(a["loc"] - b["loc"]).max()), float(np.abs(a["conf"] - b["conf"]).max())), flush=True)
from oracle import nets  # noqa: E402
import __graft_entry__ as entry  # noqa: E402
frt = entry.load_pkg()
sd = frt.synth.retinaface_state(1)
oloc, oconf = nets.retinaface_forward(sd, res["1"]["x"])
print("wave kernel vs fp32 oracle (2 frames): loc %.3g conf %.3g" % (
    float(np.abs(res["1"]["loc"][:2] - oloc).max()), float(np.abs(res["1"]["conf"][:2] - oconf).max())))
