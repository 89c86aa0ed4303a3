"""Run only the detector (B frames of 640x640, a few passes) - a short target for rocprofv3 kernel traces."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

frt = entry.load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = frt.synth
tmp = tempfile.mkdtemp()
path = frt.write_weights(os.path.join(tmp, "det.frtw"), s.retinaface_state(1), 1)
det = frt.RetinaFace(path, 640, 640, (3, 640, 640), B, 4)
frames = s.make_frames(4)
import numpy as np  # noqa: E402
frames = np.concatenate([frames] * ((B + 3) // 4))[:B]
for _ in range(reps):
    out = det.findFaceBatch(frames)
print("ok", sum(len(o) for o in out))
