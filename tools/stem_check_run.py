"""Target of the fused-stem check (kernels_det_stem.hip): FRT_LIB=<libfrt_tuning.so> FRT_DET_STEM_CHECK=1 python tools/stem_check_run.py
prints how many elements of the 32-channel tensor differ between the three stand-alone kernels and det_stem_kernel (0)."""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
frt = entry.load_pkg()
s = frt.synth
import numpy as np
tmp = tempfile.mkdtemp()
path = frt.write_weights(os.path.join(tmp, "det.frtw"), s.retinaface_state(1), 1)
det = frt.RetinaFace(path, 640, 640, (3, 640, 640), 12, 4)
fr = np.concatenate([s.make_frames(4, 640, 640)] * 3)
out = det.findFaceBatch(fr)
print("ok", sum(len(o) for o in out))
