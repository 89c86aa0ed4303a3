"""Stress of the pipeline's host entry points from several threads (GPU box): synchronous frt_pipeline_run calls - which take the one-stream
path when they find nothing else in flight and the stage streams otherwise - mixed with submit / wait triples and object-level calls, every
result compared with the single-threaded answer for the same frames."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

frt = entry.load_pkg()
s = frt.synth
import tempfile  # noqa: E402

tmp = tempfile.mkdtemp()
MODE = os.environ.get("FRT_PROF_MODE", "ir")
dp = frt.write_weights(os.path.join(tmp, "d.frtw"), s.retinaface_state(1), 1)
rp = frt.write_weights(os.path.join(tmp, "r.frtw"), s.arcface_state(2, MODE, calib=s.load_calibration(MODE)), 2 if MODE == "ir" else 3)
B, K, H, W = 2, 4, 320, 320
det = frt.RetinaFace(dp, W, H, (3, H, W), B, K, 0.4, 0.6)
rec = frt.ArcFaceIR50(rp, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
rec.setGallery(s.make_gallery(5000))
rec.initMatMul()
pipe = frt.Pipeline(det, rec, B)
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
batches = [s.make_frames(B, H, W, start=5 * i) for i in range(NT + 1)]
want = [tuple(a.copy() for a in pipe.run(b)) for b in batches]
errors = []


def sync_worker(i):
    try:
        for it in range(ITERS):
            r, e = pipe.run(batches[i])
            if not (np.array_equal(r, want[i][0]) and np.array_equal(e, want[i][1])):
                errors.append(("run", i, it))
                return
            if it % 7 == i % 7:
                time.sleep(0.002)  # let the pipeline drain now and then: the next call is a lone one
    except Exception as ex:  # noqa: BLE001
        errors.append(("run", i, repr(ex)))


def async_worker(i):
    try:
        res = [np.zeros(B * K, frt.RESULT_DTYPE) for _ in range(3)]
        for it in range(ITERS // 3):
            t = [pipe.submit(batches[i], res[k]) for k in range(3)]
            for k in range(3):
                pipe.wait(t[k])
                if not np.array_equal(res[k], want[i][0]):
                    errors.append(("submit", i, it, k))
                    return
    except Exception as ex:  # noqa: BLE001
        errors.append(("submit", i, repr(ex)))


ts = [threading.Thread(target=sync_worker, args=(i,)) for i in range(NT)] + [threading.Thread(target=async_worker, args=(NT,))]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
print("threads", NT + 1, "iters", ITERS, "seconds %.1f" % (time.perf_counter() - t0), "errors", errors[:5])
sys.exit(1 if errors else 0)
