"""Matcher with few queries (argv: numbers of queries), 1M x 512 gallery: per-call time of top1 - the shape of one frame / a four-frame batch."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

import __graft_entry__ as entry

frt = entry.load_pkg()
gal = frt.synth.make_gallery(1000000)
m = frt.MatMul(0)
m.init(gal)
for F in [int(v) for v in sys.argv[1:]] or [4, 16, 32, 64, 128]:
    q = frt.synth.make_queries(gal, np.arange(F) * 7001 + 3, noise=0.05)
    for _ in range(3):
        idx, sim = m.top1(q)
    assert np.array_equal(idx, np.arange(F) * 7001 + 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        idx, sim = m.top1(q)
    print("queries %3d: top1 %.3f ms per call (host in, host out)" % (F, (time.perf_counter() - t0) * 50), flush=True)
