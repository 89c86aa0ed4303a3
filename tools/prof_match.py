"""Run only the matcher (1M x 512 gallery, 128 queries, a few calls) - a short target for rocprofv3.  argv[1] = "miss": queries that match
nothing (the benchmark's case: embeddings of synthetic frames against a random gallery); default: planted queries."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

import __graft_entry__ as entry

frt = entry.load_pkg()
gal = frt.synth.make_gallery(1000000)
if len(sys.argv) > 1 and sys.argv[1] == "miss":
    q = np.random.default_rng(5).standard_normal((128, 512)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
else:
    q = frt.synth.make_queries(gal, np.arange(128) * 7001 + 3, noise=0.05)
m = frt.MatMul(0)
m.init(gal)
for _ in range(3):
    idx, sim = m.top1(q)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    idx, sim = m.top1(q)
print("top1 ms", (time.perf_counter() - t0) * 100, idx[:4], sim[:4])
