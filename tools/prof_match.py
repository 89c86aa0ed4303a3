import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
import numpy as np, torch
frt = entry.load_pkg()
gal = frt.synth.make_gallery(1000000)
q = frt.synth.make_queries(gal, np.arange(128) * 7001 + 3, noise=0.05)
m = frt.MatMul(0); m.init(gal)
for _ in range(3): idx, sim = m.top1(q)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): idx, sim = m.top1(q)
print("top1 ms", (time.perf_counter() - t0) * 100, idx[:4], sim[:4])
