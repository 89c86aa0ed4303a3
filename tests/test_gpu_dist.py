"""Multi-GPU paths over the `nccl` backend (= RCCL on ROCm) on ONE GPU: a 1-rank process group still loads RCCL, creates the
communicator and runs the collectives on the device, so the code bench.py / dist.py run at N > 1 is exercised end to end here (the
world-size-2 logic is covered on CPU by tests/test_distributed.py).  Runs in a subprocess: a process group can only be initialised
once per process."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
frt = load_pkg()
from frt_amd import dist as fd
from oracle import match

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = frt.synth
main = torch.cuda.Stream()
torch.cuda.set_stream(main)

# ---- config 4: result records all-gathered on a side stream
rec = np.zeros(8, frt.RESULT_DTYPE)
rec["match_idx"] = np.arange(8) * 7
rec["valid"] = 1
d = torch.from_numpy(rec.view(np.uint8).reshape(8, -1).copy()).cuda()
side = torch.cuda.Stream()
ev = torch.cuda.Event(); ev.record(main)
with torch.cuda.stream(side):
    side.wait_event(ev)
    allr = fd.all_gather_results(d)
side.synchronize()
assert np.array_equal(allr.cpu().numpy().reshape(-1).view(frt.RESULT_DTYPE), rec)

# ---- config 5: fp16-stored gallery shard with a row offset, embeddings all-gathered, device top-1, winners all-gathered + merged
N, off = 70000, 1000000
gal = s.make_gallery(N)
gal[N - 5] = gal[17]                       # duplicate rows: the lower global index must win
g16 = gal.astype(np.float16).astype(np.float32)
q = gal[[17, 40000, 69999]] + 0.01
q[0] = g16[17]
mm = frt.MatMul(0)
mm.setStorage(True)
mm.init(gal)
mm.setRowOffset(off)
q_all = fd.all_gather_embeddings(torch.from_numpy(q).cuda())
idx = torch.zeros(3, dtype=torch.int32, device="cuda"); sim = torch.zeros(3, device="cuda")
mm.top1_dev(q_all.data_ptr(), 3, idx.data_ptr(), sim.data_ptr(), main.cuda_stream)
gi, gs = fd.sharded_top1(idx, sim)
torch.cuda.synchronize()
oi, osim = match.top1(q, g16)
assert gi.cpu().tolist() == [int(i) + off for i in oi] and gi.cpu().tolist()[0] == 17 + off, (gi, oi)
assert np.abs(gs.cpu().numpy() - osim).max() < 1e-5
mm.close()
dist.barrier()
dist.destroy_process_group()
print("DIST_OK")
'''


def test_rccl_paths_on_one_gpu(tmp_path):
    p = tmp_path / "dist_nccl.py"
    p.write_text("ROOT = %r\n" % ROOT + SCRIPT)
    out = subprocess.run([sys.executable, str(p)], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))
    assert out.returncode == 0 and "DIST_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_bench_dist_modes_run(tmp_path):
    """bench.py's N > 1 code (per-step gather on a side stream; sharded fp16 gallery) on a forced 1-rank RCCL group, tiny sizes."""
    import json
    for extra in ([], ["--sharded-gallery"], ["--strong"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--batch", "4", "--gallery", "40000",
                              "--no-cpu-baseline", "--no-extras"] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT,
                             env=dict(os.environ, FRT_BENCH_FORCE_DIST="1", MASTER_PORT="29534"))
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        line = next(l for l in out.stdout.splitlines() if l.startswith("{"))
        d = json.loads(line)
        assert d["value"] > 0 and d["config"]["faces_per_step"] == 16, d
