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

# ---- the same exchanges through libfrt's own RCCL binding (frt_comm_*: ncclAllGather from C++, no torch.distributed on the data path)
cg = fd.CommGroup(frt, 0, 1, 0)
side = torch.cuda.ExternalStream(cg.stream)
recv = torch.zeros_like(d)
ev2 = torch.cuda.Event(); ev2.record(main)
with torch.cuda.stream(side):
    side.wait_event(ev2)
    cg.all_gather(d, recv)                      # on the communicator's own stream
side.synchronize()
assert torch.equal(recv, d)
N2, off2, k = 70000, 2 * 70000, 5
g2 = s.make_gallery(N2, seed=9).astype(np.float16).astype(np.float32)
g2[N2 - 3] = g2[21]
q2 = s.make_queries(g2, [21, 5000, N2 - 1, 33333], noise=0.02)
q2[0] = g2[21]
mm = frt.MatMul(0)
mm.setStorage(True)
mm.init(g2)
mm.setRowOffset(off2)
dq = torch.from_numpy(q2).cuda()
dq16 = torch.empty(4, 512, dtype=torch.float16, device="cuda")
qall = torch.empty(4, 512, dtype=torch.float16, device="cuda")
li = torch.zeros(4, k, dtype=torch.int32, device="cuda"); ls = torch.zeros(4, k, device="cuda")
gi = torch.zeros(1, 4, k, dtype=torch.int32, device="cuda"); gs = torch.zeros(1, 4, k, device="cuda")
fi = torch.zeros(4, k, dtype=torch.int32, device="cuda"); fs = torch.zeros(4, k, device="cuda")
cs = main.cuda_stream
frt.embeds_to_half_dev(dq.data_ptr(), 4 * 512, dq16.data_ptr(), cs)
cg.all_gather(dq16, qall, cs)                   # exchange 1: fp16 embeddings
mm.topk_dev(qall.data_ptr(), 4, k, li.data_ptr(), ls.data_ptr(), cs, fp16=True)
cg.all_gather(li, gi, cs)                       # exchange 2: the top-k lists
cg.all_gather(ls, gs, cs)
frt.merge_topk_dev(1, 4, k, gi.data_ptr(), gs.data_ptr(), fi.data_ptr(), fs.data_ptr(), cs)
torch.cuda.synchronize()
wi, ws = match.topk(q2.astype(np.float16).astype(np.float32), g2, k, row_offset=off2)
assert np.array_equal(fi.cpu().numpy(), wi), (fi, wi)
assert np.abs(fs.cpu().numpy() - ws).max() < 1e-5
assert fi.cpu().numpy()[0, :2].tolist() == [off2 + 21, off2 + N2 - 3]
mm.close()
cg.close()
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


def test_bench_dist_modes_run(tmp_path, synth):
    """bench.py's N > 1 code (per-step gather through frt_comm on its own stream; sharded fp16 gallery with fp16 embedding exchange and
    top-k lists) on a forced 1-rank RCCL group, tiny sizes.  The sharded mode's ANSWERS are checked against the oracle: the dump holds
    the last step's exchanged fp16 queries and the merged lists."""
    import json

    import numpy as np
    from oracle import match
    dump = str(tmp_path / "final.npz")
    for extra in ([], ["--sharded-gallery", "--topk", "5", "--dump-final", dump], ["--strong"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--batch", "4", "--gallery", "40000",
                              "--no-cpu-baseline", "--no-extras"] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT,
                             env=dict(os.environ, FRT_BENCH_FORCE_DIST="1", MASTER_PORT="29534"))
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        line = next(l for l in out.stdout.splitlines() if l.startswith("{"))
        d = json.loads(line)
        assert d["value"] > 0 and d["config"]["faces_per_step"] == 16, d
    z = np.load(dump)
    assert int(z["world"]) == 1 and int(z["k"]) == 5 and z["queries_f16"].dtype == np.float16 and z["queries_f16"].shape == (16, 512)
    g16 = synth.make_gallery(int(z["gallery_rows"]), seed=int(z["gallery_seed"])).astype(np.float16).astype(np.float32)  # the fp16-stored shard
    q = z["queries_f16"].astype(np.float32)
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 2e-3                  # real embeddings went through the exchange
    wi, ws = match.topk(q, g16, 5)
    assert np.array_equal(z["idx"], wi)
    assert np.abs(z["sim"] - ws).max() < 1e-5
