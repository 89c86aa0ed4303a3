"""N > 1 path on CPU: world_size-2 gloo processes exercise the frame sharding, the result all-gather and the sharded-gallery
top-1 merge (same code bench.py / a multi-GPU caller runs over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frt = load_pkg()
    from frt_amd import dist as fd
    s = frt.synth
    # --- config 4: frames sharded, results all-gathered
    n_frames, K = 7, 4
    b, e = fd.shard_range(n_frames, rank, world)
    rec = np.zeros((4 * K,), frt.RESULT_DTYPE)  # fixed capacity per rank (ceil(7/2) frames)
    for i, f in enumerate(range(b, e)):
        for k in range(K):
            rec[i * K + k] = (f, k, f + 10, k + 10, 0.5 + 0.01 * k, f, 100 * f + k, 0.9, 1)
    loc = torch.from_numpy(rec.view(np.uint8).reshape(len(rec), -1).copy())
    allr = fd.all_gather_results(loc).numpy().reshape(-1).view(frt.RESULT_DTYPE)
    valid = allr[allr["valid"] == 1]
    ok1 = sorted(valid["match_idx"].tolist()) == sorted(100 * f + k for f in range(n_frames) for k in range(K))
    # --- config 5: gallery sharded, embeddings gathered, top-1 merged with the first-maximum rule
    N = 1001
    gal = s.make_gallery(N)
    gal[900] = gal[3]      # duplicate rows in different shards: global index 3 must win
    gal[499] = gal[500]    # duplicates across the shard boundary
    q_local = s.make_queries(gal, [3, 900] if rank == 0 else [500, 42], noise=0.0 if rank == 0 else 0.01)
    q_all = fd.all_gather_embeddings(torch.from_numpy(q_local)).numpy()
    gb, ge = fd.gallery_shard(N, rank, world)
    li, ls = fd.numpy_top1(q_all, gal[gb:ge], gb)
    gi, gs = fd.sharded_top1(torch.from_numpy(li), torch.from_numpy(ls))
    full_i, full_s = fd.numpy_top1(q_all, gal, 0)
    ok2 = gi.tolist() == full_i.tolist() == [3, 3, 499, 42] and np.allclose(gs.numpy(), full_s, atol=1e-6)
    # the C-ABI merge agrees with the torch merge
    other = [torch.empty_like(torch.from_numpy(li)) for _ in range(world)]
    others = [torch.empty_like(torch.from_numpy(ls)) for _ in range(world)]
    dist.all_gather(other, torch.from_numpy(li))
    dist.all_gather(others, torch.from_numpy(ls))
    mi, ms = frt.merge_top1(other[0].numpy(), others[0].numpy(), other[1].numpy(), others[1].numpy())
    ok3 = mi.tolist() == gi.tolist() and np.array_equal(ms, gs.numpy())
    # --- configs[4] as written: fp16 embeddings exchanged, per-shard top-k lists gathered and merged (frt_merge_topk) == the ranking over
    #     the whole gallery; duplicates in different shards come out "lower global index first"
    from oracle import match
    k = 5
    q16 = q_all.astype(np.float16).astype(np.float32)             # what travels
    ti, ts = fd.numpy_topk(q16, gal[gb:ge], gb, k)
    mi, ms = fd.sharded_topk(torch.from_numpy(ti), torch.from_numpy(ts), frt.merge_topk)
    wi, ws = match.topk(q16, gal, k)
    ok4 = np.array_equal(mi, wi) and np.allclose(ms, ws, atol=1e-6) and mi[0, :2].tolist() == [3, 900] and mi[2, :2].tolist() == [499, 500]
    q.put((rank, ok1, ok2, ok3 and ok4))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything(frt):
    from frt_amd import dist as fd
    for n in (0, 1, 7, 32, 1000003):
        for w in (1, 2, 3, 8):
            r = [fd.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def test_merge_handles_empty_shards(frt):
    from frt_amd import dist as fd
    i, s = fd.merge_top1(torch.tensor([[-1, 5, -1], [7, 2, -1]]), torch.tensor([[0.0, 0.4, 0.0], [0.3, 0.4, 0.0]]))
    assert i.tolist() == [7, 2, -1] and np.allclose(s.numpy(), [0.3, 0.4, 0.0])


def test_world_size_2_gloo(frt):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res


def _bench(args, env_extra, timeout=300):
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` as the driver types it (no torch.distributed.run around it) must start two ranks: under the launch-check
    stub (no device needed) both join a gloo group and rank 0 reports the number of ranks that really took part (round-3 review item 1:
    it used to run ONE rank and print n_gpus 1)."""
    import json
    o = _bench(["--gpus", "2", "--steps", "2"], {"FRT_BENCH_LAUNCH_CHECK": "1"})
    assert o.returncode == 0, o.stderr[-2000:]
    lines = [l for l in o.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, o.stdout          # ONE JSON line, from rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["local_rank_sum"] == 1   # LOCAL_RANK 0 and 1: one device each


def test_bench_refuses_more_gpus_than_devices():
    """Without devices (this container) `--gpus 2` exits non-zero with a message that says why; so does a rank count that contradicts --gpus."""
    o = _bench(["--gpus", "2", "--steps", "2"], {})
    assert o.returncode != 0 and "refusing to run" in o.stderr and not o.stdout.strip()
    o = _bench(["--gpus", "4", "--steps", "2"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert o.returncode != 0 and "WORLD_SIZE" in o.stderr
