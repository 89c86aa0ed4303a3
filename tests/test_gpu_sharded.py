"""BASELINE configs[4] at its PER-RANK shape (round-2 VERDICT, parity item 2): "640x640 batch=64, 10M x 512 gallery sharded across 8 GPUs,
fp16 embeddings, RCCL top-k all-gather".  At N = 8 every rank searches the all-gathered queries of the whole node against ITS 1.25M-row
fp16-stored shard with global row indices.  Here one GPU plays one rank: the screened search (fp16 coarse scan + exact re-rank) with
F in {129, 256, 300} queries - more than one 128-query block, a ragged last block - a row offset, duplicate rows inside and across
128-row tiles, fp16 queries, and top-k lists; all against oracle/match.py on the fp16-rounded rows (bit-identical indices, |dsim| < 1e-5).
Reference semantics: src/matmul.h:7-16 (S = E * G^T in fp32), src/arcface.cpp:203-217 (first maximum)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_SHARD = 1_250_000
OFFSET = 3 * N_SHARD  # rank 3 of 8
SIM_TOL = 1e-5


@pytest.fixture(scope="module")
def shard(frt, synth):
    """fp16-representable rows (what an fp16-stored shard holds) with planted duplicates; one matcher for the module."""
    g = synth.make_gallery(N_SHARD)
    for s in range(0, N_SHARD, 1 << 16):  # round to fp16 in place, chunked (2.5 GB array)
        g[s:s + (1 << 16)] = g[s:s + (1 << 16)].astype(np.float16).astype(np.float32)
    # duplicates: inside one 32-row block, across blocks of one 128-row tile, across neighbouring tiles, far apart, last row
    for dst, src in ((130, 129), (200, 140), (128 * 77, 128 * 77 - 1), (1_000_003, 5), (N_SHARD - 1, 640_000)):
        g[dst] = g[src]
    mm = frt.MatMul(0)
    mm.setStorage(True)
    mm.init(g)
    mm.setRowOffset(OFFSET)
    yield g, mm
    mm.close()


def queries(synth, g, F, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    idx = r.integers(0, N_SHARD, F)
    idx[:10] = [129, 130, 140, 200, 128 * 77 - 1, 128 * 77, 5, 1_000_003, 640_000, N_SHARD - 1]  # the duplicated rows, both copies
    q = synth.make_queries(g, idx, noise=0.02, seed=seed)
    q[:10] = g[idx[:10]]  # exact copies: bit-identical similarities for the pair, the lower index must win
    return q, idx


@pytest.mark.parametrize("F", [129, 256, 300])
def test_screened_top1_on_a_fp16_shard_matches_the_oracle(frt, synth, shard, F):
    from oracle import match
    g, mm = shard
    q, idx = queries(synth, g, F, 100 + F)
    got_i, got_s = mm.top1(q)
    want_i, want_s = match.top1(q, g)
    assert np.array_equal(got_i, want_i + OFFSET)
    assert np.abs(got_s - want_s).max() < SIM_TOL
    # the planted duplicates resolve to the lower index
    assert got_i[:10].tolist() == [OFFSET + v for v in (129, 129, 140, 140, 128 * 77 - 1, 128 * 77 - 1, 5, 5, 640_000, 640_000)]
    # bit-identical to the unscreened exact scan of the same stored rows (full matrix of a few queries)
    full = mm.calculate(q[:12])
    assert np.array_equal(full.argmax(1) + OFFSET, got_i[:12]) and np.array_equal(full.max(1), got_s[:12])


def test_fp16_queries_and_topk_lists_on_the_device_match_the_oracle(frt, synth, shard):
    """The exchange format of configs[4]: embeddings travel as fp16, every rank answers with top-k lists (global indices)."""
    import torch
    from oracle import match
    g, mm = shard
    F, k = 256, 5
    q, idx = queries(synth, g, F, 7)
    dq = torch.from_numpy(q).cuda()
    dq16 = torch.empty(F, 512, dtype=torch.float16, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        frt.embeds_to_half_dev(dq.data_ptr(), F * 512, dq16.data_ptr(), st.cuda_stream)
        di = torch.full((F, k), -7, dtype=torch.int32, device="cuda")
        ds = torch.zeros(F, k, device="cuda")
        mm.topk_dev(dq16.data_ptr(), F, k, di.data_ptr(), ds.data_ptr(), st.cuda_stream, fp16=True)
    st.synchronize()
    q16 = q.astype(np.float16)
    assert np.array_equal(dq16.cpu().numpy(), q16)                      # round-to-nearest-even, like NumPy
    want_i, want_s = match.topk(q16.astype(np.float32), g, k, row_offset=OFFSET)
    assert np.array_equal(di.cpu().numpy(), want_i)
    assert np.abs(ds.cpu().numpy() - want_s).max() < SIM_TOL
    # entry 0 of the list IS the top-1 answer for the same (fp16-rounded) queries, bit for bit
    i1, s1 = mm.top1(q16.astype(np.float32))
    assert np.array_equal(i1, di.cpu().numpy()[:, 0]) and np.array_equal(s1, ds.cpu().numpy()[:, 0])
    # the host entry point agrees
    hi, hs = mm.topk(q16.astype(np.float32)[:40], k)
    assert np.array_equal(hi, want_i[:40]) and np.array_equal(hs, ds.cpu().numpy()[:40])
    # duplicates sit next to each other in the list, lower index first
    assert di.cpu().numpy()[0, :2].tolist() == [OFFSET + 129, OFFSET + 130]
    assert di.cpu().numpy()[6, :2].tolist() == [OFFSET + 5, OFFSET + 1_000_003]


@pytest.mark.parametrize("N,F,k,fp16", [(3, 2, 5, False), (100, 40, 16, False), (40000, 70, 4, True), (70000, 33, 1, True)])
def test_topk_small_and_ragged_galleries(frt, synth, N, F, k, fp16):
    """fewer rows than k (empty slots), unscreened (N < 32768) and screened galleries, fp32- and fp16-stored."""
    from oracle import match
    g = synth.make_gallery(N)
    if N > 10:
        g[N - 1] = g[1]
    if fp16:
        g = g.astype(np.float16).astype(np.float32)
    q = synth.make_queries(g, np.arange(F) % N, noise=0.05)
    q[0] = g[1 % N]
    mm = frt.MatMul(0)
    mm.setStorage(fp16)
    mm.init(g)
    gi, gs = mm.topk(q, k)
    wi, ws = match.topk(q, g, k)
    assert np.array_equal(gi, wi)
    fin = np.isfinite(ws)
    assert np.array_equal(np.isfinite(gs), fin) and np.abs(gs[fin] - ws[fin]).max() < SIM_TOL
    assert np.array_equal(gi[:, 0], mm.top1(q)[0])
    mm.close()


def test_two_shards_merged_on_the_device_equal_the_whole_gallery(frt, synth):
    """per-shard top-k with global indices + frt_merge_topk_dev == top-k over the unsharded gallery (duplicates across shards included)."""
    import torch
    from oracle import match
    N, F, k = 90000, 130, 5
    g = synth.make_gallery(N).astype(np.float16).astype(np.float32)
    g[70000] = g[11]
    g[45000] = g[44999]  # across the shard boundary
    q = synth.make_queries(g, np.arange(F) * 601 % N, noise=0.03)
    q[0], q[1] = g[11], g[44999]
    dq = torch.from_numpy(q).cuda()
    lists_i = torch.zeros(2, F, k, dtype=torch.int32, device="cuda")
    lists_s = torch.zeros(2, F, k, device="cuda")
    mms = []
    for r, (b, e) in enumerate(((0, 45000), (45000, N))):
        mm = frt.MatMul(0)
        mm.setStorage(True)
        mm.init(g[b:e])
        mm.setRowOffset(b)
        mm.topk_dev(dq.data_ptr(), F, k, lists_i[r].data_ptr(), lists_s[r].data_ptr(), torch.cuda.current_stream().cuda_stream)
        mms.append(mm)
    oi = torch.zeros(F, k, dtype=torch.int32, device="cuda")
    os_ = torch.zeros(F, k, device="cuda")
    frt.merge_topk_dev(2, F, k, lists_i.data_ptr(), lists_s.data_ptr(), oi.data_ptr(), os_.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    wi, ws = match.topk(q, g, k)
    assert np.array_equal(oi.cpu().numpy(), wi) and np.abs(os_.cpu().numpy() - ws).max() < SIM_TOL
    assert wi[0, :2].tolist() == [11, 70000] and wi[1, :2].tolist() == [44999, 45000]
    # host merge == device merge
    hi, hs = frt.merge_topk(lists_i.cpu().numpy(), lists_s.cpu().numpy())
    assert np.array_equal(hi, oi.cpu().numpy()) and np.array_equal(hs, os_.cpu().numpy())
    for mm in mms:
        mm.close()


def test_topk_pair_list_overflow_falls_back_to_the_exact_scan(frt, synth):
    """top-k on a gallery of identical rows: every tile qualifies for every query, the pair list overflows and each of the k passes is answered
    by the unscreened exact scan with the exclusion rule - the list is rows 0, 1, 2 with identical similarities."""
    N, k = 45000, 3
    g = np.tile(synth.make_gallery(1), (N, 1))
    q = np.concatenate([g[:1], synth.make_gallery(39, seed=8)])
    m = frt.MatMul(0)
    m.init(g)
    i, s = m.topk(q, k)
    assert np.array_equal(i, np.tile(np.arange(k, dtype=np.int32), (40, 1)))
    assert np.array_equal(s[:, 0], s[:, 1]) and np.array_equal(s[:, 1], s[:, 2]) and np.array_equal(s[:, 0], m.top1(q)[1])
    m.close()


def test_topk_with_two_thousand_queries_on_the_shard(frt, synth, shard):
    """configs[4] at N = 8: every rank searches the 8 x 256 all-gathered queries of the node against its shard."""
    from oracle import match
    g, mm = shard
    F, k = 2048, 5
    q, idx = queries(synth, g, F, 11)
    gi, gs = mm.topk(q, k)
    pick = np.concatenate([np.arange(12), np.arange(1000, 1040), np.arange(F - 8, F)])  # the oracle on a sample (it is O(F N) on the host)
    wi, ws = match.topk(q[pick], g, k, row_offset=OFFSET)
    assert np.array_equal(gi[pick], wi) and np.abs(gs[pick] - ws).max() < SIM_TOL
    i1, s1 = mm.top1(q)
    assert np.array_equal(gi[:, 0], i1) and np.array_equal(gs[:, 0], s1)      # entry 0 is the top-1 answer, bit for bit
    assert np.array_equal(gi[10:, 0], idx[10:] + OFFSET)                       # planted rows win (the first ten are the duplicated ones)
    assert (np.diff(gs, axis=1) <= 0).all()                                    # lists are sorted
