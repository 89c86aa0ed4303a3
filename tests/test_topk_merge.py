"""Top-k list merge of the sharded-gallery path (BASELINE configs[4]: "RCCL top-k all-gather"): the C-ABI host merge (frt_merge_topk) against
the oracle's rule - higher similarity first, LOWER global index among equal similarities (std::max_element carried across shards,
/root/reference/src/arcface.cpp:203-217), empty slots (-1) skipped.  CPU only: no device call."""
import numpy as np
import pytest


def random_lists(r, shards, n, k, ties):
    """per-shard sorted lists with global indices, some empty tails, optional cross-shard ties"""
    idx = np.full((shards, n, k), -1, np.int32)
    sim = np.full((shards, n, k), -np.inf, np.float32)
    for s in range(shards):
        for q in range(n):
            m = int(r.integers(0, k + 1))
            v = r.choice(np.arange(-8, 9, dtype=np.float32) / 8, size=m) if ties else r.standard_normal(m).astype(np.float32)
            i = r.choice(np.arange(s * 1000, (s + 1) * 1000), size=m, replace=False)
            order = np.lexsort((i, -v))
            idx[s, q, :m] = i[order]
            sim[s, q, :m] = v[order]
    return idx, sim


@pytest.mark.parametrize("shards,n,k,ties", [(1, 5, 1, False), (2, 33, 5, True), (8, 64, 5, True), (8, 7, 16, False), (3, 1, 4, True)])
def test_host_merge_matches_oracle(frt, shards, n, k, ties):
    from oracle import match
    r = np.random.default_rng(shards * 100 + n + k)
    idx, sim = random_lists(r, shards, n, k, ties)
    gi, gs = frt.merge_topk(idx, sim)
    oi, osim = match.merge_topk(idx, sim)
    assert np.array_equal(gi, oi)
    assert np.array_equal(gs, osim)


def test_merge_of_shard_lists_equals_the_whole_gallery_ranking(frt, synth):
    from oracle import match
    g = synth.make_gallery(3000)
    g[2500] = g[7]      # duplicates in different shards: the lower global index comes first
    g[1000] = g[999]    # ... across a shard boundary
    q = np.concatenate([g[[7, 999]], synth.make_queries(g, [1234, 2999])])
    k = 5
    whole = match.topk(q, g, k)
    parts = [match.topk(q, g[b:e], k, row_offset=b) for b, e in ((0, 1000), (1000, 2000), (2000, 3000))]
    gi, gs = frt.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert np.array_equal(gi, whole[0]) and np.array_equal(gs, whole[1])
    assert gi[0, :2].tolist() == [7, 2500] and gi[1, :2].tolist() == [999, 1000]


def test_k1_merge_is_merge_top1(frt):
    r = np.random.default_rng(5)
    idx, sim = random_lists(r, 2, 50, 1, True)
    gi, gs = frt.merge_topk(idx, sim)
    i1, s1 = frt.merge_top1(idx[0, :, 0], sim[0, :, 0], idx[1, :, 0], sim[1, :, 0])
    both_empty = (idx[0, :, 0] < 0) & (idx[1, :, 0] < 0)
    assert np.array_equal(gi[:, 0][~both_empty], i1[~both_empty]) and np.array_equal(gs[:, 0][~both_empty], s1[~both_empty])
