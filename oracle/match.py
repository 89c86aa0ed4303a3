"""ORACLE (test infrastructure only): cosine-similarity matrix + top-1.

``MatMul::calculate`` (``/root/reference/src/matmul.h:7-16``, ``matmul.cpp:36-77``): ``out[i*m + j] = sum_k B[i,k]*A[j,k]``
in fp32 (cuBLASLt, ``CUBLAS_COMPUTE_32F``, closed third-party binary from CUDA 11.3 -> PARITY UNPINNED, semantics fully
specified by the header comment).  ``ArcFaceIR50::getOutputs`` (``arcface.cpp:203-217``): per row ``std::max_element``
= FIRST maximum on ties.  NumPy ``argmax`` has the same first-maximum rule.
"""
import numpy as np


def similarity(embeds, gallery):
    return np.ascontiguousarray(embeds, np.float32) @ np.ascontiguousarray(gallery, np.float32).T


def top1(embeds, gallery, chunk=1 << 18):
    """(idx int32 [F], sim float32 [F]) with lowest-index tie-break, streaming the gallery in chunks."""
    e = np.ascontiguousarray(embeds, np.float32)
    best = np.full(e.shape[0], -np.inf, np.float32)
    arg = np.zeros(e.shape[0], np.int64)
    for s in range(0, gallery.shape[0], chunk):
        sim = e @ gallery[s:s + chunk].T
        a = sim.argmax(1)
        v = sim[np.arange(e.shape[0]), a]
        upd = v > best  # strict: an equal value later in the gallery never displaces an earlier one
        best[upd] = v[upd]
        arg[upd] = a[upd] + s
    return arg.astype(np.int32), best


def topk(embeds, gallery, k, row_offset=0, chunk=1 << 17):
    """(idx int32 [F, k], sim float32 [F, k]): the first k entries of every query's exact ranking - higher similarity first, LOWER
    index first among equal similarities, i.e. ``std::max_element`` (``arcface.cpp:210``) applied k times to the rows not yet taken.
    Slots beyond the gallery size hold -1 / -inf.  Indices are global (``+ row_offset``).  Streams the gallery in chunks; per chunk the
    candidates are all entries >= the chunk's k-th largest value (every tie included), so no lower-index duplicate is ever dropped."""
    e = np.ascontiguousarray(embeds, np.float32)
    F = e.shape[0]
    cand_v = [np.empty(0, np.float32) for _ in range(F)]
    cand_i = [np.empty(0, np.int64) for _ in range(F)]
    for s in range(0, gallery.shape[0], chunk):
        sim = e @ np.ascontiguousarray(gallery[s:s + chunk], np.float32).T
        n = sim.shape[1]
        kk = min(k, n)
        kth = np.partition(sim, n - kk, axis=1)[:, n - kk]
        for f in range(F):
            j = np.nonzero(sim[f] >= kth[f])[0]
            v = np.concatenate([cand_v[f], sim[f, j]])
            i = np.concatenate([cand_i[f], j + s])
            order = np.lexsort((i, -v))[:k]
            cand_v[f], cand_i[f] = v[order], i[order]
    out_i = np.full((F, k), -1, np.int32)
    out_v = np.full((F, k), -np.inf, np.float32)
    for f in range(F):
        m = len(cand_i[f])
        out_i[f, :m] = cand_i[f] + row_offset
        out_v[f, :m] = cand_v[f]
    return out_i, out_v


def merge_topk(idx_all, sim_all):
    """Reference merge of per-shard lists [shards, n, k] (global indices, -1 = empty) -> ([n, k], [n, k]), same order rule."""
    idx_all = np.asarray(idx_all, np.int64)
    sim_all = np.asarray(sim_all, np.float32)
    shards, n, k = idx_all.shape
    iv = idx_all.transpose(1, 0, 2).reshape(n, shards * k)
    sv = sim_all.transpose(1, 0, 2).reshape(n, shards * k)
    key_i = np.where(iv < 0, np.iinfo(np.int64).max, iv)
    key_s = np.where(iv < 0, -np.inf, sv)
    order = np.lexsort((key_i, -key_s), axis=1)[:, :k]
    oi = np.take_along_axis(iv, order, 1)
    os_ = np.take_along_axis(key_s, order, 1)
    return np.where(oi >= 0, oi, -1).astype(np.int32), os_.astype(np.float32)
