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
