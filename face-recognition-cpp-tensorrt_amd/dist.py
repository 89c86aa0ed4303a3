"""Multi-GPU layer of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm).

The reference is single-GPU; frames are independent units, so the path shards by frames with NO collective
on the data path.  Two exchange patterns exist, both tiny (KiB) and latency-bound:

* replicated gallery (BASELINE config 4): every rank detects/embeds/matches its own frames against its own full gallery
  replica; one ``all_gather`` of the per-face result records (36 B each) so every rank holds the whole batch's answer.
* sharded gallery (config 5): ``all_gather`` the embeddings, every rank computes top-1 over ITS gallery rows with GLOBAL
  row indices, ``all_gather`` the (idx, sim) pairs, merge with "higher similarity, then LOWER global index" - the
  ``std::max_element`` first-maximum rule of ``ArcFaceIR50::getOutputs`` (/root/reference/src/arcface.cpp:210).

Everything here works on CPU tensors with the ``gloo`` backend too (tests/test_distributed.py, world_size 2).

Since round 3 the DATA-PATH exchanges of ``bench.py`` no longer go through ``torch.distributed``: they are ``ncclAllGather`` calls made by
``libfrt.so`` itself (``frt_comm_*`` in ``include/frt.h``, :class:`CommGroup` below) - the path a C++ host uses.  ``torch.distributed`` only
carries the 128-byte communicator id to the ranks and the benchmark's barrier / max-over-ranks bookkeeping.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [begin, end) of ``n_items`` for ``rank`` (first ``n_items % world`` ranks get one extra)."""
    base, extra = divmod(int(n_items), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gallery_shard(n_rows, rank, world):
    """Row range of a gallery sharded across ranks; the begin offset is the shard's global row offset."""
    return shard_range(n_rows, rank, world)


def all_gather_results(local_results, group=None):
    """all_gather of fixed-size per-face result records (uint8 view of RESULT_DTYPE rows) -> [world * F] records."""
    world = dist.get_world_size(group)
    out = torch.empty((world * local_results.shape[0],) + tuple(local_results.shape[1:]), dtype=local_results.dtype,
                      device=local_results.device)
    dist.all_gather_into_tensor(out, local_results.contiguous(), group=group)
    return out


def all_gather_embeddings(local_emb, group=None):
    """[F_local, D] -> [world * F_local, D] (every rank contributes the same F_local; pad unused face slots with zeros)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * local_emb.shape[0], local_emb.shape[1]), dtype=local_emb.dtype, device=local_emb.device)
    dist.all_gather_into_tensor(out, local_emb.contiguous(), group=group)
    return out


def merge_top1(idx, sim):
    """idx, sim: [world, F] per-shard winners with GLOBAL indices (-1 = shard had no rows) -> ([F], [F]).

    Higher similarity wins; on equal similarity the LOWER global index wins (first maximum)."""
    idx = torch.as_tensor(idx)
    sim = torch.as_tensor(sim)
    valid = idx >= 0
    s = torch.where(valid, sim, torch.full_like(sim, float("-inf")))
    best = s.max(dim=0).values
    big = torch.iinfo(idx.dtype).max
    cand = torch.where(valid & (s == best.unsqueeze(0)), idx, torch.full_like(idx, big))
    out_idx = cand.min(dim=0).values
    none = out_idx == big
    out_idx = torch.where(none, torch.full_like(out_idx, -1), out_idx)
    out_sim = torch.where(none, torch.zeros_like(best), best)
    return out_idx, out_sim


def sharded_top1(local_idx, local_sim, group=None):
    """all_gather the per-shard (global idx, sim) winners and merge; every rank returns the same answer."""
    world = dist.get_world_size(group)
    F = local_idx.shape[0]
    gi = torch.empty((world * F,), dtype=local_idx.dtype, device=local_idx.device)
    gs = torch.empty((world * F,), dtype=local_sim.dtype, device=local_sim.device)
    dist.all_gather_into_tensor(gi, local_idx.contiguous(), group=group)
    dist.all_gather_into_tensor(gs, local_sim.contiguous(), group=group)
    return merge_top1(gi.view(world, F), gs.view(world, F))


def numpy_top1(emb, gallery_rows, row_offset):
    """Host stand-in for the device matcher in CPU tests: exact fp32 dot products, first maximum, global indices."""
    if gallery_rows.shape[0] == 0:
        return np.full(emb.shape[0], -1, np.int32), np.zeros(emb.shape[0], np.float32)
    s = emb.astype(np.float32) @ gallery_rows.astype(np.float32).T
    a = s.argmax(1)
    return (a + row_offset).astype(np.int32), s[np.arange(len(a)), a].astype(np.float32)


def numpy_topk(emb, gallery_rows, row_offset, k):
    """Host stand-in for the device matcher's top-k in CPU tests (same order rule: higher similarity, then LOWER global index)."""
    n = emb.shape[0]
    idx = np.full((n, k), -1, np.int32)
    sim = np.full((n, k), -np.inf, np.float32)
    if gallery_rows.shape[0] == 0:
        return idx, sim
    s = emb.astype(np.float32) @ gallery_rows.astype(np.float32).T
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    kk = order.shape[1]
    idx[:, :kk] = order + row_offset
    sim[:, :kk] = np.take_along_axis(s, order, 1)
    return idx, sim


def sharded_topk(local_idx, local_sim, merge, group=None):
    """all_gather the per-shard top-k lists [F, k] (global indices) over ``torch.distributed`` and merge them with ``merge`` (the C-ABI
    ``frt_merge_topk`` through ``frt_amd.merge_topk``); every rank returns the same ([F, k], [F, k])."""
    world = dist.get_world_size(group)
    F, k = local_idx.shape
    gi = torch.empty((world * F, k), dtype=local_idx.dtype, device=local_idx.device)
    gs = torch.empty((world * F, k), dtype=local_sim.dtype, device=local_sim.device)
    dist.all_gather_into_tensor(gi, local_idx.contiguous(), group=group)
    dist.all_gather_into_tensor(gs, local_sim.contiguous(), group=group)
    return merge(gi.view(world, F, k).cpu().numpy(), gs.view(world, F, k).cpu().numpy())


class CommGroup:
    """One RCCL communicator per rank created through the C ABI (``frt_comm_create``).  The 128-byte id is made on rank 0 and handed to the
    other ranks by ``broadcast`` (any callable ``bytes | None -> bytes``; bench.py passes a torch.distributed object broadcast, a C++ host
    would use its own channel).  World size 1 needs no channel."""

    def __init__(self, frt, rank, world, device, broadcast=None):
        uid = frt.comm_unique_id() if rank == 0 else None
        if world > 1:
            if broadcast is None:
                raise ValueError("CommGroup: world > 1 needs a broadcast callable for the communicator id")
            uid = broadcast(uid)
        self.frt, self.rank, self.world = frt, rank, world
        self.comm = frt.Comm(uid, rank, world, device)

    @property
    def stream(self):
        return self.comm.stream

    def all_gather(self, send, recv, hip_stream=None):
        """send: device tensor of this rank; recv: device tensor of world x send.nbytes; asynchronous on ``hip_stream`` (default: the comm's)."""
        nbytes = send.numel() * send.element_size()
        assert recv.numel() * recv.element_size() == nbytes * self.world
        self.comm.all_gather(send.data_ptr(), recv.data_ptr(), nbytes, hip_stream)

    def close(self):
        self.comm.close()
