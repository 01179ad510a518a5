"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), a contiguous block of trees per rank, X replicated.

Trees are independent (createMulti maps over the IntMap, Internal.hs:234-240) and `knn` only
concatenates per-tree candidates (RPTree.hs:176), so the build needs NO communication and a
query needs exactly one exchange step: ONE all-gather of every rank's exchange record — its
local top-k distances, ids and counts packed back to back (rpt_knn_record_layout:
nq * k * (8 + 4) + nq * 4 bytes per rank) — followed by a k-way merge
(rpt_knn_merge_records_dev).  The kernels that fill the record, the collective and the merge
are ordered on the device (the ctx stream is handed to torch as an ExternalStream), so a
query costs a single host synchronisation at the end.  Top-k of a union is a
subset of the union of per-shard top-ks, and because shard g holds trees [g*T/G, (g+1)*T/G)
the stable order (distance, shard, rank-in-shard) equals the reference's (distance, candidate
position) order.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from . import (RPT_KNN_DEDUP, RPT_KNN_KEEP_DUPLICATES, RPT_PROJ_AUTO, Dataset, _build, gen)


def tree_shard(T, world, rank):
    """Contiguous block of trees of `rank` -> (lo, hi).  T must be divisible by world."""
    if T % world != 0:
        raise ValueError("number of trees (%d) must be divisible by the world size (%d)" % (T, world))
    per = T // world
    return rank * per, (rank + 1) * per


def gather_topk(ids, dist_, cnt, group=None):
    """All-gather per-rank top-k lists -> shard-major tensors [G][nq][k], [G][nq][k], [G][nq].
    Works on CUDA tensors (RCCL) and on CPU tensors (gloo)."""
    world = dist.get_world_size(group)
    outs = []
    for x in (ids, dist_, cnt):
        x = x.contiguous()
        out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        try:
            dist.all_gather_into_tensor(out, x, group=group)
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty_like(x) for _ in range(world)]
            dist.all_gather(parts, x, group=group)
            out = torch.stack(parts)
        outs.append(out)
    return tuple(outs)


def record_layout(nq, k):
    """(bytes, off_dist, off_ids, off_count) of one shard's exchange record (C ABI)."""
    import ctypes as C
    v = [C.c_int64() for _ in range(4)]
    _lib.check(_lib.lib().rpt_knn_record_layout(nq, k, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


class ExchangeRecord:
    """One shard's kNN result as a single byte buffer + typed views into it (dist [nq][k] f64,
    ids [nq][k] i32, count [nq] i32).  device: torch device (CPU for the gloo rehearsal)."""

    def __init__(self, nq, k, device):
        self.nq, self.k = nq, k
        self.bytes, od, oi, oc = record_layout(nq, k)
        self.buf = torch.zeros(self.bytes, dtype=torch.uint8, device=device)
        self.dist = self.buf[od:od + nq * k * 8].view(torch.float64).view(nq, k)
        self.ids = self.buf[oi:oi + nq * k * 4].view(torch.int32).view(nq, k)
        self.count = self.buf[oc:oc + nq * 4].view(torch.int32)

    @staticmethod
    def views_of(gathered, g, nq, k):
        """(ids, dist, count) views of shard g inside an all-gathered [G][bytes] tensor."""
        _, od, oi, oc = record_layout(nq, k)
        row = gathered[g]
        return (row[oi:oi + nq * k * 4].view(torch.int32).view(nq, k),
                row[od:od + nq * k * 8].view(torch.float64).view(nq, k),
                row[oc:oc + nq * 4].view(torch.int32))


def gather_records(rec, group=None, out=None, via_host=False):
    """All-gather the ranks' exchange records -> uint8 tensor [G][bytes] (one collective).
    via_host: stage through host memory (rehearsal of several ranks on ONE GPU over gloo, which
    has no device all-gather; never used on a multi-GPU node)."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, rec.bytes), dtype=torch.uint8, device=rec.buf.device)
    if via_host:
        parts = [torch.empty(rec.bytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, rec.buf.cpu(), group=group)
        out.copy_(torch.stack(parts))
        return out
    try:
        dist.all_gather_into_tensor(out, rec.buf, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(rec.buf) for _ in range(world)]
        dist.all_gather(parts, rec.buf, group=group)
        out.copy_(torch.stack(parts))
    return out


class ShardedForest:
    """This rank's tree shard of a T-tree forest + the collective query."""

    def __init__(self, ctx, data, seed, maxd, minl, ntrees, pnz, dim, mode=RPT_PROJ_AUTO,
                 group=None, hyperplanes=None):
        self.ctx, self.group = ctx, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.data = Dataset.of(ctx, data)
        if hyperplanes is None:
            _, R = gen.forest_hyperplanes(seed, ntrees, maxd, pnz, dim)   # identical on all ranks
        else:
            R = np.asarray(hyperplanes, dtype=np.float64)
        lo, hi = tree_shard(ntrees, self.world, self.rank)
        self.R = R
        self.local = _build(ctx, self.data, np.ascontiguousarray(R[lo:hi]), maxd, minl, mode)
        self._stream = torch.cuda.ExternalStream(ctx.stream, device=ctx.device)
        self._bufs = {}

    def _buffers(self, nq, k):
        key = (nq, k)
        if key not in self._bufs:
            dev = torch.device("cuda", self.ctx.device)
            rec = ExchangeRecord(nq, k, dev)
            gathered = torch.empty((self.world, rec.bytes), dtype=torch.uint8, device=dev)
            out = (torch.empty((nq, k), dtype=torch.int32, device=dev),
                   torch.empty((nq, k), dtype=torch.float64, device=dev),
                   torch.empty((nq,), dtype=torch.int32, device=dev))
            self._bufs = {key: (rec, gathered, out)}   # one shape at a time stays resident
        return self._bufs[key]

    def knn(self, queries, k, dedup=False):
        """queries: Dataset (same on every rank).  Returns device tensors (ids, dist, count),
        valid until the next call with the same shape.  One host synchronisation."""
        L = _lib.lib()
        nq = queries.n
        flags = RPT_KNN_DEDUP if dedup else RPT_KNN_KEEP_DUPLICATES
        rec, gathered, (oi, od, oc) = self._buffers(nq, k)
        _lib.check(L.rpt_knn_dev(self.ctx._h, self.local._h, self.data._h, queries._h, k, flags,
                                 rec.ids.data_ptr(), rec.dist.data_ptr(), rec.count.data_ptr()))
        if self.world == 1:
            self.ctx.sync()
            return rec.ids, rec.dist, rec.count
        with torch.cuda.stream(self._stream):   # the collective waits for / is waited on by the
            gather_records(rec, self.group, out=gathered)   # ctx stream: no host sync in between
        _lib.check(L.rpt_knn_merge_records_dev(self.ctx._h, gathered.data_ptr(), rec.bytes,
                                               self.world, nq, k, flags, oi.data_ptr(),
                                               od.data_ptr(), oc.data_ptr()))
        self.ctx.sync()
        return oi, od, oc
