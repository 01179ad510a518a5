"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), a contiguous block of trees per rank, X replicated.

Trees are independent (createMulti maps over the IntMap, Internal.hs:234-240) and `knn` only
concatenates per-tree candidates (RPTree.hs:176), so the build needs NO communication and a
query needs exactly one exchange step: an all-gather of every rank's local top-k
(nq * k * (4 + 8) + nq * 4 bytes per rank), followed by a k-way merge.  Top-k of a union is a
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

    def knn(self, queries, k, dedup=False):
        """queries: Dataset (same on every rank).  Returns device tensors (ids, dist, count)."""
        L = _lib.lib()
        dev = torch.device("cuda", self.ctx.device)
        nq = queries.n
        flags = RPT_KNN_DEDUP if dedup else RPT_KNN_KEEP_DUPLICATES
        ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
        dd = torch.empty((nq, k), dtype=torch.float64, device=dev)
        cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
        _lib.check(L.rpt_knn_dev(self.ctx._h, self.local._h, self.data._h, queries._h, k, flags,
                                 ids.data_ptr(), dd.data_ptr(), cnt.data_ptr()))
        self.ctx.sync()
        if self.world == 1:
            return ids, dd, cnt
        gi, gd, gc = gather_topk(ids, dd, cnt, self.group)
        torch.cuda.synchronize()
        oi, od, oc = torch.empty_like(ids), torch.empty_like(dd), torch.empty_like(cnt)
        _lib.check(L.rpt_knn_merge_dev(self.ctx._h, gi.data_ptr(), gd.data_ptr(), gc.data_ptr(),
                                       self.world, nq, k, flags, oi.data_ptr(), od.data_ptr(),
                                       oc.data_ptr()))
        self.ctx.sync()
        return oi, od, oc
