"""Multi-GPU sharding of the hot path over the C ABI's multi-GPU entry points
(include/rptree_hip.h: rpt_comm_* / rpt_forest_build_sharded / rpt_knn_sharded*, implemented
on librccl in csrc/comm.hip): a contiguous block of trees per GPU, X replicated.

Trees are independent (createMulti maps over the IntMap, Internal.hs:234-240) and `knn` only
concatenates per-tree candidates (RPTree.hs:176), so the build needs NO communication and a
query needs exactly one exchange step: ONE ncclAllGather of every rank's exchange record — its
local top-k distances, ids and counts packed back to back (rpt_knn_record_layout:
nq * k * (8 + 4) + nq * 4 bytes per rank) — followed by a k-way merge.  The kernels that fill
the record, the collective and the merge are enqueued on the same ctx stream, so they are
ordered on the device and a query costs one host synchronisation at the end.  Top-k of a union
is a subset of the union of per-shard top-ks, and because shard g holds trees
[g*T/G, (g+1)*T/G) the stable order (distance, shard, rank-in-shard) equals the reference's
(distance, candidate position) order.

Two launch styles, same data path:
  Comm.local(n)                    one process drives n GPUs (rpt_comm_init)
  Comm.from_process_group(ctx)     one process per GPU under torch.distributed.run: the RCCL id
                                   made by rank 0 travels through the process group's store
                                   (rpt_comm_unique_id / rpt_comm_init_rank)
torch.distributed is only the control plane there (rendezvous, barriers); the exchange of the
data path is this library's own RCCL communicator.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import (RPT_KNN_DEDUP, RPT_KNN_KEEP_DUPLICATES, RPT_PROJ_AUTO, Context, Dataset, RPForest,
               _live)


def tree_shard(T, world, rank):
    """Contiguous block of trees of `rank` -> (lo, hi) = (rank*T // world, (rank+1)*T // world),
    the split rpt_forest_build_sharded uses.  Every rank needs at least one tree."""
    if T < world:
        raise ValueError("%d trees cannot be sharded over %d ranks" % (T, world))
    return rank * T // world, (rank + 1) * T // world


def record_layout(nq, k):
    """(bytes, off_dist, off_ids, off_count) of one shard's exchange record (C ABI)."""
    v = [C.c_int64() for _ in range(4)]
    _lib.check(_lib.lib().rpt_knn_record_layout(nq, k, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def _ptr_array(handles):
    arr = (C.c_void_p * len(handles))()
    for i, h in enumerate(handles):
        arr[i] = h if isinstance(h, int) or h is None else h.value
    return arr


class Comm:
    """rpt_comm: the devices this process drives + their RCCL communicators."""

    def __init__(self, handle):
        self._h = handle
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().rpt_comm_info(handle, C.byref(a), C.byref(b), C.byref(c)))
        self.nranks, self.nlocal, self.first_rank = a.value, b.value, c.value
        self._ctx = []
        for g in range(self.nlocal):
            h = C.c_void_p()
            _lib.check(_lib.lib().rpt_comm_ctx(handle, g, C.byref(h)))
            self._ctx.append(h)
        _live.add(self)             # released at interpreter exit in dependency order

    @staticmethod
    def local(n_gpus):
        """One process, n_gpus devices (ncclCommInitAll); the contexts belong to the comm."""
        h = C.c_void_p()
        _lib.check(_lib.lib().rpt_comm_init(int(n_gpus), C.byref(h)))
        c = Comm(h)
        c.contexts = [Context._borrowed(x, g) for g, x in enumerate(c._ctx)]
        for x in c.contexts:
            x._owner = c            # the comm owns these rpt_ctx: it must outlive their users
        return c

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(_lib.RPT_COMM_UID_BYTES)
        _lib.check(_lib.lib().rpt_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def rank(ctx, nranks, rank, uid):
        """One process per GPU: this process is `rank` of `nranks` (ncclCommInitRank)."""
        if len(uid) != _lib.RPT_COMM_UID_BYTES:
            raise ValueError("unique id must be %d bytes" % _lib.RPT_COMM_UID_BYTES)
        h = C.c_void_p()
        _lib.check(_lib.lib().rpt_comm_init_rank(ctx._h, int(nranks), int(rank), uid, C.byref(h)))
        c = Comm(h)
        c.contexts = [ctx]
        return c

    @staticmethod
    def from_process_group(ctx, group=None):
        """Under torch.distributed.run: rank 0 makes the RCCL id, the group's object broadcast
        carries its bytes, every rank joins with its own ctx."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return Comm.rank(ctx, world, rank, box[0])

    def sync(self):
        _lib.check(_lib.lib().rpt_comm_sync(self._h))

    def close(self):
        """rpt_comm_destroy.  After rpt_comm_init the communicator OWNS its contexts, and destroying
        them under a live Dataset / RPForest would leave that handle with a dangling ctx (its own
        close dereferences it): everything still alive on those contexts is closed first, forests
        before datasets."""
        if self._h is None:
            return
        mine = [c for c in getattr(self, "contexts", []) if getattr(c, "_owner", None) is self]
        if mine:
            objs = [o for o in list(_live) if getattr(o, "ctx", None) in mine or
                    (type(o).__name__ == "ShardedForest" and o.comm is self)]
            for kind in ("ShardedForest", "RPStreamForest", "RPForest", "Dataset"):
                for o in objs:
                    if type(o).__name__ == kind:
                        try:
                            o.close()
                        except Exception:
                            pass
            for c in mine:
                c._h = None
        _lib.lib().rpt_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedForest:
    """rpt_sharded_forest: this process's tree shards of a T-tree forest + the collective query.
    `datasets`: one replica of the point set per local device (Dataset on comm.contexts[g])."""

    def __init__(self, comm, datasets, R, maxd, minl, mode=RPT_PROJ_AUTO):
        R = np.ascontiguousarray(R, dtype=np.float64)
        T, L, d = R.shape
        if L != maxd:
            raise ValueError("hyperplane block has %d levels, maxDepth is %d" % (L, maxd))
        if len(datasets) != comm.nlocal:
            raise ValueError("need one dataset replica per local device (%d)" % comm.nlocal)
        self.comm, self.datasets, self.R, self.T, self.L, self.min_leaf = comm, list(datasets), R, T, L, minl
        h = C.c_void_p()
        _lib.check(_lib.lib().rpt_forest_build_sharded(
            comm._h, _ptr_array([x._h for x in datasets]), C.c_void_p(R.ctypes.data), T, L,
            int(minl), int(mode), C.byref(h)))
        self._h = h
        _live.add(self)

    def local(self, g=0):
        """(forest, first_tree, n_trees) of local device g; the forest handle is borrowed."""
        f, lo, nt = C.c_void_p(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().rpt_sharded_forest_local(self._h, g, C.byref(f), C.byref(lo), C.byref(nt)))
        fr = RPForest(self.comm.contexts[g], f, self.datasets[g],
                      np.ascontiguousarray(self.R[lo.value:lo.value + nt.value]), self.L,
                      self.min_leaf, owns=False)
        fr._keep_alive = self
        return fr, lo.value, nt.value

    def knn_dev(self, queries, k, flags, ids_ptrs, dist_ptrs, count_ptrs):
        """queries: one replica per local device; *_ptrs: device addresses of the per-device
        outputs.  Enqueued on the ctx streams: comm.sync() before reading."""
        _lib.check(_lib.lib().rpt_knn_sharded_dev(
            self.comm._h, self._h, _ptr_array([x._h for x in self.datasets]),
            _ptr_array([x._h for x in queries]), int(k), int(flags), _ptr_array(ids_ptrs),
            _ptr_array(dist_ptrs), _ptr_array(count_ptrs)))

    def knn(self, queries, k, dedup=False):
        """-> host arrays (ids[nq][k], dist[nq][k], count[nq]) of the merged answer."""
        nq = queries[0].n
        ids = np.empty((nq, k), dtype=np.int32)
        dist = np.empty((nq, k), dtype=np.float64)
        cnt = np.empty(nq, dtype=np.int32)
        flags = RPT_KNN_DEDUP if dedup else RPT_KNN_KEEP_DUPLICATES
        _lib.check(_lib.lib().rpt_knn_sharded(
            self.comm._h, self._h, _ptr_array([x._h for x in self.datasets]),
            _ptr_array([x._h for x in queries]), int(k), flags, C.c_void_p(ids.ctypes.data),
            C.c_void_p(dist.ctypes.data), C.c_void_p(cnt.ctypes.data)))
        return ids, dist, cnt

    def close(self):
        if self._h is not None:
            _lib.lib().rpt_sharded_forest_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
