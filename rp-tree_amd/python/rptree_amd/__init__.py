"""rptree_amd — host-side mirror of the Data.RPTree API surface over the MI355X C ABI.

The reference host language is Haskell (no GHC in this image), so this Python layer plays the
role the Haskell `Data.RPTree` wrapper would play: same function names, argument order and
meaning as the reference, calling the C ABI of include/rptree_hip.h.  Reference citations are
file:line in ocramz/rp-tree v0.7.1.

    forestBatch / treeBatch   Batch.hs:29-63
    knn                       RPTree.hs:168-176
    candidates                RPTree.hs:289-314
    recallWith                RPTree.hs:259-282
    rpTreeCfg / RPTreeConfig  Conduit.hs:123-141
    SVector / DVector / Embed Internal.hs:56-59,92-133
    leaves / levels / points / treeSize / leafSizes   Internal.hs:199-208, RPTree.hs:362-367
"""
import atexit
import ctypes as C
import math
import weakref
from collections import namedtuple

import numpy as np

from . import gen
from ._lib import (RPT_BF16, RPT_F32, RPT_F64, RPT_KNN_DEDUP, RPT_KNN_DEDUP_DISTANCE,
                   RPT_KNN_KEEP_DUPLICATES, RPT_KNN_METRIC_REFERENCE,
                   RPT_PROJ_AUTO, RPT_PROJ_EXACT, RPT_PROJ_MFMA, RPTError, check, lib)

__all__ = [
    "Context", "Dataset", "RPForest", "RPTree", "SVector", "DVector", "Embed", "fromListSv",
    "fromVectorSv", "fromListDv", "fromVectorDv", "forestBatch", "treeBatch", "knn", "knnBatch",
    "candidates", "recallWith", "rpTreeCfg", "RPTreeConfig", "leaves", "levels", "points",
    "treeSize", "leafSizes", "metricL2", "inner", "project", "splitSegments", "topology",
    "bruteKnn", "RPTError", "forest", "tree", "saveForest", "loadForest", "importForest",
    "knnH", "knnHBatch", "knnPQ", "candidatesBatch", "to_bf16", "from_bf16", "RPStreamForest",
]

_DT = {np.dtype(np.float64): RPT_F64, np.dtype(np.float32): RPT_F32}


def to_bf16(a):
    """float array -> bf16 bit patterns (uint16), round to nearest even (numpy has no bf16)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) >> 16).astype(np.uint16)
    nan = (u & np.uint32(0x7FFFFFFF)) > np.uint32(0x7F800000)
    out[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return out


def from_bf16(u16):
    """bf16 bit patterns (uint16) -> float32 (exact)."""
    return (np.ascontiguousarray(u16, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def _vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


# ---------------------------------------------------------------------------------------
# vector types (Internal.hs:92-133)
# ---------------------------------------------------------------------------------------
class SVector(namedtuple("SVector", "svDim svIdx svVal")):
    """Sparse vector: dimension + (index, value) components, indices ascending (unchecked in
    the reference, Internal.hs:99-105)."""
    __slots__ = ()


class DVector(namedtuple("DVector", "dvVec")):
    __slots__ = ()


Embed = namedtuple("Embed", "eEmbed eData")           # Internal.hs:56-59


def fromListSv(n, ll):                                # Internal.hs:106-107
    idx = np.array([i for i, _ in ll], dtype=np.int32)
    val = np.array([x for _, x in ll], dtype=np.float64)
    return SVector(int(n), idx, val)


def fromVectorSv(n, idx, val):                        # Internal.hs:116-119
    return SVector(int(n), np.asarray(idx, dtype=np.int32), np.asarray(val, dtype=np.float64))


def fromListDv(ll):                                   # Internal.hs:128-129
    return DVector(np.array(ll, dtype=np.float64))


fromVectorDv = fromListDv                             # Internal.hs:130-131


# ---------------------------------------------------------------------------------------
# parameters (Conduit.hs:123-141)
# ---------------------------------------------------------------------------------------
RPTreeConfig = namedtuple("RPTreeConfig", "fpMaxTreeDepth fpDataChunkSize fpProjNzDensity")


def rpTreeCfg(minl, n, d):
    """Conduit.hs:132-141: maxd = ceiling(logBase 2 (n/minl)); chunk = ceiling(n/100);
    pnz = min 1 (1/logBase 10 d)."""
    maxd = math.ceil(math.log(n / minl) / math.log(2.0))
    nchunk = math.ceil(n / 100)
    with np.errstate(divide="ignore"):
        pnz_min = 1.0 / (math.log(d) / math.log(10.0)) if d != 1 else math.inf
    return RPTreeConfig(int(maxd), int(nchunk), min(pnz_min, 1.0))


# ---------------------------------------------------------------------------------------
# handles
# ---------------------------------------------------------------------------------------
# Handles still alive when the interpreter exits (e.g. kept by a traceback) are released HERE,
# forests before datasets before contexts, while the HIP runtime is still up; a __del__ that
# ran during interpreter teardown would call into a runtime whose static state is gone.
_live = weakref.WeakSet()


def _close_all():
    objs = list(_live)
    for kind in ("ShardedForest", "RPForest", "Dataset", "Comm", "Context"):
        for o in objs:
            if type(o).__name__ == kind:
                try:
                    o.close()
                except Exception:
                    pass


atexit.register(_close_all)


class Context:
    """One MI355X device + stream (rpt_ctx).  Raises if there is no usable HIP device."""

    def __init__(self, device=0):
        h = C.c_void_p()
        check(lib().rpt_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self._owns = True
        _live.add(self)

    @classmethod
    def _borrowed(cls, handle, device):
        """Wrap a ctx owned by someone else (rpt_comm_ctx): never destroyed from here."""
        self = cls.__new__(cls)
        self._h, self.device, self._owns = handle, int(device), False
        return self

    def sync(self):
        check(lib().rpt_ctx_sync(self._h))

    def set_option(self, name, value):
        """Algorithm switch of this context (rpt_ctx_set_option); returns the previous value."""
        old = self.get_option(name)
        check(lib().rpt_ctx_set_option(self._h, name.encode(), int(value)))
        return old

    def get_option(self, name):
        v = C.c_int64()
        check(lib().rpt_ctx_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    @property
    def stream(self):
        s = C.c_void_p()
        check(lib().rpt_ctx_stream(self._h, C.byref(s)))
        return s.value

    def close(self):
        if self._h is not None:
            if self._owns:
                lib().rpt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class Dataset:
    """Point set / query batch resident in HBM (rpt_dataset)."""

    def __init__(self, ctx, handle, n, d, dtype, is_csr, keep=None):
        self.ctx, self._h, self.n, self.d, self.dtype, self.is_csr = ctx, handle, n, d, dtype, is_csr
        self._keep = keep
        self._host = None
        _live.add(self)

    @staticmethod
    def dense(ctx, X, dtype=None):
        """Row-major host array -> HBM.  float64 / float32 arrays keep their type; a uint16 array
        with dtype=RPT_BF16 holds bf16 bit patterns (see to_bf16)."""
        X = np.ascontiguousarray(X)
        if X.ndim != 2:
            raise ValueError("dense data must be a 2-D array [n][d]")
        if dtype == RPT_BF16:
            if X.dtype != np.uint16:
                X = to_bf16(X)
            dt = RPT_BF16
        else:
            if X.dtype not in _DT:
                X = X.astype(np.float64)
            dt = _DT[X.dtype]
        h = C.c_void_p()
        check(lib().rpt_dataset_dense_host(ctx._h, _vp(X), X.shape[0], X.shape[1], dt, C.byref(h)))
        ds = Dataset(ctx, h, X.shape[0], X.shape[1], dt, False)
        ds._host = X if dt == RPT_F64 else None     # recallWith's value semantics read rows back
        return ds

    @staticmethod
    def dense_device(ctx, ptr, n, d, dtype, keep=None):
        """Borrow device memory (e.g. a torch tensor's data_ptr()); `keep` pins the owner.
        The ctx stream is NOT ordered against whatever stream produced that memory: the caller
        must have synchronised the producer (Dataset.from_torch does)."""
        h = C.c_void_p()
        check(lib().rpt_dataset_dense_dev(ctx._h, C.c_void_p(ptr), n, d, dtype, C.byref(h)))
        return Dataset(ctx, h, n, d, dtype, False, keep)

    @staticmethod
    def from_torch(ctx, t):
        """Borrow a contiguous 2-D torch tensor on ctx's device (float64 / float32 / bfloat16).
        torch's current stream is synchronised first: the library enqueues on its own
        non-blocking stream, which nothing else orders behind the kernels that filled `t`."""
        import torch
        dt = {torch.float64: RPT_F64, torch.float32: RPT_F32, torch.bfloat16: RPT_BF16}[t.dtype]
        if t.dim() != 2 or not t.is_contiguous() or t.device.index != ctx.device:
            raise ValueError("need a contiguous 2-D tensor on cuda:%d" % ctx.device)
        torch.cuda.current_stream(t.device).synchronize()
        return Dataset.dense_device(ctx, t.data_ptr(), t.shape[0], t.shape[1], dt, keep=t)

    @staticmethod
    def csr(ctx, rowptr, col, val, d):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val)
        if val.dtype not in _DT:
            val = val.astype(np.float64)
        h = C.c_void_p()
        check(lib().rpt_dataset_csr_host(ctx._h, _vp(rowptr), _vp(col), _vp(val), len(rowptr) - 1,
                                         int(d), _DT[val.dtype], C.byref(h)))
        return Dataset(ctx, h, len(rowptr) - 1, int(d), _DT[val.dtype], True)

    @staticmethod
    def csr_from_torch(ctx, rowptr, col, val, d):
        """Borrow CSR arrays held in torch tensors on ctx's device (int64 rowptr, int32 col,
        float64 / float32 val).  Not validated (see rpt_dataset_csr_dev); torch's current stream
        is synchronised first."""
        import torch
        dt = {torch.float64: RPT_F64, torch.float32: RPT_F32}[val.dtype]
        if rowptr.dtype != torch.int64 or col.dtype != torch.int32:
            raise ValueError("rowptr must be int64 and col int32")
        torch.cuda.current_stream(val.device).synchronize()
        n = rowptr.numel() - 1
        h = C.c_void_p()
        check(lib().rpt_dataset_csr_dev(ctx._h, C.c_void_p(rowptr.data_ptr()), C.c_void_p(col.data_ptr()),
                                        C.c_void_p(val.data_ptr()), n, int(d), dt, val.numel(),
                                        C.byref(h)))
        return Dataset(ctx, h, n, int(d), dt, True, keep=(rowptr, col, val))

    @staticmethod
    def of(ctx, data):
        """Pack `V.Vector (Embed v Double x)`-like input once at the boundary.  Accepts a
        Dataset, a 2-D array, (rowptr, col, val, d), a scipy CSR matrix, or a sequence of
        Embed / DVector / SVector values."""
        if isinstance(data, Dataset):
            return data
        if isinstance(data, np.ndarray):
            return Dataset.dense(ctx, data)
        if isinstance(data, tuple) and len(data) == 4 and not isinstance(data, (SVector,)):
            return Dataset.csr(ctx, *data)
        if hasattr(data, "indptr") and hasattr(data, "indices"):
            return Dataset.csr(ctx, data.indptr, data.indices, data.data, data.shape[1])
        vs = [x.eEmbed if isinstance(x, Embed) else x for x in data]
        if not vs:
            raise ValueError("empty dataset")
        if isinstance(vs[0], DVector):
            return Dataset.dense(ctx, np.stack([np.asarray(v.dvVec, dtype=np.float64) for v in vs]))
        if isinstance(vs[0], SVector):
            rowptr = np.zeros(len(vs) + 1, dtype=np.int64)
            rowptr[1:] = np.cumsum([len(v.svIdx) for v in vs])
            col = np.concatenate([v.svIdx for v in vs]).astype(np.int32)
            val = np.concatenate([v.svVal for v in vs]).astype(np.float64)
            return Dataset.csr(ctx, rowptr, col, val, vs[0].svDim)
        return Dataset.dense(ctx, np.asarray(vs, dtype=np.float64))

    def close(self):
        if self._h is not None:
            lib().rpt_dataset_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _query_dataset(ctx, like, q):
    """One query or a batch -> Dataset with the layout (dense/CSR) and dtype of `like`."""
    if isinstance(q, Dataset):
        return q, q.n
    npdt = np.float64 if like.dtype == RPT_F64 else np.float32
    if like.is_csr:
        if isinstance(q, SVector):
            q = [q]
        if isinstance(q, tuple) and len(q) == 4:
            return Dataset.csr(ctx, q[0], q[1], np.asarray(q[2], dtype=npdt), q[3]), len(q[0]) - 1
        rowptr = np.zeros(len(q) + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum([len(v.svIdx) for v in q])
        col = (np.concatenate([v.svIdx for v in q]) if len(q) else np.zeros(0)).astype(np.int32)
        val = (np.concatenate([v.svVal for v in q]) if len(q) else np.zeros(0)).astype(npdt)
        return Dataset.csr(ctx, rowptr, col, val, like.d), len(q)
    if isinstance(q, DVector):
        q = q.dvVec
    if like.dtype == RPT_BF16:       # data and queries share one element type on the device
        a = np.asarray(q)
        if a.dtype != np.uint16:
            a = to_bf16(a)
        if a.ndim == 1:
            a = a[None, :]
        return Dataset.dense(ctx, a, dtype=RPT_BF16), a.shape[0]
    a = np.asarray(q, dtype=npdt)
    if a.ndim == 1:
        a = a[None, :]
    return Dataset.dense(ctx, a), a.shape[0]


# ---------------------------------------------------------------------------------------
# forest (Internal.hs:139-182)
# ---------------------------------------------------------------------------------------
class RPTree:
    """View of one tree of a flat forest: `RPTree d l a` (Internal.hs:172-175)."""

    def __init__(self, forest, t):
        self.forest, self.t = forest, t

    @property
    def _rpVectors(self):
        R = self.forest.R[self.t]
        out = []
        for l in range(R.shape[0]):
            idx = np.nonzero(R[l])[0].astype(np.int32)
            out.append(SVector(R.shape[1], idx, R[l, idx]))
        return out


class RPForest:
    """`RPForest d a` = IntMap of trees keyed 0..T-1 (Internal.hs:182), held in HBM in the flat
    layout of include/rptree_hip.h; `perm`, `thr`, `mglo`, `mghi` copy it out."""

    def __init__(self, ctx, handle, data, R, max_depth, min_leaf, owns=True):
        self.ctx, self._h, self.data = ctx, handle, data
        self._owns = owns           # False: a shard borrowed from an rpt_sharded_forest
        self.R = R
        self.T, self.L, self.d = R.shape
        assert self.L == max_depth
        self.min_leaf = int(min_leaf)
        self.N = data.n
        self._perm = self._nodes = None
        _live.add(self)

    # IntMap-like access
    def __len__(self):
        return self.T

    def __getitem__(self, t):
        if not 0 <= t < self.T:
            raise KeyError(t)
        return RPTree(self, t)

    def __iter__(self):
        return (RPTree(self, t) for t in range(self.T))

    def keys(self):
        return range(self.T)

    @property
    def perm(self):
        if self._perm is None:
            p = np.empty((self.T, self.N), dtype=np.int32)
            check(lib().rpt_forest_get_perm(self._h, _vp(p)))
            self._perm = p
        return self._perm

    def _get_nodes(self):
        if self._nodes is None:
            nodes = (1 << self.L) - 1
            a = [np.empty((self.T, nodes), dtype=np.float64) for _ in range(3)]
            check(lib().rpt_forest_get_nodes(self._h, _vp(a[0]), _vp(a[1]), _vp(a[2])))
            self._nodes = a
        return self._nodes

    thr = property(lambda self: self._get_nodes()[0])
    mglo = property(lambda self: self._get_nodes()[1])
    mghi = property(lambda self: self._get_nodes()[2])

    def proj(self):
        dt = np.float64 if self.data.dtype == RPT_F64 else np.float32
        p = np.empty((self.T, self.L, self.N), dtype=dt)
        check(lib().rpt_forest_get_proj(self._h, _vp(p)))
        return p

    @property
    def mode(self):
        m = C.c_int32()
        check(lib().rpt_forest_get_mode(self._h, C.byref(m)))
        return m.value

    def stats(self):
        a, b = C.c_int64(), C.c_int64()
        check(lib().rpt_forest_stats(self._h, C.byref(a), C.byref(b)))
        return {"tie_nodes": a.value, "big_mid_nodes": b.value}

    def topology(self):
        return topology(self.N, self.L, self.min_leaf)

    def close(self):
        if self._h is not None:
            if self._owns:
                lib().rpt_forest_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def topology(n, max_depth, min_leaf):
    """DFS list of (level, heap, offset, size, is_leaf) — Internal.hs:289,495,503."""
    cnt = C.c_int64()
    check(lib().rpt_topology(n, max_depth, min_leaf, None, 0, C.byref(cnt)))
    out = np.empty((cnt.value, 5), dtype=np.int64)
    check(lib().rpt_topology(n, max_depth, min_leaf, _vp(out), cnt.value, C.byref(cnt)))
    return out


def _build(ctx, ds, R, maxd, minl, mode):
    R = np.ascontiguousarray(R, dtype=np.float64)
    T, L, d = R.shape
    if L != maxd:
        raise ValueError("hyperplane block has %d levels, maxDepth is %d" % (L, maxd))
    h = C.c_void_p()
    check(lib().rpt_forest_build(ctx._h, ds._h, _vp(R), T, L, int(minl), int(mode), C.byref(h)))
    return RPForest(ctx, h, ds, R, maxd, minl)


def forestBatch(seed, maxd, minl, ntrees, pnz, dim, src, *, ctx=None, mode=RPT_PROJ_AUTO,
                hyperplanes=None):
    """Batch.hs:48-63.  seed, max tree depth, min leaf size, number of trees, nonzero density of
    the projection vectors, their dimension, dataset.  The hyperplanes are sampled on the host
    exactly where the reference samples them (Batch.hs:59-61) unless `hyperplanes` (a dense
    [T][L][dim] block, e.g. produced by the real Haskell generator) is supplied."""
    ctx = ctx or default_context()
    ds = Dataset.of(ctx, src)
    if hyperplanes is None:
        _, R = gen.forest_hyperplanes(seed, ntrees, maxd, pnz, dim)
    else:
        R = np.asarray(hyperplanes, dtype=np.float64)
    if R.shape[2] != ds.d:
        # the reference never checks `dim` against the data; HBM indexing must
        raise ValueError("projection vector dimension %d != data dimension %d" % (R.shape[2], ds.d))
    return _build(ctx, ds, R, maxd, minl, mode)


def treeBatch(seed, maxDepth, minLeaf, pnz, dim, src, **kw):
    """Batch.hs:29-41 (= forestBatch with one tree: same draw order for a single tree)."""
    return forestBatch(seed, maxDepth, minLeaf, 1, pnz, dim, src, **kw)


class RPStreamForest(RPForest):
    """A forest built by the streaming insert (rpt_forest_stream_build): its trees have the
    EXPLICIT topology the fold over chunks produced — `kind[h]` 0 absent / 1 Bin / 2 Tip per heap
    slot (2^(L+1)-1 of them), a Tip's points = perm[t][leaf_off[h] : leaf_off[h] + leaf_len[h]] —
    the same for every tree.  `held` = points stored per tree (< N after the reference's data-loss
    branch, Internal.hs:277), `dropped` = how many that branch discarded (counted over the fold)."""

    def __init__(self, ctx, handle, data, R, max_depth, min_leaf):
        super().__init__(ctx, handle, data, R, max_depth, min_leaf)
        n = C.c_int64()
        check(lib().rpt_forest_get_topology(handle, C.byref(n), None, None, None, None, None))
        self.slots = n.value
        self.kind = np.empty(self.slots, dtype=np.int8)
        self.leaf_off = np.empty(self.slots, dtype=np.int64)
        self.leaf_len = np.empty(self.slots, dtype=np.int64)
        held, dropped = C.c_int64(), C.c_int64()
        check(lib().rpt_forest_get_topology(handle, C.byref(n), _vp(self.kind), _vp(self.leaf_off),
                                            _vp(self.leaf_len), C.byref(held), C.byref(dropped)))
        self.held, self.dropped = held.value, dropped.value

    def _get_nodes(self):
        if self._nodes is None:
            a = [np.empty((self.T, self.slots), dtype=np.float64) for _ in range(3)]
            check(lib().rpt_forest_get_nodes(self._h, _vp(a[0]), _vp(a[1]), _vp(a[2])))
            self._nodes = a
        return self._nodes

    thr = property(lambda self: self._get_nodes()[0])
    mglo = property(lambda self: self._get_nodes()[1])
    mghi = property(lambda self: self._get_nodes()[2])

    def topology(self):
        raise TypeError("a streamed forest has an explicit topology: kind / leaf_off / leaf_len")

    def tips(self, t):
        """{heap: ids} of tree t's Tips, heap order"""
        return {int(h): self.perm[t, self.leaf_off[h]:self.leaf_off[h] + self.leaf_len[h]]
                for h in np.nonzero(self.kind == 2)[0]}


def forest(seed, maxd, minl, ntrees, chunksize, pnz, dim, src, *, ctx=None, mode=RPT_PROJ_AUTO,
           hyperplanes=None):
    """Conduit.hs:104-121 `forest`: the streaming build with the reference's semantics — the source
    is cut into chunks of `chunksize` points (C.chunksOf: the last may be shorter) and every chunk
    is folded into every tree with `insert` (Internal.hs:245-297): a chunk part that meets a Bin is
    split at its OWN median and averaged into the threshold, a part that meets a Tip is put in
    front of its points, an empty part that meets a Bin drops the subtree (the reference's
    data-loss quirk, kept; `dropped` counts it).  Hyperplanes are sampled exactly like
    forestBatch's (same draw order, Conduit.hs:116-118).  Dense rows or SVector rows (a (rowptr, col,
    val, dim) tuple: `forest` is polymorphic in `Inner SVector v`, Conduit.hs:104-113).  With chunksize >= the
    number of points the result is forestBatch's forest (as a streamed-forest handle)."""
    ctx = ctx or default_context()
    xs = src if isinstance(src, (np.ndarray, Dataset, tuple)) else list(src)    # (tuple: CSR arrays + dim)
    ds = Dataset.of(ctx, xs)
    if hyperplanes is None:
        _, R = gen.forest_hyperplanes(seed, ntrees, maxd, pnz, dim)
    else:
        R = np.asarray(hyperplanes, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    if R.shape[2] != ds.d:
        raise ValueError("projection vector dimension %d != data dimension %d" % (R.shape[2], ds.d))
    T, L, _ = R.shape
    if L != maxd:
        raise ValueError("hyperplane block has %d levels, maxDepth is %d" % (L, maxd))
    h = C.c_void_p()
    check(lib().rpt_forest_stream_build(ctx._h, ds._h, _vp(R), T, L, int(minl), int(chunksize),
                                        int(mode), C.byref(h)))
    return RPStreamForest(ctx, h, ds, R, maxd, minl)


def tree(seed, maxDepth, minLeaf, chunksize, pnz, dim, src, **kw):
    """Conduit.hs:58-72 `tree`: see `forest`."""
    return forest(seed, maxDepth, minLeaf, 1, chunksize, pnz, dim, src, **kw)


def saveForest(path, forest_):
    """Flat on-disk format: hyperplanes, perm and node arrays in one .npz —
    a 10M-point forest needs no boxed `Embed`s.  The counterpart of serialiseRPForest
    (Internal.hs:185-188) for the flat layout; the data itself is not stored."""
    np.savez_compressed(path, R=forest_.R, min_leaf=forest_.min_leaf, N=forest_.N,
                        perm=forest_.perm, thr=forest_.thr, mglo=forest_.mglo, mghi=forest_.mghi,
                        mode=forest_.mode)


def loadForest(path, data, ctx=None):
    """deserialiseRPForest (Internal.hs:191-196) for the flat format: back into HBM via
    rpt_forest_import.  `data` is the point set the forest was built on."""
    ctx = ctx or (data.ctx if isinstance(data, Dataset) else default_context())
    z = np.load(path)
    ds = Dataset.of(ctx, data)
    if int(z["N"]) != ds.n:
        raise ValueError("forest was built on %d points, data has %d" % (int(z["N"]), ds.n))
    mode = int(z["mode"]) if "mode" in z.files else RPT_PROJ_AUTO
    return importForest(ctx, ds, z["R"], int(z["min_leaf"]), z["perm"], z["thr"], z["mglo"],
                        z["mghi"], mode=mode)


def importForest(ctx, data, R, min_leaf, perm, thr, mglo, mghi, mode=RPT_PROJ_AUTO):
    """Rebuild a device forest from flat arrays (e.g. after deserialiseRPForest).  mode: the
    projection mode the thresholds were computed with (queries project the same way)."""
    ds = Dataset.of(ctx, data)
    R = np.ascontiguousarray(R, dtype=np.float64)
    T, L, _ = R.shape
    h = C.c_void_p()
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (thr, mglo, mghi)]
    check(lib().rpt_forest_import(ctx._h, ds._h, _vp(R), T, L, int(min_leaf), _vp(perm),
                                  _vp(a[0]), _vp(a[1]), _vp(a[2]), C.byref(h)))
    check(lib().rpt_forest_set_mode(h, int(mode)))
    return RPForest(ctx, h, ds, R, L, min_leaf)


# ---- accessors (Internal.hs:199-208, RPTree.hs:362-367) ----
def leaves(tree):
    """All leaf buckets of a tree, left to right (ids)."""
    f, t = tree.forest, tree.t
    if isinstance(f, RPStreamForest):
        return _stream_leaves(f, t)
    topo = f.topology()
    return [f.perm[t, o:o + n] for (_, _, o, n, leaf) in topo if leaf]


def levels(tree):
    return tree.forest.L


def points(tree):
    if isinstance(tree.forest, RPStreamForest):
        return tree.forest.perm[tree.t, :tree.forest.held]
    return tree.forest.perm[tree.t]


def _stream_leaves(f, t):
    """Tips of a streamed tree in left-to-right (DFS) order"""
    out, stack = [], [0]
    while stack:
        h = stack.pop()
        if f.kind[h] == 2:
            out.append(f.perm[t, f.leaf_off[h]:f.leaf_off[h] + f.leaf_len[h]])
        elif f.kind[h] == 1:
            stack.append(2 * h + 2)
            stack.append(2 * h + 1)
    return out


def leafSizes(tree):
    if isinstance(tree.forest, RPStreamForest):
        return [len(x) for x in _stream_leaves(tree.forest, tree.t)]
    return [int(n) for (_, _, _, n, leaf) in tree.forest.topology() if leaf]


def treeSize(tree):
    return sum(leafSizes(tree))


# ---- algebra helpers on the host (single pairs; the batch lives on the device) ----
def inner(u, v):
    """Inner SVector DVector / SVector SVector / DVector DVector (Internal.hs:322-341) with the
    reference's summation order."""
    if isinstance(u, SVector) and isinstance(v, DVector):
        acc = 0.0
        m = min(len(u.svIdx), len(v.dvVec))
        for j in range(m - 1, -1, -1):
            acc = float(u.svVal[j]) * float(v.dvVec[u.svIdx[j]]) + acc
        return acc
    if isinstance(u, SVector) and isinstance(v, SVector):
        prods, a, b = [], 0, 0
        while a < len(u.svIdx) and b < len(v.svIdx):
            if u.svIdx[a] == v.svIdx[b]:
                prods.append(float(u.svVal[a]) * float(v.svVal[b]))
                a += 1
                b += 1
            elif u.svIdx[a] < v.svIdx[b]:
                a += 1
            else:
                b += 1
        acc = 0.0
        for p in reversed(prods):
            acc = p + acc
        return acc
    acc = 0.0
    for x, y in zip(u.dvVec, v.dvVec):
        acc = acc + float(x) * float(y)
    return acc


def metricL2(u, v):
    """metricDDL2 (Internal.hs:403-406) for dense pairs; the device kernels use the same
    definition (true Euclidean distance)."""
    a = np.asarray(u.dvVec if isinstance(u, DVector) else u, dtype=np.float64)
    b = np.asarray(v.dvVec if isinstance(v, DVector) else v, dtype=np.float64)
    return float(np.sqrt(np.sum((a - b) ** 2)))


# ---------------------------------------------------------------------------------------
# queries
# ---------------------------------------------------------------------------------------
def candidates(tree, q):
    """RPTree.hs:289-314: ids of the leaf buckets reached by q in this tree, left to right."""
    f = tree.forest
    off, ids = candidatesBatch(f, q)
    t = tree.t
    return ids[off[t]:off[t + 1]]


def candidatesBatch(forest, qs):
    """-> (off[nq*T+1], ids): candidates of (query i, tree t) = ids[off[i*T+t]:off[i*T+t+1]]"""
    ctx = forest.ctx
    qd, nq = _query_dataset(ctx, forest.data, qs)
    total = C.c_int64()
    off = np.empty(nq * forest.T + 1, dtype=np.int64)
    check(lib().rpt_candidates(ctx._h, forest._h, qd._h, _vp(off), None, 0, C.byref(total)))
    ids = np.empty(max(total.value, 1), dtype=np.int32)
    check(lib().rpt_candidates(ctx._h, forest._h, qd._h, _vp(off), _vp(ids), total.value,
                               C.byref(total)))
    return off, ids[:total.value]


def knnBatch(k, forest, qs, dedup=False, vote=0, reference_metric=False):
    """knn for a batch of queries -> (ids[nq][k], dist[nq][k], count[nq]).
    dedup: False = the reference's knn (duplicates kept), True = each id once,
    RPT_KNN_DEDUP_DISTANCE = knnPQ's `nub` (one entry per distance).
    vote = v > 0: only points found by at least v trees are ranked (RPT_KNN_VOTE; the
    reference's commented-out counts / keepCounts, RPTree.hs:464-478), ties by ascending id.
    reference_metric: SVector data are ranked by the reference's own metricSSL2 (Internal.hs:
    389-393: the merge of the index lists stops at the shorter vector's end) instead of the true
    Euclidean distance."""
    ctx = forest.ctx
    qd, nq = _query_dataset(ctx, forest.data, qs)
    ids = np.empty((nq, k), dtype=np.int32)
    dist = np.empty((nq, k), dtype=np.float64)
    cnt = np.empty(nq, dtype=np.int32)
    flags = (RPT_KNN_DEDUP_DISTANCE if dedup == RPT_KNN_DEDUP_DISTANCE
             else RPT_KNN_DEDUP if dedup else RPT_KNN_KEEP_DUPLICATES)
    flags |= int(vote) << 8                       # RPT_KNN_VOTE(v)
    if reference_metric:                          # SVector data: the truncating metricSSL2
        flags |= RPT_KNN_METRIC_REFERENCE
    check(lib().rpt_knn_host(ctx._h, forest._h, forest.data._h, qd._h, int(k), flags, _vp(ids),
                             _vp(dist), _vp(cnt)))
    return ids, dist, cnt


def knn_last_uncertified(ctx=None):
    """queries of the last knn call whose f32 prefilter cut could not be certified (re-run exactly)"""
    ctx = ctx or default_context()
    v = C.c_int64()
    check(lib().rpt_knn_last_uncertified(ctx._h, C.byref(v)))
    return int(v.value)


def knn(distf, k, tts, q, dedup=False):
    """RPTree.hs:168-176: `knn distf k forest q` -> [(distance, point id)] in increasing
    distance order, duplicates across trees kept (the reference never de-duplicates).
    Only distf = metricL2 is accelerated (the metric is evaluated on the device)."""
    if distf is not metricL2:
        raise NotImplementedError("the device path evaluates metricL2 only")
    ids, dist, cnt = knnBatch(k, tts, q, dedup=dedup)
    return [(float(dist[0, i]), int(ids[0, i])) for i in range(int(cnt[0]))]


def knnPQ(distf, k, tts, q):
    """RPTree.hs:181-194: like knn, but the heap's `nub` keeps ONE entry per distance value
    (the first in candidate order here; the reference's pick among ties depends on the heap)."""
    if distf is not metricL2:
        raise NotImplementedError("the device path evaluates metricL2 only")
    ids, dist, cnt = knnBatch(k, tts, q, dedup=RPT_KNN_DEDUP_DISTANCE)
    return [(float(dist[0, i]), int(ids[0, i])) for i in range(int(cnt[0]))]


def knnHBatch(k, forest, qs):
    """knnH for a batch of queries -> (off[nq+1], ids, dist): the result of query i is
    ids[off[i]:off[i+1]] with its distances."""
    ctx = forest.ctx
    qd, nq = _query_dataset(ctx, forest.data, qs)
    total = C.c_int64()
    off = np.empty(nq + 1, dtype=np.int64)
    check(lib().rpt_knnh_host(ctx._h, forest._h, forest.data._h, qd._h, int(k), _vp(off), None,
                              None, 0, C.byref(total)))
    ids = np.empty(max(total.value, 1), dtype=np.int32)
    dist = np.empty(max(total.value, 1), dtype=np.float64)
    check(lib().rpt_knnh_host(ctx._h, forest._h, forest.data._h, qd._h, int(k), _vp(off),
                              _vp(ids), _vp(dist), total.value, C.byref(total)))
    return off, ids[:total.value], dist[:total.value]


def knnH(distf, k, tts, q):
    """RPTree.hs:199-217: leaves are visited in increasing margin priority (candidatesH
    :318-342); whole buckets are taken while the running count stays <= k (at least one), the
    bucket taken last first.  As in the reference the result is NOT sorted by distance and NOT
    cut to k: [(distance, point id)]."""
    if distf is not metricL2:
        raise NotImplementedError("the device path evaluates metricL2 only")
    off, ids, dist = knnHBatch(k, tts, q)
    return [(float(dist[i]), int(ids[i])) for i in range(int(off[0]), int(off[1]))]


def bruteKnn(forest_or_data, qs, k, ctx=None):
    data = forest_or_data.data if isinstance(forest_or_data, RPForest) else forest_or_data
    ctx = ctx or data.ctx
    qd, nq = _query_dataset(ctx, data, qs)
    ids = np.empty((nq, k), dtype=np.int32)
    dist = np.empty((nq, k), dtype=np.float64)
    check(lib().rpt_brute_knn_host(ctx._h, data._h, qd._h, int(k), _vp(ids), _vp(dist)))
    return ids, dist


def recallWith(distf, tt, k, q):
    """RPTree.hs:259-282: mean over trees of |candidates(tree, q) ∩ true kNN| / k, the truth by
    brute force over all points."""
    if distf is not metricL2:
        raise NotImplementedError("the device path evaluates metricL2 only")
    true_ids, _ = bruteKnn(tt, q, k)
    kk = set(int(i) for i in true_ids[0] if i >= 0)
    off, ids = candidatesBatch(tt, q)
    rs = []
    for t in range(tt.T):
        aa = set(ids[off[t]:off[t + 1]].tolist())
        rs.append(len(aa & kk) / k)
    return sum(rs) / len(rs)


# ---------------------------------------------------------------------------------------
# stand-alone kernels (parity tests)
# ---------------------------------------------------------------------------------------
def project(data, R, mode=RPT_PROJ_AUTO, ctx=None):
    """P[c][i] = R[c] `inner` x_i (the N inner products of Internal.hs:504 as one batch)."""
    ctx = ctx or (data.ctx if isinstance(data, Dataset) else default_context())
    ds = Dataset.of(ctx, data)
    R = np.ascontiguousarray(R, dtype=np.float64)
    Cn = R.shape[0]
    dt = np.float64 if ds.dtype == RPT_F64 else np.float32
    P = np.empty((Cn, ds.n), dtype=dt)
    check(lib().rpt_project_host(ctx._h, ds._h, _vp(R), Cn, int(mode), _vp(P)))
    return P


def splitSegments(keys, perm, seg_off, seg_len, ctx=None):
    """partitionAtMedian (Internal.hs:486-505) of every segment on caller-supplied projections.
    Returns (perm sorted per segment, thr_mg[S][3])."""
    ctx = ctx or default_context()
    keys = np.ascontiguousarray(keys, dtype=np.float64)
    perm = np.array(perm, dtype=np.int32)
    so = np.ascontiguousarray(seg_off, dtype=np.int64)
    sl = np.ascontiguousarray(seg_len, dtype=np.int64)
    out = np.empty((len(so), 3), dtype=np.float64)
    check(lib().rpt_split_segments(ctx._h, _vp(keys), len(keys), _vp(perm), _vp(so), _vp(sl),
                                   len(so), _vp(out)))
    return perm, out
