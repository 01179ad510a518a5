"""ctypes binding of the CPU oracle (oracle/librptree_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.  See rptree_oracle.h for
what the oracle restates (reference file:line) and its pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librptree_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "rptree_oracle.cpp")
    hdr = os.path.join(_HERE, "rptree_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "librptree_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


def _p(a, ty):
    if a is None:
        return None
    return a.ctypes.data_as(ty)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        L = _lib
        L.rpo_inner_ss.restype = C.c_double
        L.rpo_inner_sd.restype = C.c_double
        L.rpo_inner_dd.restype = C.c_double
        L.rpo_metric_dd.restype = C.c_double
        L.rpo_metric_sd.restype = C.c_double
        L.rpo_metric_ss.restype = C.c_double
        L.rpo_sum_sd.restype = C.c_int64
        L.rpo_diff_sd.restype = C.c_int64
        L.rpo_partition_at_median.restype = C.c_int64
        L.rpo_candidates_dense.restype = C.c_int64
        L.rpo_candidates_sparse.restype = C.c_int64
        L.rpo_knn_dense.restype = C.c_int32
        L.rpo_knn_csr.restype = C.c_int32
        L.rpo_recall_with_dense.restype = C.c_double
        L.rpo_knn_h_dense.restype = C.c_int64
        L.rpo_knn_h_csr.restype = C.c_int64
        L.rpo_candidates_h_dense.restype = C.c_int64
        L.rpo_stream_knn_h_dense.restype = C.c_int64
        L.rpo_data_normal_sparse2.restype = C.c_int64
        L.rpo_data_sparse_uniform.restype = C.c_int64
        L.rpo_next_double.restype = C.c_double
        L.rpo_next_word64.restype = C.c_uint64
        L.rpo_normal.restype = C.c_double
        L.rpo_keep_counts.restype = C.c_int64
        L.rpo_recall_with_dense_values.restype = C.c_double
        L.rpo_metric_dd_libm.restype = C.c_double
        L.rpo_pow2_mismatches.restype = C.c_int64
        L.rpo_stream_candidates_dense.restype = C.c_int64
        L.rpo_stream_knn_dense.restype = C.c_int32
    return _lib


class Gen(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("gamma", C.c_uint64)]


def _sv(idx, val):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    return idx, val


# ---- algebra (Internal.hs:351-470) ----
def inner_ss(i1, v1, i2, v2):
    i1, v1 = _sv(i1, v1)
    i2, v2 = _sv(i2, v2)
    return lib().rpo_inner_ss(C.c_int64(len(i1)), _p(i1, _i32p), _p(v1, _f64p),
                              C.c_int64(len(i2)), _p(i2, _i32p), _p(v2, _f64p))


def inner_sd(i1, v1, x):
    i1, v1 = _sv(i1, v1)
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().rpo_inner_sd(C.c_int64(len(i1)), _p(i1, _i32p), _p(v1, _f64p),
                              C.c_int64(len(x)), _p(x, _f64p))


def inner_dd(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().rpo_inner_dd(C.c_int64(min(len(a), len(b))), _p(a, _f64p), _p(b, _f64p))


def metric_dd(u, v):
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    return lib().rpo_metric_dd(C.c_int64(min(len(u), len(v))), _p(u, _f64p), _p(v, _f64p))


def metric_dd_libm(u, v):
    """metricDDL2 with `** 2` through this box's libm pow (see sq() in rptree_oracle.cpp)"""
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    return lib().rpo_metric_dd_libm(C.c_int64(len(u)), _p(u, _f64p), _p(v, _f64p))


def pow2_mismatches(seed, n):
    return int(lib().rpo_pow2_mismatches(C.c_uint64(seed), C.c_int64(n)))


def metric_sd(i1, v1, x):
    i1, v1 = _sv(i1, v1)
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().rpo_metric_sd(C.c_int64(len(i1)), _p(i1, _i32p), _p(v1, _f64p),
                               C.c_int64(len(x)), _p(x, _f64p))


def metric_ss(i1, v1, i2, v2):
    i1, v1 = _sv(i1, v1)
    i2, v2 = _sv(i2, v2)
    return lib().rpo_metric_ss(C.c_int64(len(i1)), _p(i1, _i32p), _p(v1, _f64p),
                               C.c_int64(len(i2)), _p(i2, _i32p), _p(v2, _f64p))


def sum_sd(i1, v1, x, minus=False):
    i1, v1 = _sv(i1, v1)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(len(x), dtype=np.float64)
    f = lib().rpo_diff_sd if minus else lib().rpo_sum_sd
    m = f(C.c_int64(len(i1)), _p(i1, _i32p), _p(v1, _f64p), C.c_int64(len(x)), _p(x, _f64p),
          _p(out, _f64p))
    return out[:m].copy()


def diff_sd(i1, v1, x):
    return sum_sd(i1, v1, x, minus=True)


def tree_cfg(min_leaf, n, d):
    """Conduit.hs:132-141 rpTreeCfg -> (maxDepth, chunk, pnz)"""
    md = C.c_int32()
    ch = C.c_int64()
    pnz = C.c_double()
    lib().rpo_tree_cfg(C.c_int32(min_leaf), C.c_int64(n), C.c_int32(d), C.byref(md), C.byref(ch),
                       C.byref(pnz))
    return md.value, ch.value, pnz.value


def partition_at_median(p):
    """Internal.hs:486-512 on precomputed projections -> (nh, order, thr, mglo, mghi)"""
    p = np.ascontiguousarray(p, dtype=np.float64)
    order = np.empty(len(p), dtype=np.int32)
    tm = np.empty(3, dtype=np.float64)
    nh = lib().rpo_partition_at_median(C.c_int64(len(p)), _p(p, _f64p), _p(order, _i32p),
                                       _p(tm, _f64p))
    return nh, order, tm[0], tm[1], tm[2]


# ---- generators (third-party RNG restatement: self-consistent, parity unpinned) ----
def forest_hyperplanes(seed, T, L, pnz, d):
    """Batch.hs:57-61 -> dense-ified R[T][L][d], nnz[T][L]"""
    R = np.zeros((T, L, d), dtype=np.float64)
    nnz = np.zeros((T, L), dtype=np.int32)
    lib().rpo_forest_hyperplanes(C.c_uint64(seed), C.c_int32(T), C.c_int32(L), C.c_double(pnz),
                                 C.c_int32(d), _p(R, _f64p), _p(nnz, _i32p))
    return R, nnz


def data_normal_dense2(seed, n, d):
    X = np.empty((n, d), dtype=np.float64)
    lib().rpo_data_normal_dense2(C.c_uint64(seed), C.c_int64(n), C.c_int32(d), _p(X, _f64p))
    return X


def data_circle2d2(seed, n):
    X = np.empty((n, 2), dtype=np.float64)
    lib().rpo_data_circle2d2(C.c_uint64(seed), C.c_int64(n), _p(X, _f64p))
    return X


def _csr_gen(fn, seed, n, d, pnz):
    cap = int(n * d * min(1.0, pnz * 1.2 + 0.05)) + 1024
    rowptr = np.empty(n + 1, dtype=np.int64)
    col = np.empty(cap, dtype=np.int32)
    val = np.empty(cap, dtype=np.float64)
    nnz = fn(C.c_uint64(seed), C.c_int64(n), C.c_int32(d), C.c_double(pnz), _p(rowptr, _i64p),
             _p(col, _i32p), _p(val, _f64p), C.c_int64(cap))
    if nnz > cap:
        raise RuntimeError("csr capacity too small")
    return rowptr, col[:nnz].copy(), val[:nnz].copy()


def data_normal_sparse2(seed, n, d, pnz):
    return _csr_gen(lib().rpo_data_normal_sparse2, seed, n, d, pnz)


def data_sparse_uniform(seed, n, d, pnz):
    return _csr_gen(lib().rpo_data_sparse_uniform, seed, n, d, pnz)


# ---- forest build (Batch.hs:48-63 -> Internal.hs) ----
class Forest:
    """Flat forest (layout in rptree_oracle.h)."""

    def __init__(self, N, d, R, L, min_leaf, perm, thr, mglo, mghi, proj=None):
        self.N, self.d, self.R, self.L, self.min_leaf = N, d, R, L, min_leaf
        self.T = R.shape[0]
        self.perm, self.thr, self.mglo, self.mghi, self.proj = perm, thr, mglo, mghi, proj


def _alloc(N, T, L, want_proj):
    nodes = (1 << L) - 1
    perm = np.empty((T, N), dtype=np.int32)
    thr = np.empty((T, nodes), dtype=np.float64)
    mglo = np.empty((T, nodes), dtype=np.float64)
    mghi = np.empty((T, nodes), dtype=np.float64)
    proj = np.full((T, L, N), np.nan, dtype=np.float64) if want_proj else None
    return perm, thr, mglo, mghi, proj


def _rows(X):
    """dense rows for the typed entry points: float32 stays float32 (upcast exactly on read by
    the oracle), everything else becomes float64 -> (array, xdtype)"""
    X = np.asarray(X)
    if X.dtype == np.float32:
        return np.ascontiguousarray(X), 1
    return np.ascontiguousarray(X, dtype=np.float64), 0


def forest_build_dense(X, R, min_leaf, want_proj=False, threads=1):
    """float32 X: the reference's arithmetic on the exactly-upcast rows.  threads > 1 builds
    trees concurrently (identical result; the reference is single-threaded)."""
    X, xdt = _rows(X)
    R = np.ascontiguousarray(R, dtype=np.float64)
    N, d = X.shape
    T, L, d2 = R.shape
    assert d2 == d
    perm, thr, mglo, mghi, proj = _alloc(N, T, L, want_proj)
    lib().rpo_forest_build_dense_ex(C.c_void_p(X.ctypes.data), C.c_int32(xdt), C.c_int64(N),
                                    C.c_int32(d), _p(R, _f64p), C.c_int32(T), C.c_int32(L),
                                    C.c_int32(min_leaf), _p(perm, _i32p), _p(thr, _f64p),
                                    _p(mglo, _f64p), _p(mghi, _f64p), _p(proj, _f64p),
                                    C.c_int32(threads))
    return Forest(N, d, R, L, min_leaf, perm, thr, mglo, mghi, proj)


def forest_build_csr(rowptr, col, val, d, R, min_leaf, want_proj=False, threads=1):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    N = len(rowptr) - 1
    T, L, d2 = R.shape
    assert d2 == d
    perm, thr, mglo, mghi, proj = _alloc(N, T, L, want_proj)
    lib().rpo_forest_build_csr_ex(_p(rowptr, _i64p), _p(col, _i32p), _p(val, _f64p),
                                  C.c_int64(N), C.c_int32(d), _p(R, _f64p), C.c_int32(T),
                                  C.c_int32(L), C.c_int32(min_leaf), _p(perm, _i32p),
                                  _p(thr, _f64p), _p(mglo, _f64p), _p(mghi, _f64p),
                                  _p(proj, _f64p), C.c_int32(threads))
    return Forest(N, d, R, L, min_leaf, perm, thr, mglo, mghi, proj)


def _fargs(f):
    return (_p(f.R, _f64p), C.c_int32(f.T), C.c_int32(f.L), C.c_int32(f.min_leaf))


def _targs(f):
    return (_p(f.perm, _i32p), _p(f.thr, _f64p), _p(f.mglo, _f64p), _p(f.mghi, _f64p))


def candidates_dense(f, q, t):
    """RPTree.hs:289-314 for tree t -> ids in leaf (left-to-right) order"""
    q = np.ascontiguousarray(q, dtype=np.float64)
    out = np.empty(f.N, dtype=np.int32)
    n = lib().rpo_candidates_dense(_p(q, _f64p), C.c_int32(f.d), *_fargs(f), C.c_int64(f.N),
                                   *_targs(f), C.c_int32(t), _p(out, _i32p), C.c_int64(f.N))
    return out[:n].copy()


def candidates_sparse(f, qi, qv, t):
    qi, qv = _sv(qi, qv)
    out = np.empty(f.N, dtype=np.int32)
    n = lib().rpo_candidates_sparse(C.c_int64(len(qi)), _p(qi, _i32p), _p(qv, _f64p),
                                    C.c_int32(f.d), *_fargs(f), C.c_int64(f.N), *_targs(f),
                                    C.c_int32(t), _p(out, _i32p), C.c_int64(f.N))
    return out[:n].copy()


def knn_dense(f, X, q, k, dedup=False):
    """RPTree.hs:168-176 with metricL2 (metricDDL2) -> (ids, dists)"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    ids = np.empty(k, dtype=np.int32)
    dist = np.empty(k, dtype=np.float64)
    m = lib().rpo_knn_dense(_p(X, _f64p), C.c_int64(f.N), C.c_int32(f.d), _p(q, _f64p),
                            *_fargs(f), *_targs(f), C.c_int32(k), C.c_int32(int(dedup)),
                            _p(ids, _i32p), _p(dist, _f64p))
    return ids[:m].copy(), dist[:m].copy()


def knn_dense_batch(f, X, Q, k, dedup=0, vote_thr=0, threads=1):
    """knn for a batch of dense queries -> (ids[nq][k], dist[nq][k], count[nq]); float32 X is
    upcast exactly on read; vote_thr > 0 = keepCounts before the distances (extension)."""
    X, xdt = _rows(X)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    if Q.ndim == 1:
        Q = Q[None, :]
    nq = Q.shape[0]
    ids = np.empty((nq, k), dtype=np.int32)
    dist = np.empty((nq, k), dtype=np.float64)
    cnt = np.empty(nq, dtype=np.int32)
    lib().rpo_knn_dense_batch(C.c_void_p(X.ctypes.data), C.c_int32(xdt), C.c_int64(f.N),
                              C.c_int32(f.d), _p(Q, _f64p), C.c_int64(nq), *_fargs(f),
                              *_targs(f), C.c_int32(k), C.c_int32(int(dedup)),
                              C.c_int32(int(vote_thr)), _p(ids, _i32p), _p(dist, _f64p),
                              _p(cnt, _i32p), C.c_int32(threads))
    return ids, dist, cnt


def keep_counts(ids, thr):
    """RPTree.hs:464-478 counts + keepCounts -> (ids ascending, counts)"""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    oi = np.empty(len(ids), dtype=np.int32)
    oc = np.empty(len(ids), dtype=np.int32)
    m = lib().rpo_keep_counts(_p(ids, _i32p), C.c_int64(len(ids)), C.c_int32(thr), _p(oi, _i32p),
                              _p(oc, _i32p))
    return oi[:m].copy(), oc[:m].copy()


def knn_csr(f, rowptr, col, val, qi, qv, k, dedup=False, true_l2=False):
    qi, qv = _sv(qi, qv)
    ids = np.empty(k, dtype=np.int32)
    dist = np.empty(k, dtype=np.float64)
    m = lib().rpo_knn_csr(_p(rowptr, _i64p), _p(col, _i32p), _p(val, _f64p), C.c_int64(f.N),
                          C.c_int32(f.d), C.c_int64(len(qi)), _p(qi, _i32p), _p(qv, _f64p),
                          *_fargs(f), *_targs(f), C.c_int32(k), C.c_int32(int(dedup)),
                          C.c_int32(int(true_l2)), _p(ids, _i32p), _p(dist, _f64p))
    return ids[:m].copy(), dist[:m].copy()


def knn_pq_dense(f, X, q, k):
    """RPTree.hs:181-194 knnPQ: like knn, entries of equal distance collapse to one"""
    return knn_dense(f, X, q, k, dedup=2)


def candidates_h_dense(f, q, t):
    """RPTree.hs:318-342 for tree t -> (priority, perm offset, length) per leaf reached, DFS order"""
    q = np.ascontiguousarray(q, dtype=np.float64)
    cap = 1 << min(f.L, 20)
    prio = np.empty(cap, dtype=np.float64)
    off = np.empty(cap, dtype=np.int64)
    ln = np.empty(cap, dtype=np.int64)
    tt = _targs(f)
    n = lib().rpo_candidates_h_dense(_p(q, _f64p), C.c_int32(f.d), *_fargs(f), C.c_int64(f.N),
                                     tt[1], tt[2], tt[3], C.c_int32(t), _p(prio, _f64p),
                                     _p(off, _i64p), _p(ln, _i64p), C.c_int64(cap))
    return prio[:n].copy(), off[:n].copy(), ln[:n].copy()


def knn_h_dense(f, X, q, k):
    """RPTree.hs:199-217 knnH with metricL2 -> (ids, dists): whole buckets, unsorted"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    cap = 4096
    while True:
        ids = np.empty(cap, dtype=np.int32)
        dist = np.empty(cap, dtype=np.float64)
        m = lib().rpo_knn_h_dense(_p(X, _f64p), C.c_int64(f.N), C.c_int32(f.d), _p(q, _f64p),
                                  *_fargs(f), *_targs(f), C.c_int32(k), _p(ids, _i32p),
                                  _p(dist, _f64p), C.c_int64(cap))
        if m <= cap:
            return ids[:m].copy(), dist[:m].copy()
        cap = int(m)


def knn_h_csr(f, rowptr, col, val, qi, qv, k, true_l2=False):
    qi, qv = _sv(qi, qv)
    cap = 4096
    while True:
        ids = np.empty(cap, dtype=np.int32)
        dist = np.empty(cap, dtype=np.float64)
        m = lib().rpo_knn_h_csr(_p(rowptr, _i64p), _p(col, _i32p), _p(val, _f64p),
                                C.c_int64(f.N), C.c_int32(f.d), C.c_int64(len(qi)),
                                _p(qi, _i32p), _p(qv, _f64p), *_fargs(f), *_targs(f),
                                C.c_int32(k), C.c_int32(int(true_l2)), _p(ids, _i32p),
                                _p(dist, _f64p), C.c_int64(cap))
        if m <= cap:
            return ids[:m].copy(), dist[:m].copy()
        cap = int(m)


def recall_with_dense(f, X, q, k):
    """RPTree.hs:259-282 mean per-tree candidate recall"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    return lib().rpo_recall_with_dense(_p(X, _f64p), C.c_int64(f.N), C.c_int32(f.d),
                                       _p(q, _f64p), *_fargs(f), *_targs(f), C.c_int32(k))


def recall_with_dense_values(f, X, q, k):
    """RPTree.hs:259-282 with Set-of-values semantics (equal rows are one element)"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    return lib().rpo_recall_with_dense_values(_p(X, _f64p), C.c_int64(f.N), C.c_int32(f.d),
                                              _p(q, _f64p), *_fargs(f), *_targs(f), C.c_int32(k))


class StreamForest:
    """Result of the streaming build (heap arrays of 2^(L+1)-1 slots per tree)."""

    def __init__(self, N, L, kind, thr, mglo, mghi, leaf_off, leaf_len, leaf_ids, held):
        self.N, self.L = N, L
        self.kind, self.thr, self.mglo, self.mghi = kind, thr, mglo, mghi
        self.leaf_off, self.leaf_len, self.leaf_ids, self.held = leaf_off, leaf_len, leaf_ids, held

    def leaves(self, t):
        """Tip payloads of tree t in heap order: {heap: ids}"""
        return {int(h): self.leaf_ids[t, self.leaf_off[t, h]:self.leaf_off[t, h] + self.leaf_len[t, h]]
                for h in np.nonzero(self.kind[t] == 2)[0]}


def stream_forest_dense(X, R, min_leaf, chunk):
    """Conduit.hs:104-121 `forest` on a source yielding the rows of X in order, in chunks of
    `chunk` points (insertMultiC / Internal.hs:245-297)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    N, d = X.shape
    T, L, _ = R.shape
    S = (1 << (L + 1)) - 1
    kind = np.zeros((T, S), dtype=np.int8)
    thr, mglo, mghi = (np.empty((T, S), dtype=np.float64) for _ in range(3))
    leaf_off = np.zeros((T, S), dtype=np.int64)
    leaf_len = np.zeros((T, S), dtype=np.int64)
    leaf_ids = np.full((T, max(N, 1)), -1, dtype=np.int32)
    held = np.zeros(T, dtype=np.int64)
    lib().rpo_stream_forest_dense(_p(X, _f64p), C.c_int64(N), C.c_int32(d), _p(R, _f64p),
                                  C.c_int32(T), C.c_int32(L), C.c_int32(min_leaf),
                                  C.c_int64(chunk), kind.ctypes.data_as(C.POINTER(C.c_int8)),
                                  _p(thr, _f64p), _p(mglo, _f64p), _p(mghi, _f64p),
                                  _p(leaf_off, _i64p), _p(leaf_len, _i64p), _p(leaf_ids, _i32p),
                                  _p(held, _i64p))
    return StreamForest(N, L, kind, thr, mglo, mghi, leaf_off, leaf_len, leaf_ids, held)


def stream_forest_csr(rowptr, col, val, d, R, min_leaf, chunk):
    """Conduit.hs:104-121 `forest` on a source of SVector rows (CSR arrays), chunks of `chunk` rows"""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    N = len(rowptr) - 1
    T, L, _ = R.shape
    S = (1 << (L + 1)) - 1
    kind = np.zeros((T, S), dtype=np.int8)
    thr, mglo, mghi = (np.empty((T, S), dtype=np.float64) for _ in range(3))
    leaf_off = np.zeros((T, S), dtype=np.int64)
    leaf_len = np.zeros((T, S), dtype=np.int64)
    leaf_ids = np.full((T, max(N, 1)), -1, dtype=np.int32)
    held = np.zeros(T, dtype=np.int64)
    lib().rpo_stream_forest_csr(_p(rowptr, _i64p), _p(col, _i32p), _p(val, _f64p), C.c_int64(N), C.c_int32(d),
                                _p(R, _f64p), C.c_int32(T), C.c_int32(L), C.c_int32(min_leaf),
                                C.c_int64(chunk), kind.ctypes.data_as(C.POINTER(C.c_int8)),
                                _p(thr, _f64p), _p(mglo, _f64p), _p(mghi, _f64p),
                                _p(leaf_off, _i64p), _p(leaf_len, _i64p), _p(leaf_ids, _i32p),
                                _p(held, _i64p))
    return StreamForest(N, L, kind, thr, mglo, mghi, leaf_off, leaf_len, leaf_ids, held)


def _sargs(sf):
    i8p = C.POINTER(C.c_int8)
    return (sf.kind.ctypes.data_as(i8p), _p(sf.thr, _f64p), _p(sf.mglo, _f64p), _p(sf.mghi, _f64p),
            _p(sf.leaf_off, _i64p), _p(sf.leaf_len, _i64p), _p(sf.leaf_ids, _i32p))


def stream_candidates_dense(sf, R, q, t):
    """RPTree.hs:289-314 on tree t of a streamed forest (R = its hyperplane block [T][L][d])"""
    R = np.ascontiguousarray(R, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    T, L, d = R.shape
    out = np.empty(max(sf.N, 1), dtype=np.int32)
    m = lib().rpo_stream_candidates_dense(_p(q, _f64p), C.c_int32(d), _p(R, _f64p), C.c_int32(T),
                                          C.c_int32(L), C.c_int64(sf.N), *_sargs(sf), C.c_int32(t),
                                          _p(out, _i32p), C.c_int64(len(out)))
    return out[:m].copy()


def stream_knn_dense(sf, R, X, q, k, dedup=0):
    """RPTree.hs:168-176 `knn metricL2` over a streamed forest -> (ids, dist)"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    T, L, d = R.shape
    ids = np.empty(k, dtype=np.int32)
    dist = np.empty(k, dtype=np.float64)
    m = lib().rpo_stream_knn_dense(_p(X, _f64p), C.c_int64(sf.N), C.c_int32(d), _p(q, _f64p),
                                   _p(R, _f64p), C.c_int32(T), C.c_int32(L), *_sargs(sf),
                                   C.c_int32(k), C.c_int32(int(dedup)), _p(ids, _i32p), _p(dist, _f64p))
    return ids[:m], dist[:m]


def stream_knn_h_dense(sf, R, X, q, k):
    """RPTree.hs:199-217 `knnH metricL2` over a streamed forest -> (ids, dist): whole buckets, unsorted"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    T, L, d = R.shape
    cap = 4096
    while True:
        ids = np.empty(cap, dtype=np.int32)
        dist = np.empty(cap, dtype=np.float64)
        m = lib().rpo_stream_knn_h_dense(_p(X, _f64p), C.c_int64(sf.N), C.c_int32(d), _p(q, _f64p),
                                         _p(R, _f64p), C.c_int32(T), C.c_int32(L), *_sargs(sf),
                                         C.c_int32(k), _p(ids, _i32p), _p(dist, _f64p), C.c_int64(cap))
        if m <= cap:
            return ids[:m].copy(), dist[:m].copy()
        cap = int(m)


def brute_knn_dense(X, q, k):
    X = np.ascontiguousarray(X, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    N, d = X.shape
    k = min(k, N)
    ids = np.empty(k, dtype=np.int32)
    dist = np.empty(k, dtype=np.float64)
    lib().rpo_brute_knn_dense(_p(X, _f64p), C.c_int64(N), C.c_int32(d), _p(q, _f64p),
                              C.c_int32(k), _p(ids, _i32p), _p(dist, _f64p))
    return ids, dist


# ---- topology helper (pure function of N, minLeaf, maxDepth) ----
def topology(N, L, min_leaf):
    """-> list of (level, heap, off, n, is_leaf) in DFS order.  Internal.hs:289,495,503."""
    out = []

    def go(level, heap, off, n):
        leaf = level >= L or n <= min_leaf
        out.append((level, heap, off, n, leaf))
        if not leaf:
            nh = n // 2
            go(level + 1, 2 * heap + 1, off, nh)
            go(level + 1, 2 * heap + 2, off + nh, n - nh)

    go(0, 0, 0, N)
    return out
