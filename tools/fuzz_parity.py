"""Randomised parity sweep: forests of random shape (points, dimension, trees, minLeaf, depth,
tie-heavy or continuous data, dense or sparse) built on the device in exact-order mode and by the
oracle; perm, thresholds, margins, kNN ids must be identical.
usage: python tools/fuzz_parity.py [seconds] [seed] [heavy]
heavy: shapes that take the round-2 paths — 131 072+ points x 128 (the split streams on 16-bit
codes), SVector rows of 600-900 dimensions (32 hyperplanes per CSR pass in two column halves), CSR
kNN — more often."""
import sys
import time

import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path.insert(0, _os.path.join(_ROOT, "rp-tree_amd", "python"))
sys.path.insert(0, _ROOT)
import numpy as np

import rptree_amd as rp
from oracle import oracle as o

def run(budget, seed, ctx=None, verbose=True, heavy=False):
    """-> number of cases checked; raises AssertionError with the failing case's description"""
    rng = np.random.default_rng(seed)
    ctx = ctx or rp.Context(0)
    t_end = time.time() + budget
    n_cases = 0
    while time.time() < t_end:
        n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, 60000),
                            rng.integers(60000, 300000)], p=[0.1, 0.35, 0.45, 0.1]))
        d = int(rng.choice([1, 2, 3, 8, 16, 33, 64, 128, 130]))
        T = int(rng.integers(1, 6))
        force_kind = None
        if heavy:
            r = rng.random()
            if r < 0.3:      # code path: all levels streamed needs N >= 2^17, the exact kernel d = 128
                n, d = int(rng.integers(131072, 400000)), 128
                force_kind = str(rng.choice(["cont", "ties", "const"]))
            elif r < 0.6:    # CSR in two column halves
                n, d, force_kind = int(rng.integers(2000, 40000)), int(rng.choice([600, 784, 900])), "sparse"
        min_leaf = int(rng.choice([0, 1, 2, 5, 20, 100, 300, 2000]))
        Lcfg, _, pnz = o.tree_cfg(max(min_leaf, 1), max(n, 2), max(d, 2))
        L = int(np.clip(rng.choice([Lcfg, Lcfg - 2, Lcfg + 2, 3]), 1, 18))
        kind = force_kind or rng.choice(["cont", "ties", "const", "sparse"])
        if n * L > (8_000_000 if heavy else 3_000_000):
            continue
        cseed = int(rng.integers(1, 1 << 30))
        R, _ = o.forest_hyperplanes(cseed, T, L, float(rng.choice([pnz, 1.0, 0.3])), d)
        desc = "n=%d d=%d T=%d minLeaf=%d L=%d %s seed=%d" % (n, d, T, min_leaf, L, kind, cseed)
        if kind == "sparse":
            rowptr, col, val = o.data_normal_sparse2(cseed, n, d, 0.3 if d < 500 else 0.1)
            fo = o.forest_build_csr(rowptr, col, val, d, R, min_leaf)
            f = rp.forestBatch(0, L, min_leaf, T, 0, d, (rowptr, col, val, d), ctx=ctx, hyperplanes=R)
        else:
            X = o.data_normal_dense2(cseed, n, d)
            if kind == "ties":
                X = np.round(X * float(rng.choice([1, 2, 10])))
            elif kind == "const":
                X[:] = 1.0
                X[: n // 3] = rng.standard_normal((n // 3, d))
            fo = o.forest_build_dense(X, R, min_leaf)
            f = rp.forestBatch(0, L, min_leaf, T, 0, d, X, ctx=ctx, hyperplanes=R)
        assert np.array_equal(f.perm, fo.perm), "perm: " + desc
        assert np.array_equal(f.thr, fo.thr, equal_nan=True), "thr: " + desc
        assert np.array_equal(f.mglo, fo.mglo, equal_nan=True), "mglo: " + desc
        assert np.array_equal(f.mghi, fo.mghi, equal_nan=True), "mghi: " + desc
        if kind == "sparse" and heavy:
            # SVector queries: the fused CSR kernel against the unfused general path (bit-equal)
            k = int(rng.choice([1, 5, 10]))
            nqq = min(n, 8)
            qq = (rowptr[:nqq + 1].copy(), col[:rowptr[nqq]].copy(), val[:rowptr[nqq]] * 1.01, d)
            a = rp.knnBatch(k, f, qq)
            old = ctx.set_option("knn_general", 1)
            try:
                b = rp.knnBatch(k, f, qq)
            finally:
                ctx.set_option("knn_general", old)
            for x, y in zip(a, b):
                assert np.array_equal(x, y), "csr knn fused vs general: " + desc
        if kind != "sparse":
            k = int(rng.choice([1, 3, 10, 70]))
            qs = X[rng.integers(0, n, 4)] + (0.0 if kind != "cont" else 0.01)
            ids, dist, cnt = rp.knnBatch(k, f, qs)
            for i in range(4):
                wi, wd = o.knn_dense(fo, X, qs[i], k)
                assert np.array_equal(ids[i, :cnt[i]], wi), "knn ids: " + desc
        f.close()
        n_cases += 1
        if verbose and n_cases % (5 if heavy else 25) == 0:
            print("%d cases ok, last: %s" % (n_cases, desc), flush=True)
    return n_cases


if __name__ == "__main__":
    n_ok = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0,
               int(sys.argv[2]) if len(sys.argv) > 2 else 1, heavy=len(sys.argv) > 3)
    print("fuzz ok: %d cases" % n_ok)
