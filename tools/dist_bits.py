"""Are the kNN distances the device returns the reference's bits?  Device (every f64 query path)
against the oracle's metricDDL2 (left fold of (u - v) ** 2, libm pow) on continuous data."""
import sys, os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "rp-tree_amd", "python")); sys.path.insert(0, _ROOT)
import numpy as np
import rptree_amd as rp
from oracle import oracle as o

n, d, T, ml, k, nq = 60000, 128, 8, 64, 10, 300
X = o.data_normal_dense2(1234, n, d)
Q = o.data_normal_dense2(4321, nq, d)
L, _, pnz = o.tree_cfg(ml, n, d)
R, _ = o.forest_hyperplanes(7, T, L, pnz, d)
ctx = rp.default_context()
f = rp.forestBatch(7, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R)
fo = o.Forest(n, d, R, L, ml, f.perm, f.thr, f.mglo, f.mghi)
for name, opts, kw in [("f32 prefilter (default)", {}, {}), ("all-f64 kernel", {"knn_no_pre32": 1}, {}),
                       ("each id once", {}, {"dedup": True}), ("general path", {"knn_general": 1}, {}),
                       ("one wave per query", {"knn_wave": 1}, {}), ("wave, all-f64", {"knn_wave": 1, "knn_no_pre32": 1}, {})]:
    for a, b in opts.items():
        ctx.set_option(a, b)
    ids, dist, cnt = rp.knnBatch(k, f, Q, **kw)
    wi, wd, wc = o.knn_dense_batch(fo, X, Q, k, dedup=1 if kw.get("dedup") else 0, threads=8)
    same_ids = sum(np.array_equal(ids[i, :cnt[i]], wi[i, :wc[i]]) for i in range(nq))
    same_bits = sum(np.array_equal(dist[i, :cnt[i]], wd[i, :wc[i]]) for i in range(nq))
    print("%-26s ids identical %d/%d   distances bit-identical %d/%d" % (name, same_ids, nq, same_bits, nq), flush=True)
    for a in opts:
        ctx.set_option(a, -1 if a == "knn_wave" else 0)
