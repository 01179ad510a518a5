"""C3-shaped check (BASELINE configs[2]): sparse 784-dim rows (density 0.19, values U(0,1]),
hyperplane density by rpTreeCfg.  Times the build, verifies validity and the cut property."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import numpy as np
import rptree_amd as rp
from rptree_amd import _lib
import ctypes as C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d, dens, min_leaf = 784, 0.19, 128
rng = np.random.default_rng(1)
cols, counts = [], []
for r0 in range(0, n, 20000):            # Bernoulli support per entry (sparse, Gen.hs:178-195)
    m = rng.random((min(20000, n - r0), d)) < dens
    counts.append(m.sum(axis=1))
    cols.append(np.nonzero(m)[1].astype(np.int32))
col = np.concatenate(cols)
rowptr = np.zeros(n + 1, dtype=np.int64); rowptr[1:] = np.cumsum(np.concatenate(counts))
val = 1.0 - rng.random(rowptr[-1])
cfg = rp.rpTreeCfg(min_leaf, n, d)
ctx = rp.default_context()
ds = rp.Dataset.csr(ctx, rowptr, col, val, d)
_, R = rp.gen.forest_hyperplanes(5, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
L_ = _lib.lib()
for it in range(3):
    _lib.check(L_.rpt_prof_reset(ctx._h)); _lib.check(L_.rpt_prof_enable(ctx._h, 1))
    t0 = time.perf_counter()
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_AUTO)
    ctx.sync(); dt = time.perf_counter() - t0
    ms, cnt = C.c_double(), C.c_int64()
    _lib.check(L_.rpt_prof_get(ctx._h, 0, C.byref(ms), C.byref(cnt)))
    print("build %.2f ms  (N=%d nnz=%d T=%d L=%d)  projection %.2f ms in %d launches" % (dt*1e3, n, rowptr[-1], T, cfg.fpMaxTreeDepth, ms.value, cnt.value))
    if it < 2: f.close()
for t in range(T):
    assert np.array_equal(np.bincount(f.perm[t], minlength=n), np.ones(n, dtype=np.int64))
P = f.proj()
topo = [r for r in f.topology() if not r[4]]
for (level, heap, off, m, _) in topo[:10] + topo[-10:]:
    for t in range(T):
        nh = m // 2
        left = P[t, level][f.perm[t, off:off+nh]]; right = P[t, level][f.perm[t, off+nh:off+m]]
        assert left.max() <= f.thr[t, heap] == right.min()
# projection exactness vs the reference order on a few (row, hyperplane) pairs
for (t, l, i) in [(0, 0, 0), (1, 3, 17), (T-1, cfg.fpMaxTreeDepth-1, n-1)]:
    a, b = rowptr[i], rowptr[i+1]
    acc = 0.0
    for j in range(b-1, a-1, -1):
        if R[t, l, col[j]] != 0: acc = val[j] * R[t, l, col[j]] + acc
    assert P[t, l, i] == acc, (P[t, l, i], acc)
print("valid; stats", f.stats())

# kNN over the sparse forest (general path: CSR rows, true Euclidean distance): 2000 queries
nq = 2000
qr = rowptr[:nq + 1].copy()
qc, qv = col[:qr[-1]].copy(), val[:qr[-1]].copy()
for it in range(2):
    t0 = time.perf_counter()
    ids, dist, cnt = rp.knnBatch(10, f, (qr, qc, qv, d))
    dt = time.perf_counter() - t0
    print("knn %d queries: %.2f ms = %.3f M queries/s (host call, incl. query upload)" % (nq, dt * 1e3, nq / dt / 1e6))
# every row finds itself; the CSR distance |q|^2 + sum((x-q)^2 - q^2) cancels to ~1e-8 |q|, not 0
assert (ids[:, 0] == np.arange(nq)).all() and (dist[:, 0] < 1e-6).all()
_lib.check(L_.rpt_prof_reset(ctx._h)); _lib.check(L_.rpt_prof_enable(ctx._h, 1))
ids, dist, cnt = rp.knnBatch(10, f, (qr, qc, qv, d))
for name, which in (("query projections + traversal plan", 2), ("distance / top-k kernel", 3)):
    ms, c = C.c_double(), C.c_int64()
    _lib.check(L_.rpt_prof_get(ctx._h, which, C.byref(ms), C.byref(c)))
    print("  %-36s %.2f ms in %d spans" % (name, ms.value, c.value))

# 10 000 fresh queries of the data's distribution (Bernoulli(0.19) support, U(0,1] values): the
# (u16 column, f32 value) prefilter applies.  (Queries that ARE data points, as above, tie with
# their own copies in every tree across the prefilter's cut and are answered exactly; after a
# batch with > 25 % such queries the forest stops trying.)
nq2 = 10000
rq = np.random.default_rng(3)
mq = rq.random((nq2, d)) < dens
qr2 = np.zeros(nq2 + 1, dtype=np.int64); qr2[1:] = np.cumsum(mq.sum(axis=1))
qc2 = np.nonzero(mq)[1].astype(np.int32)
qv2 = 1.0 - rq.random(qr2[-1])
keep = None
for label, opts in (("half table", {}), ("f32 prefilter", {"knn_csr_pre32": 1, "knn_no_pre16": 1}),
                    ("all exact", {"knn_no_pre32": 1})):
    for kk, vv in opts.items():
        ctx.set_option(kk, vv)
    g = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_AUTO)   # a fresh forest: prefilter state
    rp.knnBatch(10, g, (qr2, qc2, qv2, d))
    _lib.check(L_.rpt_prof_reset(ctx._h)); _lib.check(L_.rpt_prof_enable(ctx._h, 1))
    t0 = time.perf_counter()
    i2, d2, c2 = rp.knnBatch(10, g, (qr2, qc2, qv2, d))
    dt = time.perf_counter() - t0
    ms, c = C.c_double(), C.c_int64()
    _lib.check(L_.rpt_prof_get(ctx._h, 3, C.byref(ms), C.byref(c)))
    _lib.check(L_.rpt_prof_enable(ctx._h, 0))
    tier = C.c_int32(-1)
    _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
    print("knn %d queries, %s (tier %d): %.2f ms = %.3f M queries/s (host call); distance / top-k kernel %.2f ms = %.3f M queries/s; uncertified %d"
          % (nq2, label, tier.value, dt * 1e3, nq2 / dt / 1e6, ms.value, nq2 / ms.value / 1e3, rp.knn_last_uncertified(ctx)))
    if keep is None:
        keep = (i2.copy(), d2.copy(), c2.copy())
    else:
        assert np.array_equal(keep[0], i2) and np.array_equal(keep[1], d2) and np.array_equal(keep[2], c2)
    g.close()
    for kk in opts:
        ctx.set_option(kk, 0)
print("prefiltered answers identical to the exact kernel's")
