"""Vote-threshold sweep (the reference's commented-out counts / keepCounts,
RPTree.hs:464-478): recall@k against brute force, points ranked per query and queries/s for
v = 1 .. vmax on a forest of T trees.  The trade a user tunes: a higher v ranks fewer points
(cheaper) but drops true neighbours that few trees found.
usage: python tools/vote_sweep.py [n] [d] [trees] [minLeaf] [k] [nq] [vmax]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import numpy as np
import rptree_amd as rp

a = [int(v) for v in sys.argv[1:]]
n, d, T, min_leaf, k, nq, vmax = (a + [200_000, 32, 32, 64, 10, 2000, 6][len(a):])[:7]
X = rp.gen.normal_dense2(1234, n, d)
Q = rp.gen.normal_dense2(4321, nq, d)
cfg = rp.rpTreeCfg(min_leaf, n, d)
ctx = rp.default_context()
f = rp.forestBatch(7, cfg.fpMaxTreeDepth, min_leaf, T, cfg.fpProjNzDensity, d, X, ctx=ctx)
bi, bd = rp.bruteKnn(f, Q, k)
print("n=%d d=%d trees=%d minLeaf=%d depth=%d k=%d queries=%d" % (n, d, T, min_leaf, cfg.fpMaxTreeDepth, k, nq))
print("%4s %10s %14s %12s %12s" % ("v", "recall@k", "answered(<k)", "ms/batch", "queries/s"))
for v in [0] + list(range(1, vmax + 1)):
    # v = 0: the reference's knn with each id once (RPT_KNN_DEDUP) for comparison; v >= 1: voting
    kw = dict(dedup=True) if v == 0 else dict(vote=v)
    rp.knnBatch(k, f, Q, **kw)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        ids, dist, cnt = rp.knnBatch(k, f, Q, **kw)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    hit = sum(len(set(ids[i, :cnt[i]].tolist()) & set(bi[i].tolist())) for i in range(nq))
    print("%4s %10.4f %14d %12.3f %12.0f" % ("dedup" if v == 0 else v, hit / float(nq * k), int((cnt < k).sum()), ms,
                                              nq / ms * 1e3), flush=True)
