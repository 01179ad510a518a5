"""C4-shaped shard (10M x 128 f32, 8 trees per GPU, minLeaf 128 -> 17 levels, k = 10):
build + kNN on one GPU, validity checks, timings.  usage: python tools/try_c4.py [npoints] [trees]"""
import sys
import time

import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path.insert(0, _os.path.join(_ROOT, "rp-tree_amd", "python"))
sys.path.insert(0, _ROOT)
import numpy as np
import torch

import rptree_amd as rp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d, min_leaf, k, nq = 128, 128, 10, 100_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
coin = (torch.rand(N, 1, device=dev, generator=g) < 0.5).to(torch.float32) * 2.0
X = torch.randn(N, d, device=dev, dtype=torch.float32, generator=g) * 0.5 + coin
Q = X[torch.randint(0, N, (nq,), device=dev, generator=g)].clone() + 0.01
torch.cuda.synchronize()
ctx = rp.Context(0)
cfg = rp.rpTreeCfg(min_leaf, N, d)
ds = rp.Dataset.dense_device(ctx, X.data_ptr(), N, d, rp.RPT_F32, keep=X)
qs = rp.Dataset.dense_device(ctx, Q.data_ptr(), nq, d, rp.RPT_F32, keep=Q)
_, R = rp.gen.forest_hyperplanes(99, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
print("N %d d %d T %d L %d" % (N, d, T, cfg.fpMaxTreeDepth), flush=True)
for it in range(3):
    t0 = time.perf_counter()
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)
    ctx.sync()
    t1 = time.perf_counter()
    print("build %d: %.2f ms = %.1f M vectors/s" % (it, (t1 - t0) * 1e3, N / (t1 - t0) / 1e6), flush=True)
    if it < 2:
        f.close()
perm = f.perm
for t in range(min(T, 2)):
    assert np.array_equal(np.bincount(perm[t], minlength=N), np.ones(N, dtype=np.int64))
print("perm rows are permutations; stats", f.stats(), flush=True)
ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
from rptree_amd import _lib
L_ = _lib.lib()
for it in range(2):
    t0 = time.perf_counter()
    _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
    ctx.sync()
    t1 = time.perf_counter()
    print("knn %d: %.2f ms = %.2f M queries/s" % (it, (t1 - t0) * 1e3, nq / (t1 - t0) / 1e6), flush=True)
print("mean distance of the best hit %.4f (queries are data points + 0.01)" % float(dist[:, 0].mean()))
