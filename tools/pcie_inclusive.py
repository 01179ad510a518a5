"""PCIe-inclusive timing of the C2 path through the host-buffer boundary (rpt_dataset_dense_host ->
rpt_forest_build -> rpt_forest_get_perm / get_nodes; rpt_knn_host).  DESIGN.md section 6."""
import sys
import time

sys.path.insert(0, "rp-tree_amd/python")
sys.path.insert(0, ".")
import numpy as np

import rptree_amd as rp
from oracle import oracle as o

N, d, T, min_leaf, k, nq = 1_000_000, 128, 32, 128, 10, 10_000
X = o.data_normal_dense2(1234, N, d)
Q = o.data_normal_dense2(4321, nq, d)
ctx = rp.Context(0)
cfg = rp.rpTreeCfg(min_leaf, N, d)
_, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
for it in range(3):
    t0 = time.perf_counter()
    ds = rp.Dataset.dense(ctx, X)                      # H2D of the 1.024 GB point matrix
    t1 = time.perf_counter()
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)
    ctx.sync()
    t2 = time.perf_counter()
    perm = f.perm                                      # D2H 128 MB
    thr = f.thr
    t3 = time.perf_counter()
    ids, dist, cnt = rp.knnBatch(k, f, Q)              # H2D queries, D2H results
    t4 = time.perf_counter()
    print("upload X %.1f ms | build %.2f ms | download perm+nodes %.1f ms | knn (host buffers) %.2f ms"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
    print("  PCIe-inclusive: build %.1f M vectors/s (upload + build + download), knn %.2f M queries/s"
          % (N / (t3 - t0) / 1e6, nq / (t4 - t3) / 1e6))
    f.close()
    ds.close()
