"""kNN time per shard size (trees per GPU) at the C2 shape: rpt_knn_dev on T = 4, 8, 16, 32 trees.
Run once per library / RPT_KNN_WAVE setting (both are read at process start)."""
import os
import sys
import time

sys.path.insert(0, "rp-tree_amd/python")
sys.path.insert(0, ".")
import numpy as np
import torch

import rptree_amd as rp
from rptree_amd import _lib

N, d, min_leaf, k, nq = 1_000_000, 128, 128, 10, 10_000
dt = os.environ.get("SWEEP_DTYPE", "f64")
tdt = torch.float64 if dt == "f64" else torch.float32
rdt = rp.RPT_F64 if dt == "f64" else rp.RPT_F32
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1234)


def synth(n):
    mu = (torch.rand(n, generator=g, device=dev) < 0.5).to(tdt) * 2.0
    return (torch.randn(n, d, generator=g, device=dev, dtype=tdt) * 0.5 + mu[:, None]).contiguous()


X, Q = synth(N), synth(nq)
ctx = rp.Context(0)
L_ = _lib.lib()
ds = rp.Dataset.dense_device(ctx, X.data_ptr(), N, d, rdt, keep=X)
qs = rp.Dataset.dense_device(ctx, Q.data_ptr(), nq, d, rdt, keep=Q)
cfg = rp.rpTreeCfg(min_leaf, N, d)
ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
tag = "%s wave=%s %s" % (os.path.basename(os.environ.get("RPTREE_HIP_LIB", "current")),
                         os.environ.get("RPT_KNN_WAVE", "auto"), dt)
for T in (4, 8, 16, 32):
    _, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)
    best = 1e9
    for it in range(6):
        ctx.sync()
        t0 = time.perf_counter()
        _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, rp.RPT_KNN_KEEP_DUPLICATES,
                                  ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    chk = int(ids.to(torch.int64).sum().item()) ^ int((dist * 1e6).to(torch.int64).sum().item())
    print("%-28s T=%2d knn %.3f ms (%.2f M q/s) checksum %d" % (tag, T, best * 1e3, nq / best / 1e6, chk))
    f.close()
