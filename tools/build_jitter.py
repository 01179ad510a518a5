"""Per-build wall times of repeated C2 forest builds (spots host-side stalls).
usage: python tools/build_jitter.py [builds] [npoints] [trees]"""
import sys
import time

sys.path.insert(0, "rp-tree_amd/python")
sys.path.insert(0, ".")
import numpy as np
import torch

import bench
import rptree_amd as rp

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
d, min_leaf = 128, 128
dev = torch.device("cuda", 0)
X = bench.synth(N, d, 1234, dev)
ctx = rp.Context(0)
cfg = rp.rpTreeCfg(min_leaf, N, d)
ds = rp.Dataset.dense_device(ctx, X.data_ptr(), N, d, rp.RPT_F64, keep=X)
_, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
ts = []
f = None
for it in range(nb):
    if f is not None:
        f.close()
    t0 = time.perf_counter()
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)
    ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.2f" % t for t in ts))
a = np.array(ts[2:])
print("median %.3f ms, max %.3f ms, builds over 2x median: %d of %d" % (
    np.median(a), a.max(), int((a > 2 * np.median(a)).sum()), len(a)))
