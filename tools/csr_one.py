"""Two C3-shaped CSR builds in one projection mode (for profiling).  usage: csr_one.py [exact|fused]"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import numpy as np
import rptree_amd as rp

n, T, d, dens, min_leaf = 1_000_000, 32, 784, 0.19, 128
rng = np.random.default_rng(1)
cols, counts = [], []
for r0 in range(0, n, 20000):
    m = rng.random((min(20000, n - r0), d)) < dens
    counts.append(m.sum(axis=1)); cols.append(np.nonzero(m)[1].astype(np.int32))
col = np.concatenate(cols)
rowptr = np.zeros(n + 1, dtype=np.int64); rowptr[1:] = np.cumsum(np.concatenate(counts))
val = 1.0 - rng.random(rowptr[-1])
cfg = rp.rpTreeCfg(min_leaf, n, d)
ctx = rp.default_context()
ds = rp.Dataset.csr(ctx, rowptr, col, val, d)
_, R = rp.gen.forest_hyperplanes(5, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
mode = rp.RPT_PROJ_MFMA if (len(sys.argv) > 1 and sys.argv[1] == "fused") else rp.RPT_PROJ_EXACT
for it in range(2):
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, mode)
    ctx.sync()
    f.close()
