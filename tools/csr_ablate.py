"""C3-shaped CSR build timings per projection mode / kernel.  usage: python tools/csr_ablate.py [n] [trees]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import numpy as np
import ctypes as C
import rptree_amd as rp
from rptree_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
d, dens, min_leaf = 784, 0.19, 128
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
L_ = _lib.lib()
for name, mode, opts in [("exact, 32 per pass", rp.RPT_PROJ_EXACT, {}),
                         ("dense bf16x2 on MFMA", rp.RPT_PROJ_MFMA, {}),
                         ("fused, 32 per pass", rp.RPT_PROJ_MFMA, {"proj_csr_nodense": 1}),
                         ("exact, 16 per pass", rp.RPT_PROJ_EXACT, {"proj_narrow": 1})]:
    for k, v in opts.items():
        ctx.set_option(k, v)
    best = 1e9
    for it in range(3):
        _lib.check(L_.rpt_prof_reset(ctx._h)); _lib.check(L_.rpt_prof_enable(ctx._h, 1))
        try:
            t0 = time.perf_counter()
            f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, mode)
            ctx.sync()
            wall = (time.perf_counter() - t0) * 1e3
            f.close()
        except Exception as e:
            print("   (build failed: %s)" % str(e)[:80])
        ctx.sync()
        ms, cnt = C.c_double(), C.c_int64()
        _lib.check(L_.rpt_prof_get(ctx._h, 0, C.byref(ms), C.byref(cnt)))        # (class 0 includes the wide launches of class 4)
        if ms.value < best:
            best, bw = ms.value, wall
    print("%-20s projection %.2f ms in %d launches, build %.2f ms" % (name, best, cnt.value, bw), flush=True)
    for k in opts:
        ctx.set_option(k, 0)
