"""C2 builds in MFMA mode under a ctx option sweep: projection / wide-launch / split times.
usage: python tools/proj_times.py option v0 v1 v2 ..."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import ctypes as C
import numpy as np
import torch
import rptree_amd as rp
from rptree_amd import _lib

import os
n, d, T, min_leaf = 1_000_000, 128, int(os.environ.get("TREES", "32")), 128
opt = sys.argv[1] if len(sys.argv) > 1 else "tune2"
vals = [int(v) for v in sys.argv[2:]] or [0]
dev = torch.device("cuda:0")
X = rp.gen.normal_dense2_torch(1234, n, d, dev)
torch.cuda.synchronize()
ctx = rp.default_context()
ds = rp.Dataset.from_torch(ctx, X)
cfg = rp.rpTreeCfg(min_leaf, n, d)
_, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
L_ = _lib.lib()
def get(which):
    ms, cnt = C.c_double(), C.c_int64()
    _lib.check(L_.rpt_prof_get(ctx._h, which, C.byref(ms), C.byref(cnt)))
    return ms.value, cnt.value
for v in vals:
    ctx.set_option(opt, v)
    for it in range(2):
        rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA).close()
    ctx.sync()
    _lib.check(L_.rpt_prof_reset(ctx._h)); _lib.check(L_.rpt_prof_enable(ctx._h, 1))
    K = int(os.environ.get("BUILDS", "5"))
    t0 = time.perf_counter()
    for it in range(K):
        rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA).close()
    ctx.sync()
    wall = (time.perf_counter() - t0) / K * 1e3
    p, w, s_ = get(0), get(4), get(1)
    _lib.check(L_.rpt_prof_enable(ctx._h, 0))
    print("%s=%-3d build %.3f ms (with events)  projection %.3f  wide launch avg %.4f ms (%d)  split %.3f" % (
        opt, v, wall, p[0] / K, w[0] / max(w[1], 1), w[1] // K, s_[0] / K), flush=True)
