"""C2 builds in exact-order mode: build / projection time (A/B of the exact kernel)."""
import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import ctypes as C
import torch
import rptree_amd as rp
from rptree_amd import _lib
n, d, T, min_leaf = 1_000_000, 128, 32, 128
dev = torch.device("cuda:0")
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.float64
X = rp.gen.normal_dense2_torch(1234, n, d, dev).to(dt).contiguous()
torch.cuda.synchronize()
ctx = rp.default_context()
ds = rp.Dataset.from_torch(ctx, X)
cfg = rp.rpTreeCfg(min_leaf, n, d)
_, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
L_ = _lib.lib()
for it in range(3):
    _lib.check(L_.rpt_prof_reset(ctx._h)); _lib.check(L_.rpt_prof_enable(ctx._h, 1))
    t0 = time.perf_counter()
    for _ in range(3):
        rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_EXACT).close()
    ctx.sync()
    wall = (time.perf_counter() - t0) / 3 * 1e3
    ms, cnt = C.c_double(), C.c_int64()
    _lib.check(L_.rpt_prof_get(ctx._h, 0, C.byref(ms), C.byref(cnt)))
    print("exact %s build %.3f ms  projection %.3f ms in %d launches (%.4f ms each)" % (str(dt), wall, ms.value / 3, cnt.value // 3, ms.value / cnt.value), flush=True)
