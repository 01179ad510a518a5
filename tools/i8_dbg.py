"""int8 tier at C2 sizes: does the shadow get built, which tier answers (debug aid)
usage: RPT_DEBUG_HOST=1 python tools/i8_dbg.py [n] [trees]"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import ctypes as C
import torch
import rptree_amd as rp
from rptree_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d, min_leaf, k, nq = 128, 128, 10, 2000
dev = torch.device("cuda:0")
X = rp.gen.normal_dense2_torch(1234, n, d, dev)
Q = rp.gen.normal_dense2_torch(4321, nq, d, dev)
torch.cuda.synchronize()
ctx = rp.default_context()
ds, qs = rp.Dataset.from_torch(ctx, X), rp.Dataset.from_torch(ctx, Q)
cfg = rp.rpTreeCfg(min_leaf, n, d)
_, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
L_ = _lib.lib()
ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)
for it in range(2):
    _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
    ctx.sync()
    unc, tier = C.c_int64(), C.c_int32()
    _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
    _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
    print("call %d: tier %d uncertified %d" % (it, tier.value, unc.value), flush=True)
