"""One 4-tree shard of C2 (what one of 8 GPUs holds): build and kNN repeated, for kernel traces.
usage: python tools/shard4.py [trees] [reps]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import torch
import rptree_amd as rp
from rptree_amd import _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, d, min_leaf, k, nq = 1_000_000, 128, 128, 10, 10_000
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
torch.cuda.synchronize()
f = None
for it in range(3):
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)
    _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
ctx.sync()
import os
if os.environ.get("SHARD4_PROF"):          # the HIP-event spans bench.py records (rpt_prof_enable): their cost
    _lib.check(L_.rpt_prof_reset(ctx._h))
    _lib.check(L_.rpt_prof_enable(ctx._h, 1))
t0 = time.perf_counter()
for it in range(reps):
    rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA).close()
ctx.sync()
tb = (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
for it in range(reps):
    _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
    ctx.sync()
tq = (time.perf_counter() - t0) / reps * 1e3
print("trees %d: build %.3f ms, knn %.3f ms per %d queries" % (T, tb, tq, nq), flush=True)
