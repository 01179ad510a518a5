"""Long f32 rows through the dense prefilter ("WIDE") instantiation of the workgroup query kernel:
2 M x 768 f32, 16 trees, k = 50 (a C5-shaped shard in f32).  usage: python tools/try_wide.py [n] [d] [k]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import ctypes as C
import torch
import rptree_amd as rp
from rptree_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
T, min_leaf, nq = 16, 256, 20_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(99)
X = torch.randn(n, d, device=dev, dtype=torch.float32, generator=g)
X /= X.norm(dim=1, keepdim=True)
qi = torch.randint(0, n, (nq,), device=dev, generator=g)
Q = (X[qi] * 1.001 + 0.003).contiguous()
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
t0 = time.perf_counter()
for it in range(5):
    _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
    ctx.sync()
tq = (time.perf_counter() - t0) / 5 * 1e3
tier, unc, cand = C.c_int32(), C.c_int64(), C.c_int64()
_lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
_lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
_lib.check(L_.rpt_knn_last_candidates(ctx._h, C.byref(cand)))
print("n %d d %d k %d: knn %.3f ms per %d queries, tier %d, uncertified %d, candidates/query %.0f" % (
    n, d, k, tq, nq, tier.value, unc.value, cand.value / nq), flush=True)
