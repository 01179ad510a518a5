"""C4 as ONE GPU holds it whole (10 M x 128 f32, up to 64 trees): kNN time per batch against the number of
trees, uncertified queries, ranking tier; optional option sweeps.  usage: python tools/c4_whole.py [nq] [T,T,..] [opt=v,opt=v ...]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import ctypes as C
import torch
import rptree_amd as rp
from rptree_amd import _lib, gen
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
trees = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [32, 64]
variants = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[3:]] or [{}]
n, d, min_leaf, k = 10_000_000, 128, 128, 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
coin = (torch.rand(n, 1, device=dev, generator=g) < 0.5).float() * 2.0
Xd = torch.randn(n, d, device=dev, dtype=torch.float32, generator=g) * 0.5 + coin
del coin
qi = torch.randint(0, n, (nq,), device=dev, generator=g)
Qd = (Xd[qi] * 1.001 + 0.003).contiguous()
torch.cuda.synchronize()
ctx = rp.default_context()
L_ = _lib.lib()
ds, qs = rp.Dataset.from_torch(ctx, Xd), rp.Dataset.from_torch(ctx, Qd)
cfg = rp.rpTreeCfg(min_leaf, n, d)
ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
for T in trees:
    _, R = gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
    ref = None
    for opts in variants:
        f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_AUTO)     # fresh tier state
        for o, v in opts.items():
            ctx.set_option(o, int(v))
        best = 1e9
        for it in range(5):
            ctx.sync()
            t0 = time.perf_counter()
            _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        unc, tier, cand, ret = C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(L_.rpt_knn_last_retries(ctx._h, C.byref(ret)))
        _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
        _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
        _lib.check(L_.rpt_knn_last_candidates(ctx._h, C.byref(cand)))
        got = (ids.cpu(), dist.cpu())
        if ref is None:
            ref = got
        same = bool((got[0] == ref[0]).all() and (got[1] == ref[1]).all())
        print("T=%2d nq=%d %-30s %.3f ms  tier %d  uncertified %5d  retried %5d  candidates/query %.0f  same as first: %s" % (
            T, nq, opts, best * 1e3, tier.value, unc.value, ret.value, cand.value / nq, same), flush=True)
        for o in opts:
            ctx.set_option(o, 0)
        f.close()
