"""A/B of the small-shard query path (round 4: shard_ranges_kernel + knn_shard_wave_kernel) against the
round-3 one-wave kernel (knn_shard_old = 1) and the workgroup kernel (knn_wave = 0): identical answers
bit for bit, ms per batch, uncertified queries.
usage: python tools/shard_ab.py [f64|f32] [nq] [trees,trees,...]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import ctypes as C
import numpy as np
import torch
import rptree_amd as rp
from rptree_amd import _lib
dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
trees = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4, 8]
n, d, min_leaf, k = 1_000_000, 128, 128, 10
dev = torch.device("cuda:0")
X = rp.gen.normal_dense2_torch(1234, n, d, dev)
Q = rp.gen.normal_dense2_torch(4321, nq, d, dev)
if dt == "f32":
    X, Q = X.float().contiguous(), Q.float().contiguous()
torch.cuda.synchronize()
ctx = rp.default_context()
ds, qs = rp.Dataset.from_torch(ctx, X), rp.Dataset.from_torch(ctx, Q)
cfg = rp.rpTreeCfg(min_leaf, n, d)
L_ = _lib.lib()
ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def run(f, reps=10):
    best = 1e9
    tot = 0.0
    for it in range(reps + 2):
        ctx.sync()
        t0 = time.perf_counter()
        _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
        ctx.sync()
        t = time.perf_counter() - t0
        if it >= 2:
            best = min(best, t)
            tot += t
    unc, tier, ret = C.c_int64(), C.c_int32(), C.c_int64()
    _lib.check(L_.rpt_knn_last_retries(ctx._h, C.byref(ret)))
    _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
    _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
    return best * 1e3, tot / reps * 1e3, (unc.value, ret.value), tier.value, (ids.cpu().numpy().copy(), dist.cpu().numpy().copy(), cnt.cpu().numpy().copy())


for T in trees:
    _, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)
    ref = None
    for name, opts in [("round-3 kernels", {"knn_shard_old": 1}),
                       ("default", {}),
                       ("workgroup kernel", {"knn_wave": 0}),
                       ("shard kernels", {"knn_wave": 1}),
                       ("shard, half tier", {"knn_wave": 1, "knn_no_pre8": 1}),
                       ("shard, f32 tier", {"knn_wave": 1, "knn_no_pre16": 1})]:
        if T > 32 and "shard" in name:
            continue
        if dt == "f32" and "f32 tier" in name:
            continue
        for o, v in opts.items():
            ctx.set_option(o, v)
        best, mean, unc, tier, got = run(f)
        for o in opts:
            ctx.set_option(o, -1 if o == "knn_wave" else 0)
        if ref is None:
            ref = got
        same = all(np.array_equal(a, b) for a, b in zip(got, ref))
        print("%s T=%2d nq=%d %-24s best %.3f mean %.3f ms  tier %d  uncertified / retried %s  identical: %s" % (
            dt, T, nq, name, best, mean, tier, unc, same), flush=True)
    f.close()
