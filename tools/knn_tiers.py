"""kNN ranking tiers at C2 (32 trees and a 4-tree shard): f32 shadow vs the half and int8 shadows with
several numbers of kept entries: ms per 10 000 queries, uncertified queries, identity of the answers.
usage: python tools/knn_tiers.py"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import ctypes as C
import numpy as np
import torch
import rptree_amd as rp
from rptree_amd import _lib
n, d, min_leaf, k, nq = 1_000_000, 128, 128, 10, 10_000
dev = torch.device("cuda:0")
X = rp.gen.normal_dense2_torch(1234, n, d, dev)
Q = rp.gen.normal_dense2_torch(4321, nq, d, dev)
torch.cuda.synchronize()
ctx = rp.default_context()
ds, qs = rp.Dataset.from_torch(ctx, X), rp.Dataset.from_torch(ctx, Q)
cfg = rp.rpTreeCfg(min_leaf, n, d)
L_ = _lib.lib()
ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for T in (32, 4):
    _, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
    ref = None
    for name, opts in [("f32 shadow", {"knn_no_pre16": 1}), ("half, keep 18", {"knn_no_pre8": 1, "knn_kp16": 18}),
                       ("half, keep 26", {"knn_no_pre8": 1, "knn_kp16": 26}),
                       ("int8, keep 14", {"knn_kp8": 14}), ("int8, keep 18", {"knn_kp8": 18}),
                       ("int8, keep 22", {"knn_kp8": 22}), ("int8, keep 26", {"knn_kp8": 26}),
                       ("int8, keep 34", {"knn_kp8": 34}), ("int8, keep 42", {"knn_kp8": 42})]:
        f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)   # fresh tier state
        for o, v in opts.items():
            ctx.set_option(o, v)
        best = 1e9
        for it in range(6):
            ctx.sync()
            t0 = time.perf_counter()
            _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        unc, tier = C.c_int64(), C.c_int32()
        _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
        _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
        got = (ids.cpu().numpy().copy(), dist.cpu().numpy().copy())
        if ref is None:
            ref = got
        same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        print("T=%2d %-14s %.3f ms  tier %d  uncertified %5d  identical to the f32 tier: %s" % (
            T, name, best * 1e3, tier.value, unc.value, same), flush=True)
        for o in opts:
            ctx.set_option(o, 0)
        f.close()
