"""C5-shaped projection batch (768-dim bf16 rows, 16 trees x 16 levels = 256 hyperplanes
per GPU): the bf16x3 matrix-pipe kernel against the f32-MFMA kernel on converted inputs.
usage: python tools/try_c5.py [npoints] [dim] [columns]"""
import ctypes as C
import os
import sys
import time

import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path.insert(0, _os.path.join(_ROOT, "rp-tree_amd", "python"))
sys.path.insert(0, _ROOT)
import numpy as np
import torch

import rptree_amd as rp
from rptree_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
Cn = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
X = torch.randn(N, d, device=dev, dtype=torch.float32, generator=g)
X = (X / X.norm(dim=1, keepdim=True)).to(torch.bfloat16).contiguous()
rng = np.random.default_rng(5)
R = rng.standard_normal((Cn, d)) * (rng.random((Cn, d)) < 0.3466)
P = torch.empty((Cn, N), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ctx = rp.Context(0)
L_ = _lib.lib()
ds = rp.Dataset.dense_device(ctx, X.data_ptr(), N, d, rp.RPT_BF16, keep=X)
Rc = np.ascontiguousarray(R)
res = {}
for name, env in (("bf16x3", None), ("f32-mfma", "1")):
    ctx.set_option("proj_bf16_f32", 1 if env else 0)
    best = 1e9
    for it in range(4):
        ctx.sync()
        t0 = time.perf_counter()
        _lib.check(L_.rpt_project_dev(ctx._h, ds._h, Rc.ctypes.data_as(C.c_void_p), Cn,
                                      rp.RPT_PROJ_MFMA, P.data_ptr()))
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    res[name] = P.clone()
    torch.cuda.synchronize()   # the clone runs on torch's stream, the next kernels on the ctx stream
    flop = 2.0 * N * d * Cn
    byt = N * d * 2 + N * Cn * 4
    print("%-9s %8.3f ms  %.1f TFLOP/s (useful)  X+P once = %.2f GB -> %.2f TB/s equivalent"
          % (name, best * 1e3, flop / best / 1e12, byt / 1e9, byt / best / 1e12), flush=True)
if len(res) == 2:   # whole-output comparison of the two kernels
    a, b = res["bf16x3"], res["f32-mfma"]
    for c in range(0, Cn, 64):
        diff = (a[c:c + 64] - b[c:c + 64]).abs() > 1e-4
        if diff.any():
            cols = diff.any(dim=1).nonzero().flatten() + c
            pts = diff.any(dim=0).nonzero().flatten()
            print("  kernels differ: columns %s, %d points in [%d, %d], tiles(256) %d..%d" % (
                cols.tolist()[:8], pts.numel(), pts.min().item(), pts.max().item(),
                pts.min().item() // 256, pts.max().item() // 256))
Rd = torch.from_numpy(R).to(dev)
rn = torch.from_numpy(np.linalg.norm(R, axis=1)).to(dev)[:, None]
for name, Pn in res.items():
    worst, bad = 0.0, 0
    for w0 in sorted({0, (N // 2) // 4096 * 4096, max(N - 4096, 0)}):
        xs = X[w0:w0 + 4096].double()
        ref = (xs @ Rd.T).T
        e = (Pn[:, w0:w0 + 4096].double() - ref).abs() / (xs.norm(dim=1)[None, :] * rn)
        worst = max(worst, e.max().item())
        nb = int((e > 1e-5).sum().item())
        bad += nb
        if nb:
            idx = (e > 1e-5).nonzero()
            print("  %s window %d: %d bad entries, columns %d..%d, points %d..%d" % (
                name, w0, nb, idx[:, 0].min().item(), idx[:, 0].max().item(),
                w0 + idx[:, 1].min().item(), w0 + idx[:, 1].max().item()))
    print("%-9s max |err| / (|x||r|) over three 4096-point windows = %.2e (%d entries above 1e-5)"
          % (name, worst, bad))

# ---- a C5-shaped shard end to end: 16 trees, minLeaf 256, k = 50 ----
if len(sys.argv) > 4 and sys.argv[4] == "build":
    T, min_leaf, k, nq = 16, 256, 50, 10_000
    cfg = rp.rpTreeCfg(min_leaf, N, d)
    _, Rf = rp.gen.forest_hyperplanes(99, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
    Q = X[torch.randint(0, N, (nq,), device=dev, generator=g)].clone()
    qs = rp.Dataset.dense_device(ctx, Q.data_ptr(), nq, d, rp.RPT_BF16, keep=Q)
    torch.cuda.synchronize()
    for name, env in (("bf16x3", None), ("f32-mfma", "1")):
        ctx.set_option("proj_bf16_f32", 1 if env else 0)
        for it in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            f = rp._build(ctx, ds, Rf, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_MFMA)
            ctx.sync()
            t1 = time.perf_counter()
            if it < 2:
                f.close()
        print("%-9s build (T %d, L %d): %.2f ms = %.1f M vectors/s" % (
            name, T, cfg.fpMaxTreeDepth, (t1 - t0) * 1e3, N / (t1 - t0) / 1e6), flush=True)
        if env is None:
            perm = f.perm
            assert np.array_equal(np.bincount(perm[0], minlength=N), np.ones(N, dtype=np.int64))
            ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
            dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
            cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
            for it in range(2):
                ctx.sync()
                t0 = time.perf_counter()
                _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, ids.data_ptr(),
                                          dist.data_ptr(), cnt.data_ptr()))
                ctx.sync()
                t1 = time.perf_counter()
            print("          knn k=%d: %d queries in %.2f ms = %.2f M queries/s; self hit first: %.3f" % (
                k, nq, (t1 - t0) * 1e3, nq / (t1 - t0) / 1e6, float((dist[:, 0] < 1e-3).double().mean())))
        f.close()
