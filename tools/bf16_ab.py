"""proj_bf16x3 shapes A/B on a C5-shaped pass (bf16 rows against 128 hyperplanes): option proj_bf16_shape;
shapes with the same number of hyperplane terms must be bit-identical (same k and term order per element).
usage: python tools/bf16_ab.py [npoints] [dim] [columns] [shapes, comma separated]"""
import ctypes as C
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "rp-tree_amd", "python"))
import numpy as np
import torch

import rptree_amd as rp
from rptree_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
Cn = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
X = torch.empty(N, d, device=dev, dtype=torch.bfloat16)
step = 1_000_000
for i in range(0, N, step):
    x = torch.randn(min(step, N - i), d, device=dev, dtype=torch.float32, generator=g)
    X[i:i + step] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
rng = np.random.default_rng(5)
R = np.ascontiguousarray(rng.standard_normal((Cn, d)) * (rng.random((Cn, d)) < 0.3466))
P = torch.empty((Cn, N), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ctx = rp.Context(0)
L_ = _lib.lib()
ds = rp.Dataset.dense_device(ctx, X.data_ptr(), N, d, rp.RPT_BF16, keep=X)
keep = {}
for old in [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "3,8,0,8,0").split(",")]:
    ctx.set_option("proj_bf16_terms", old)
    ts = []
    for it in range(5):
        ctx.sync()
        t0 = time.perf_counter()
        _lib.check(L_.rpt_project_dev(ctx._h, ds._h, R.ctypes.data_as(C.c_void_p), Cn, rp.RPT_PROJ_MFMA, P.data_ptr()))
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    byt = N * d * 2 + N * Cn * 4
    print("terms %d: best %.3f ms, median %.3f ms  (X + P once = %.2f GB -> %.2f TB/s)" % (
        old, min(ts), sorted(ts)[2], byt / 1e9, byt / min(ts) / 1e9), flush=True)
    torch.cuda.synchronize()
    if old not in keep:
        keep[old] = P[:, ::7].clone()
        torch.cuda.synchronize()
    xs = X[:4096].double()
    ref = (xs @ torch.from_numpy(R).to(dev).T).T
    e = (P[:, :4096].double() - ref).abs() / (xs.norm(dim=1)[None, :] * torch.from_numpy(np.linalg.norm(R, axis=1)).to(dev)[:, None])
    print("         max |err| / (|x||r|) over 4096 points = %.2e" % e.max().item(), flush=True)
for a in keep:
    for b in keep:
        if a < b:
            print("shapes %d and %d bit-identical: %s" % (a, b, bool(torch.equal(keep[a], keep[b]))))
