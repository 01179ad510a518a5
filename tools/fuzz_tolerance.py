"""Randomised sweep of the TOLERANCE-mode projections (RPT_PROJ_MFMA): bf16 rows (two and three hyperplane
terms, both workgroup shapes), f32 rows, dense-ified SVector rows — random point counts, row lengths and
hyperplane counts; every value must lie within 1e-5 |x||r| of the f64 contraction (bf16 rows with two terms:
8e-6: 2^-17 by construction plus the accumulation), and the variants that compute the same sums must agree bit for bit.
usage: python tools/fuzz_tolerance.py [seconds] [seed]"""
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "rp-tree_amd", "python"))
import numpy as np
import torch

import rptree_amd as rp

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = rp.Context(0)
dev = torch.device("cuda", 0)
t_end = time.time() + budget
cases = 0
worst = {"bf16x2": 0.0, "bf16x3": 0.0, "f32": 0.0, "csr": 0.0}


def sparse_R(C, d, pnz):
    return rng.standard_normal((C, d)) * (rng.random((C, d)) < pnz)


def rel_err(P, X64, R):
    want = R @ X64.T
    scale = np.linalg.norm(R, axis=1)[:, None] * np.linalg.norm(X64, axis=1)[None, :]
    scale[scale == 0] = 1.0
    return float((np.abs(P.astype(np.float64) - want) / scale).max())


while time.time() < t_end:
    kind = str(rng.choice(["bf16", "bf16", "f32", "csr"]))
    C = int(rng.choice([1, 5, 33, 64, 65, 128, 129, 200, 300]))
    pnz = float(rng.choice([0.2, 0.35, 1.0]))
    if kind == "bf16":
        n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 20000), rng.integers(20000, 120000)]))
        d = int(rng.choice([8, 64, 72, 128, 136, 200, 256, 264, 512, 768, 1000]))
        R = sparse_R(C, d, pnz)
        xb = torch.from_numpy((rng.standard_normal((n, d)) * float(rng.choice([0.01, 1.0, 300.0]))).astype(np.float32)
                              ).to(torch.bfloat16).to(dev)
        torch.cuda.synchronize()
        ds = rp.Dataset.dense_device(ctx, xb.data_ptr(), n, d, rp.RPT_BF16, keep=xb)
        X64 = xb.float().cpu().numpy().astype(np.float64)
        P = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
        e2 = rel_err(P, X64, R)
        assert e2 <= (8e-6 if d >= 64 else 2e-6), ("bf16 two terms (three below 64 elements)", n, d, C, e2)
        old = ctx.set_option("proj_bf16_terms", 8)
        P8 = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
        ctx.set_option("proj_bf16_terms", 3)
        P3 = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
        ctx.set_option("proj_bf16_terms", old)
        assert np.array_equal(P, P8), ("bf16 workgroup shapes differ", n, d, C)
        e3 = rel_err(P3, X64, R)
        assert e3 <= 2e-6, ("bf16 three terms", n, d, C, e3)
        worst["bf16x2"] = max(worst["bf16x2"], e2)
        worst["bf16x3"] = max(worst["bf16x3"], e3)
        desc = "bf16 n=%d d=%d C=%d: %.1e / %.1e" % (n, d, C, e2, e3)
    elif kind == "f32":
        n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 60000)]))
        d = int(rng.choice([4, 16, 100, 128, 132, 256, 300]))
        R = sparse_R(C, d, pnz)
        X = rng.standard_normal((n, d)).astype(np.float32)
        ds = rp.Dataset.dense(ctx, X, dtype=rp.RPT_F32)
        P = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
        e = rel_err(P, X.astype(np.float64), R)
        assert e <= 1e-5, ("f32", n, d, C, e)
        worst["f32"] = max(worst["f32"], e)
        desc = "f32 n=%d d=%d C=%d: %.1e" % (n, d, C, e)
    else:
        n = int(rng.integers(65536, 90000))          # the dense-ified path starts at 65 536 rows
        d = int(rng.choice([64, 200, 784, 1000]))
        C = min(C, 140)
        R = sparse_R(C, d, pnz)
        nnz_row = max(2, int(d * float(rng.choice([0.05, 0.2]))))
        cols = np.sort(np.stack([rng.choice(d, nnz_row, replace=False) for _ in range(500)]), axis=1)
        cols = cols[rng.integers(0, 500, n)]
        rowptr = np.arange(n + 1, dtype=np.int64) * nnz_row
        col = cols.reshape(-1).astype(np.int32)
        val = (1.0 - rng.random(n * nnz_row))
        ds = rp.Dataset.csr(ctx, rowptr, col, val, d)
        X64 = np.zeros((n, d))
        X64[np.repeat(np.arange(n), nnz_row), col] = val
        P = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
        e = rel_err(P, X64, R)
        assert e <= 1e-5, ("csr dense", n, d, C, e)
        worst["csr"] = max(worst["csr"], e)
        desc = "csr n=%d d=%d C=%d nnz/row=%d: %.1e" % (n, d, C, nnz_row, e)
    ds.close()
    cases += 1
    if cases % 10 == 0:
        print("%d cases ok, last: %s" % (cases, desc), flush=True)
print("tolerance fuzz ok: %d cases; worst |err| / (|x||r|): %s" % (cases, {k: "%.2e" % v for k, v in worst.items()}))
