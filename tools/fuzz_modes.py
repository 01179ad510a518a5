"""Randomised sweep of forests built in the TOLERANCE mode (RPT_PROJ_MFMA) on f64, f32 and bf16 rows: the split
must be exact with respect to the device's OWN stored projections whatever the projection kernel — every tree a
permutation, and on every split node of every level (Internal.hs:496-501): max(left) <= thr == min(right),
margins = the neighbours of the cut.  Self-consistency, no oracle: this is the part of a tolerance-mode build
that is still exact.
usage: python tools/fuzz_modes.py [seconds] [seed]"""
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "rp-tree_amd", "python"))
import numpy as np

import rptree_amd as rp
from rptree_amd import gen

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = rp.Context(0)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    kind = str(rng.choice(["f64", "f32", "bf16"]))
    n = int(rng.choice([rng.integers(2, 3000), rng.integers(3000, 60000), rng.integers(131072, 400000)],
                       p=[0.3, 0.4, 0.3]))
    d = int(rng.choice([8, 16, 64, 72, 128, 256] if kind == "bf16" else [3, 16, 100, 128, 132]))
    T = int(rng.integers(1, 5))
    min_leaf = int(rng.choice([1, 5, 20, 100, 300]))
    cfg = rp.rpTreeCfg(max(min_leaf, 1), max(n, 2), max(d, 2))
    L = int(np.clip(rng.choice([cfg.fpMaxTreeDepth, cfg.fpMaxTreeDepth - 2, 3]), 1, 15))
    if n * L * T > 12_000_000:
        continue
    ties = bool(rng.random() < 0.3)
    X = rng.standard_normal((n, d))
    if ties:
        X = np.round(X * 2) / 2
    if kind == "f64":
        ds = rp.Dataset.dense(ctx, X)
    elif kind == "f32":
        ds = rp.Dataset.dense(ctx, X.astype(np.float32), dtype=rp.RPT_F32)
    else:
        ds = rp.Dataset.dense(ctx, rp.to_bf16(X.astype(np.float32)), dtype=rp.RPT_BF16)
    cseed = int(rng.integers(1, 1 << 30))
    _, R = gen.forest_hyperplanes(cseed, T, L, float(rng.choice([cfg.fpProjNzDensity, 1.0])), d)
    desc = "%s n=%d d=%d T=%d minLeaf=%d L=%d ties=%d seed=%d" % (kind, n, d, T, min_leaf, L, ties, cseed)
    f = rp._build(ctx, ds, R, L, min_leaf, rp.RPT_PROJ_MFMA)
    topo = f.topology()
    P = f.proj() if any(not r[4] for r in topo) else None     # (a root that is a Tip: nothing was projected)
    perm, thr, mglo, mghi = f.perm, f.thr, f.mglo, f.mghi
    for t in range(T):
        assert np.array_equal(np.sort(perm[t]), np.arange(n)), desc
    for (level, heap, off, m, leaf) in topo:
        if leaf:
            continue
        level, heap, off, m = int(level), int(heap), int(off), int(m)
        nh = m // 2
        for t in range(T):
            p = P[t, level]
            left = p[perm[t, off:off + nh]]
            right = p[perm[t, off + nh:off + m]]
            lmax = left.max() if nh > 0 else None
            rs = np.partition(right, min(1, len(right) - 1))
            assert thr[t, heap] == rs[0], (desc, level, heap)
            if nh > 0:
                assert lmax <= thr[t, heap], (desc, level, heap)
                assert mglo[t, heap] == lmax, (desc, level, heap)
            if len(right) > 1:
                assert mghi[t, heap] == rs[1], (desc, level, heap)
    f.close()
    ds.close()
    cases += 1
    if cases % 10 == 0:
        print("%d cases ok, last: %s" % (cases, desc), flush=True)
print("mode fuzz ok: %d cases" % cases)
