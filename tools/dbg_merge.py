"""debug: shard merges of large k against a numpy reference, reporting the first mismatch"""
import sys
sys.path.insert(0, "rp-tree_amd/python"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import rptree_amd as rp
from rptree_amd import _lib
from test_gpu_abi import merge_reference
ctx = rp.default_context()
L = _lib.lib()
for (G, k, ties) in [(2, 1024, 0), (2, 1024, 1), (2, 600, 1), (4, 1024, 1), (3, 1024, 0), (3, 1024, 1), (8, 1024, 1), (8, 512, 1), (2, 2, 1)]:
    rng = np.random.default_rng(G * k)
    nq = 5
    gi = rng.integers(0, 3000, size=(G, nq, k)).astype(np.int32)
    gd = rng.random((G, nq, k)) * 50
    if ties:
        gd = np.round(gd, 1)
    gd = np.sort(gd, axis=2)
    gc = rng.integers(k // 2, k + 1, size=(G, nq)).astype(np.int32)
    ti, td, tc = (torch.from_numpy(a).cuda() for a in (gi, gd, gc))
    oi = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.rpt_knn_merge_dev(ctx._h, ti.data_ptr(), td.data_ptr(), tc.data_ptr(), G, nq, k, 0,
                                   oi.data_ptr(), od.data_ptr(), oc.data_ptr()))
    ctx.sync()
    wi, wd, wc = merge_reference(gi, gd, gc, k, False)
    a, b = oi.cpu().numpy(), od.cpu().numpy()
    bad = [(q, int(np.argmax((a[q] != wi[q]) | (b[q] != wd[q])))) for q in range(nq)
           if not (np.array_equal(a[q], wi[q]) and np.array_equal(b[q], wd[q]))]
    print(G, k, "ties" if ties else "cont", "counts ok" if np.array_equal(oc.cpu().numpy(), wc) else "COUNTS", "first mismatch (q, pos):", bad[:3])
    if bad:
        q, p = bad[0]
        print("   got ", a[q, p:p + 4], b[q, p:p + 4], "\n   want", wi[q, p:p + 4], wd[q, p:p + 4])
