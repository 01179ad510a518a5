import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/rp-tree_amd/python')
import numpy as np
from oracle import oracle as o
import rptree_amd as rp
rowptr, col, val = o.data_normal_sparse2(1234, 6000, 12, 0.25)
R, _ = o.forest_hyperplanes(7, 3, 6, 0.3, 12)
fo = o.forest_build_csr(rowptr, col, val, 12, R, 10, want_proj=True)
f = rp.forestBatch(7, 6, 10, 3, 0.3, 12, (rowptr, col, val, 12))
print("stats", f.stats())
topo = f.topology()
for t in range(3):
    d = np.nonzero(f.perm[t] != fo.perm[t])[0]
    print("tree", t, "ndiff", len(d), d[:10])
    print(" valid perm", np.array_equal(np.sort(f.perm[t]), np.arange(6000)))
    for (lv, heap, off, n, leaf) in topo:
        if leaf:
            a, b = f.perm[t, off:off+n], fo.perm[t, off:off+n]
            if not np.array_equal(a, b):
                print("  leaf lv", lv, "off", off, "n", n, "same set", set(a) == set(b))
                print("   gpu", a[:12], "\n   ora", b[:12])
                P = f.proj()
                print("   key(gpu order)", P[t, lv-1][a[:12]])
                break
    for name in ("thr", "mglo", "mghi"):
        x, y = getattr(f, name)[t], getattr(fo, name)[t]
        bad = np.nonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))[0]
        print(" ", name, "mismatch at heaps", bad[:10])
