#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (the reference is Haskell and cannot be
built here: no GHC — SURVEY.md §8c; the oracle is pinned by the reference's own 4 KATs,
tests/test_oracle_kat.py).  Fixtures are data only: inputs and expected outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as o  # noqa: E402


def main():
    o.build()
    # (i) the reference's known-answer tests, test/Data/RPTreeSpec.hs:21-45
    np.savez(os.path.join(HERE, "kat_vectors.npz"),
             vs0_idx=[1, 4], vs0_val=[3.4, 2.1], vs1_idx=[0, 3], vs1_val=[6.7, 5.5],
             v1=[1.0, 2.0, 3.0, 4.0, 5.0], sum_sd=[1, 5.4, 3, 4, 7.1],
             diff_sd=[-1, 1.4, -3, -4, -2.9], inner_ss=0.0, inner_sd=17.3)

    # (ii) partitionAtMedian on small n incl. ties and +-0.0 (Internal.hs:486-505)
    cases = [[5.0], [2.0, 1.0], [3.0, 1.0, 2.0], [0.0, -0.0, 0.0, -1.0], [1.0] * 7,
             [4.0, 4.0, 1.0, 4.0, 0.5, 4.0, 9.0, 4.0], [2.5, -1.0, 2.5, 2.5, 0.0, -1.0, 7.0]]
    out = {}
    for i, p in enumerate(cases):
        nh, order, thr, lo, hi = o.partition_at_median(p)
        out["p%d" % i] = np.array(p)
        out["order%d" % i] = order
        out["res%d" % i] = np.array([nh, thr, lo, hi])
    np.savez(os.path.join(HERE, "partition_small.npz"), n=len(cases), **out)

    # (iii) full forestBatch, dense: N=1000, d=16, T=3, minLeaf=20 (rpTreeCfg depth / pnz)
    n, d, T, ml = 1000, 16, 3, 20
    X = o.data_normal_dense2(1234, n, d)
    L, _, pnz = o.tree_cfg(ml, n, d)
    R, _ = o.forest_hyperplanes(1235137, T, L, pnz, d)
    f = o.forest_build_dense(X, R, ml)
    Q = o.data_normal_dense2(4321, 32, d)
    # (v) candidates + knn for 32 queries, (vi) recallWith
    cand = [o.candidates_dense(f, Q[i], t) for i in range(32) for t in range(T)]
    cand_off = np.concatenate([[0], np.cumsum([len(c) for c in cand])])
    knn_ids = np.full((32, 10), -1, dtype=np.int32)
    knn_dist = np.full((32, 10), np.inf)
    knn_ids_d = np.full((32, 10), -1, dtype=np.int32)
    for i in range(32):
        a, b = o.knn_dense(f, X, Q[i], 10)
        knn_ids[i, :len(a)], knn_dist[i, :len(a)] = a, b
        a, _ = o.knn_dense(f, X, Q[i], 10, dedup=True)
        knn_ids_d[i, :len(a)] = a
    recall = np.array([o.recall_with_dense(f, X, Q[i], 10) for i in range(8)])
    np.savez_compressed(os.path.join(HERE, "forest_dense_1000x16.npz"), seed_data=1234,
                        seed_forest=1235137, seed_query=4321, n=n, d=d, T=T, min_leaf=ml, L=L,
                        pnz=pnz, X=X, R=R, Q=Q, perm=f.perm, thr=f.thr, mglo=f.mglo, mghi=f.mghi,
                        cand_off=cand_off, cand_ids=np.concatenate(cand), knn_ids=knn_ids,
                        knn_dist=knn_dist, knn_ids_dedup=knn_ids_d, recall_with=recall)

    # (iv) sparse data x sparse hyperplanes with many exact-zero projections (ties at the cut)
    n, d, T, ml, L = 600, 12, 3, 10, 5
    rowptr, col, val = o.data_normal_sparse2(1234, n, d, 0.25)
    R, _ = o.forest_hyperplanes(7, T, L, 0.3, d)
    f = o.forest_build_csr(rowptr, col, val, d, R, ml, want_proj=True)
    np.savez_compressed(os.path.join(HERE, "forest_sparse_600x12.npz"), n=n, d=d, T=T, min_leaf=ml,
                        L=L, rowptr=rowptr, col=col, val=val, R=R, perm=f.perm, thr=f.thr,
                        mglo=f.mglo, mghi=f.mghi, zero_proj_root=int((f.proj[0, 0] == 0).sum()))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
