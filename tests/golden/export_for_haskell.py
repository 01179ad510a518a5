#!/usr/bin/env python3
"""Writes the inputs of a golden fixture as plain text for haskell/app/GenGolden.hs (no RNG is
involved on the Haskell side: data, dense-ified hyperplanes and queries are all explicit).

    python tests/golden/export_for_haskell.py        -> tests/golden/forest_dense_1000x16.in.txt
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    z = np.load(os.path.join(HERE, "forest_dense_1000x16.npz"))
    n, d, T, L, ml = (int(z[k]) for k in ("n", "d", "T", "L", "min_leaf"))
    Q = z["Q"]
    k = z["knn_ids"].shape[1]
    with open(os.path.join(HERE, "forest_dense_1000x16.in.txt"), "w") as f:
        f.write("%d %d %d %d %d %d %d\n" % (n, d, T, L, ml, len(Q), k))
        for a in (z["X"], z["R"], Q):
            f.write(" ".join(repr(float(v)) for v in a.ravel()) + "\n")   # repr: exact round trip
    print("wrote forest_dense_1000x16.in.txt")


if __name__ == "__main__":
    main()
