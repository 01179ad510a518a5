#!/usr/bin/env python3
"""Compares the output of haskell/app/GenGolden.hs (the real reference) with the committed
fixture (= the oracle's and the device's results), bit for bit.

    python tests/golden/diff_haskell.py forest_dense_1000x16.hs.txt
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main(path):
    z = np.load(os.path.join(HERE, "forest_dense_1000x16.npz"))
    T = int(z["T"])
    bad = 0
    for ln in open(path):
        w = ln.split()
        if not w:
            continue
        tag, vals = w[0], w[1:]
        if tag in ("perm", "thr", "mglo", "mghi"):
            t = int(vals[0])
            got = np.array([float(v) for v in vals[1:]])
            want = z[tag][t].astype(np.float64)
            ok = np.array_equal(got, want, equal_nan=True)
        elif tag == "cand":
            q, t = int(vals[0]), int(vals[1])
            a, b = z["cand_off"][q * T + t], z["cand_off"][q * T + t + 1]
            ok = [int(v) for v in vals[2:]] == z["cand_ids"][a:b].tolist()
        elif tag == "knn_ids":
            q = int(vals[0])
            want = [i for i in z["knn_ids"][q].tolist() if i >= 0]
            ok = [int(v) for v in vals[1:]] == want
        elif tag == "knn_dist":
            q = int(vals[0])
            want = z["knn_dist"][q][np.isfinite(z["knn_dist"][q])]
            ok = np.array_equal(np.array([float(v) for v in vals[1:]]), want)
        elif tag == "recall":
            q = int(vals[0])
            ok = q >= len(z["recall_with"]) or float(vals[1]) == float(z["recall_with"][q])
        else:
            continue
        if not ok:
            bad += 1
            print("MISMATCH:", " ".join(w[:3]))
    print("reference and fixture agree" if bad == 0 else "%d records differ" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
