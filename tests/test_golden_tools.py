"""The export / diff scripts around haskell/app/GenGolden.hs: the text a correct reference run
would print (rendered from the fixture here) must pass diff_haskell.py, a corrupted one must not;
the exported input must hold exactly the fixture's numbers."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(GOLD, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def render(z):
    T = int(z["T"])
    lines = []
    fmt = lambda a: " ".join("nan" if np.isnan(v) else repr(float(v)) for v in a)
    for t in range(T):
        lines.append("perm %d " % t + " ".join(str(int(i)) for i in z["perm"][t]))
        for tag in ("thr", "mglo", "mghi"):
            lines.append("%s %d " % (tag, t) + fmt(z[tag][t]))
    for q in range(len(z["Q"])):
        for t in range(T):
            a, b = z["cand_off"][q * T + t], z["cand_off"][q * T + t + 1]
            lines.append("cand %d %d " % (q, t) + " ".join(str(int(i)) for i in z["cand_ids"][a:b]))
        ids = [int(i) for i in z["knn_ids"][q] if i >= 0]
        lines.append("knn_ids %d " % q + " ".join(map(str, ids)))
        lines.append("knn_dist %d " % q + fmt(z["knn_dist"][q][:len(ids)]))
        if q < len(z["recall_with"]):
            lines.append("recall %d %r" % (q, float(z["recall_with"][q])))
    return lines


def test_diff_accepts_the_fixture_and_rejects_a_change(tmp_path):
    z = np.load(os.path.join(GOLD, "forest_dense_1000x16.npz"))
    lines = render(z)
    good = tmp_path / "good.txt"
    good.write_text("\n".join(lines) + "\n")
    diff = _load("diff_haskell")
    assert diff.main(str(good)) == 0
    w = lines[0].split()
    w[2], w[3] = w[3], w[2]                       # two points of the first leaf swapped
    bad = tmp_path / "bad.txt"
    bad.write_text("\n".join([" ".join(w)] + lines[1:]) + "\n")
    assert diff.main(str(bad)) == 1


def test_export_round_trips(tmp_path, monkeypatch):
    exp = _load("export_for_haskell")
    monkeypatch.setattr(exp, "HERE", str(tmp_path))
    z = np.load(os.path.join(GOLD, "forest_dense_1000x16.npz"))
    np.savez(os.path.join(str(tmp_path), "forest_dense_1000x16.npz"), **{k: z[k] for k in z.files})
    exp.main()
    ws = open(os.path.join(str(tmp_path), "forest_dense_1000x16.in.txt")).read().split()
    n, d, T, L, ml, nq, k = (int(v) for v in ws[:7])
    assert (n, d, T, L, ml) == tuple(int(z[x]) for x in ("n", "d", "T", "L", "min_leaf"))
    vals = np.array([float(v) for v in ws[7:]])
    want = np.concatenate([z["X"].ravel(), z["R"].ravel(), z["Q"].ravel()])
    assert np.array_equal(vals, want)
