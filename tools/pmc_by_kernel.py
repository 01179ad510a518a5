"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd databases) per kernel.
usage: python tools/pmc_by_kernel.py fetch.db write.db out.csv out.json
FETCH_SIZE is doubled (gfx950 reports 1/2 of wide coalesced reads: MI355X_MICROARCH.md, HBM /
rocprofv3 section); WRITE_SIZE is taken as is; unit KB = 1024 B.  Values are the per-launch MEAN
over the full-size launches of a kernel (largest grid; query-side launches are tiny)."""
import json
import re
import sqlite3
import sys


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("rpt::", "")
    i = n.find("(")
    return re.sub(r"\s+", "", n[:i] if i > 0 else n)


def load(path, counter):
    """-> {kernel: [full-size launches, max KB, mean KB over the full-size launches]}; full-size =
    the launches with the kernel's largest grid (query-side launches of the same kernel are tiny)"""
    db = sqlite3.connect(path)
    rows = {}
    for name, val, grid in db.execute(
            "select kernel_name, value, grid_size from counters_collection where counter_name = ?",
            (counter,)):
        k = short(name)
        if not (k.startswith("proj_") or k.startswith("stream_") or k.startswith("wsub") or
                k.startswith("leaf_sort") or k.startswith("knn_")):
            continue
        rows.setdefault(k, []).append((grid, val))
    out = {}
    for k, lst in rows.items():
        g = max(x[0] for x in lst)
        vals = [x[1] for x in lst if x[0] == g]
        out[k] = [len(vals), max(vals), sum(vals) / len(vals)]
    return out


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
rows = ["kernel,full_size_launches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_read_bytes_corrected_x2,hbm_write_bytes"]
js = {}
for k in sorted(set(f) | set(w)):
    fk = f.get(k, [0, 0.0, 0.0])
    wk = w.get(k, [0, 0.0, 0.0])
    rd, wr = fk[2] * 1024 * 2, wk[2] * 1024
    rows.append("%s,%d,%.1f,%.1f,%.0f,%.0f" % (k, fk[0], fk[2], wk[2], rd, wr))
    js[k] = {"read": rd, "write": wr, "hbm_bytes_per_launch": rd + wr, "launches": fk[0]}
open(sys.argv[3], "w").write("\n".join(rows) + "\n")
json.dump(js, open(sys.argv[4], "w"), indent=1)
print("\n".join(rows))
