"""Per-kernel means of rocprofv3 --pmc counters from a rocpd database, full-size launches only
(the largest grid of each kernel).  usage: python tools/pmc_mean.py results.db out.csv COUNTER..."""
import re
import sqlite3
import sys


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("rpt::", "")
    i = n.find("(")
    return re.sub(r"\s+", "", n[:i] if i > 0 else n)


db = sqlite3.connect(sys.argv[1])
counters = sys.argv[3:]
acc = {}
for name, cname, val, grid in db.execute(
        "select kernel_name, counter_name, value, grid_size from counters_collection"):
    if cname not in counters:
        continue
    k = short(name)
    if not (k.startswith("proj_") or k.startswith("stream_") or k.startswith(("wsub", "wsort", "wpack", "knn_", "subtree", "hist_kernel", "scatter_kernel", "sample_kernel", "mid_kernel"))):
        continue
    acc.setdefault(k, {}).setdefault(grid, {}).setdefault(cname, []).append(val)
rows = ["kernel,grid,launches," + ",".join(counters)]
for k in sorted(acc):
    g = max(acc[k])
    d = acc[k][g]
    n = max(len(v) for v in d.values())
    rows.append('"%s",%d,%d,' % (k, g, n) + ",".join(
        "%.4g" % (sum(d[c]) / len(d[c])) if c in d else "" for c in counters))
open(sys.argv[2], "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
