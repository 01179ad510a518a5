"""Per-kernel timeline / aggregate of one forest build from a rocprofv3 rocpd database.
usage: python tools/rocpd_timeline.py results.db [build_index] [--agg]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
agg = "--agg" in sys.argv
rows = db.execute("select name,start,end from kernels order by start").fetchall()


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("rpt::", "")
    i = n.find("(")
    return re.sub(r"\s+", "", n[:i] if i > 0 else n)[:44]


idx = [i for i, r in enumerate(rows) if "stream_minmax0" in r[0]]
i0 = idx[which]
j = i0
while j > 0 and "wsub_kernel" not in rows[j][0]:
    j -= 1
while j + 1 < i0 and "leaf_sort" in rows[j + 1][0]:
    j += 1
s = j + 1
e = i0
while "wsub_kernel" not in rows[e][0]:
    e += 1
while e + 1 < len(rows) and "leaf_sort" in rows[e + 1][0]:
    e += 1
t0 = rows[s][1]
prev = None
busy = 0
tot = {}
for r in rows[s:e + 1]:
    gap = (r[1] - prev) / 1e3 if prev else 0
    busy += r[2] - r[1]
    k = short(r[0])
    a = tot.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (r[2] - r[1]) / 1e3
    if not agg:
        print("%8.1f us  +gap %6.1f  dur %7.1f  %s" % ((r[1] - t0) / 1e3, gap, (r[2] - r[1]) / 1e3, k))
    prev = r[2]
for k, a in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%-46s n=%3d total %8.1f us avg %7.1f" % (k, a[0], a[1], a[1] / a[0]))
print("span %.1f us busy %.1f us" % ((rows[e][2] - t0) / 1e3, busy / 1e3))
