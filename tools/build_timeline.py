"""Kernel timeline of ONE forest build from a rocprofv3 rocpd database: every launch between two
consecutive build_init_kernel launches, with the idle gap in front of it.
usage: python tools/build_timeline.py results.db [which = -2] [--agg]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = -2
for a in sys.argv[2:]:
    if re.fullmatch(r"-?\d+", a):
        which = int(a)
agg = "--agg" in sys.argv
rows = db.execute("select name,start,end from kernels order by start").fetchall()


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("rpt::", "")
    i = n.find("(")
    return re.sub(r"\s+", "", n[:i] if i > 0 else n)[:48]


idx = [i for i, r in enumerate(rows) if "build_init_kernel" in r[0]]
s, e = idx[which], idx[which + 1] if which + 1 < 0 or which + 1 < len(idx) else len(rows)
# the build ends with its last split kernel: drop whatever the caller ran afterwards (kNN, copies)
last = s
for i in range(s, e):
    if any(k in rows[i][0] for k in ("wsort", "wsub", "subtree", "small_sort", "stream_", "csub", "proj_")):
        last = i
t0 = rows[s][1]
prev = None
busy = gaps = 0
tot = {}
for r in rows[s:last + 1]:
    gap = (r[1] - prev) / 1e3 if prev else 0
    gaps += max(gap, 0)
    busy += r[2] - r[1]
    k = short(r[0])
    a = tot.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (r[2] - r[1]) / 1e3
    if not agg:
        print("%8.1f us  +gap %6.1f  dur %7.1f  %s" % ((r[1] - t0) / 1e3, gap, (r[2] - r[1]) / 1e3, k))
    prev = r[2] if prev is None or r[2] > prev else prev
for k, a in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%-50s n=%3d total %8.1f us avg %7.1f" % (k, a[0], a[1], a[1] / a[0]))
print("span %.1f us busy %.1f us gaps %.1f us launches %d" % ((rows[last][2] - t0) / 1e3, busy / 1e3, gaps, last + 1 - s))
