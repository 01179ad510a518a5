"""per-level kernel durations of the LAST complete build in a rocprofv3 rocpd database"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end-start from kernels where name like '%stream_%' or name like '%wsub%' or name like '%proj_mfma_wide%' order by start").fetchall()
seq = [(r[0].split('<')[0].replace('void rpt::(anonymous namespace)::', ''), r[2] / 1e3) for r in rows]
idx = [i for i, (n, _) in enumerate(seq) if 'minmax0' in n]
tot = {}
for n, d in seq[idx[-2]:idx[-1]]:
    if 'init' in n: continue
    tot.setdefault(n, []).append(d)
for n, v in tot.items():
    print("  %-18s n=%2d sum=%8.1f us  [%s]" % (n, len(v), sum(v), " ".join("%.0f" % x for x in v)))
