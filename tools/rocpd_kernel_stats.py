"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd database), grouped by kernel name
AND grid size, so that the full-size launches of a kernel (the ones bench.py's roofline is about)
are not averaged together with the small query-side launches of the same kernel.
usage: python tools/rocpd_kernel_stats.py results.db out.csv "command line that was profiled" """
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, grid_x * grid_y * grid_z, count(*), sum(end-start), avg(end-start), min(end-start),"
    " max(end-start) from kernels group by name, grid_x * grid_y * grid_z order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    i = n.find("(")
    return re.sub(r"\s+", "", n[:i] if i > 0 else n)[:90]


out = ["# rocprofv3 --kernel-trace -- %s" % sys.argv[3],
       "# rocpd database aggregated per (kernel name, grid size in work-items)",
       "name,grid_work_items,calls,total_ns,avg_ns,min_ns,max_ns,pct"]
for r in rows:
    out.append('"%s",%d,%d,%d,%.0f,%d,%d,%.2f' % (short(r[0]), r[1], r[2], r[3], r[4], r[5], r[6],
                                                 100.0 * r[3] / tot))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:30]))
