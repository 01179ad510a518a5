"""avg duration of the projection kernels in a rocprofv3 rocpd database (A/B experiments)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select name, grid_x*grid_y*grid_z, count(*), avg(end-start) from kernels where name like '%proj_%' group by name, grid_x*grid_y*grid_z order by 4 desc"):
    if r[1] >= 100000:
        print("  %-70s %4d x %8.1f us" % (r[0].replace("(anonymous namespace)::", "")[:70], r[2], r[3] / 1e3))
