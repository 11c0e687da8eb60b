#!/usr/bin/env python
"""Per-kernel averages of every PMC counter in a rocprofv3 rocpd DB (kernels matching a substring)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
q = ("select kernel_name, grid_size, counter_name, count(*), avg(value) from counters_collection "
     "where kernel_name like ? group by kernel_name, grid_size, counter_name order by grid_size, counter_name")
for r in db.execute(q, ("%" + pat + "%",)):
    print("%-40s grid %8d %-28s n %3d avg %.4g" % (r[0].replace("void (anonymous namespace)::", "")[:40], r[1], r[2], r[3], r[4]))
