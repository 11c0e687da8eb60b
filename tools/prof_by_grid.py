#!/usr/bin/env python
"""Per-(kernel, grid) average duration from a rocprofv3 rocpd sqlite DB."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in db.execute("select * from kernels limit 1").description]
print(cols)
q = """select name, grid_x, grid_y, grid_z, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 from kernels
       group by name, grid_x, grid_y, grid_z order by sum(end-start) desc limit 40"""
for r in db.execute(q):
    print("%-50s grid %6d %4d %4d  n %5d  avg %9.2f us  total %10.1f us" % (r[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:50], r[1], r[2], r[3], r[4], r[5], r[6]))
