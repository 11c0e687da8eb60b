#!/usr/bin/env python
"""Dispatch-ordered kernels between two name patterns of the LAST step in a rocprofv3 rocpd DB (start, duration, gap to the predecessor):
   python tools/prof_tail.py <db> <from-pattern> <to-pattern>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, grid_x, grid_y, start, end from kernels order by start"))
a = max(i for i, r in enumerate(rows) if sys.argv[2] in r[0])
b = min([i for i, r in enumerate(rows) if i > a and sys.argv[3] in r[0]] or [len(rows) - 1])
t0 = rows[a][3]
prev_end = rows[a][3]
for name, gx, gy, s, e in rows[a:b + 1]:
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    print("%9.1f us  dur %8.1f  gap %6.1f  grid %8d %5d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, gx, gy, short))
    prev_end = e
print("span %.1f us" % ((rows[b][4] - t0) / 1e3))
