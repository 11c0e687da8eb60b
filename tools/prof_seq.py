#!/usr/bin/env python
"""Dispatch-ordered list of the kernels whose name contains a pattern (rocprofv3 rocpd sqlite DB):
   python tools/prof_seq.py <db> <pattern> [max_rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 200
t0 = None
for name, gx, gy, s, e in db.execute("select name, grid_x, grid_y, start, end from kernels where name like ? order by start limit ?", ("%" + pat + "%", lim)):
    t0 = s if t0 is None else t0
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
    print("%10.1f us  +%8.1f us  grid %8d %4d  %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, gx, gy, short))
