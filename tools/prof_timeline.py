#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 rocpd sqlite DB (kernel-trace): every kernel between two consecutive xdec_fwd_kernel
launches, with start offset, duration, the stream it ran on and how many other kernels were running when it started.
   python tools/prof_timeline.py <db> [which_step]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
rows = list(db.execute("select name, grid_x, start, end, stream_id from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "xdec_fwd_kernel" in r[0]]
a, b = marks[which], marks[which + 1]
sel = rows[a:b]
t0 = sel[0][2]
for i, (name, gx, s, e, sid) in enumerate(sel):
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    running = sum(1 for (_, _, s2, e2, _) in sel[:i] if e2 > s)
    print("%9.1f us  +%7.1f us  stream %3s  beside %d  %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, sid, running, short))
print("step: %.1f us" % ((rows[b][2] - t0) / 1000.0))
