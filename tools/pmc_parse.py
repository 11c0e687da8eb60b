#!/usr/bin/env python
"""Average a PMC counter per dispatch of kernels matching a substring (rocprofv3 rocpd sqlite)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cand = [n for n in names if "pmc" in n.lower() or "counter" in n.lower()]
print("tables:", cand)
for n in cand:
    try:
        cols = [d[0] for d in db.execute("select * from %s limit 1" % n).description]
        print(n, cols)
    except Exception as e:
        print(n, "ERR", e)
