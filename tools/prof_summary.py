#!/usr/bin/env python
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.
usage: tools/prof_summary.py <results.db> <out.csv> [note]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(sys.argv[2], "w", newline="") as f:
        if len(sys.argv) > 3:
            f.write("# %s\n" % sys.argv[3])
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, "%.1f" % tot, "%.2f" % avg, "%.2f" % pct])


if __name__ == "__main__":
    main()
