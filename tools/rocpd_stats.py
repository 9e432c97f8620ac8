#!/usr/bin/env python3
"""Kernel summary (name, calls, total / average / min / max ns, share) of a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -- python3 ...` writes DIR/<host>/<pid>_results.db).
Usage: python tools/rocpd_stats.py results.db [> profiles/rNN_name_kernel_stats.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
for name, calls, tot, avg, mn, mx in rows:
    print(f'"{name}",{calls},{tot},{avg:.1f},{mn},{mx},{100.0 * tot / total:.2f}')
