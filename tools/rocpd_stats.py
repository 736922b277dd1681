#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / min / max (us)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-70s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for n, c, s, a, mn, mx in rows:
    print("%-70s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n[:70], c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
