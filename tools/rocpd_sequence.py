#!/usr/bin/env python3
"""One iteration of a rocprofv3 rocpd (.db) kernel trace as a timeline: the dispatches between the last two launches of the kernel whose
name contains <marker> (default: k_rollout; optional third argument: index of the opening launch, e.g. 3 = an iteration of bench.py's timed region), each with its duration and the idle gap since the previous dispatch ended (us)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_rollout'
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else -2            # which launch of the marker kernel opens the printed iteration (default: the last but one)
a, b = idx[k], idx[k + 1]
prev_end = rows[a - 1][2] if a > 0 else rows[a][1]
tot_k = tot_g = 0.0
for n, s, e in rows[a:b]:
    gap = (s - prev_end) / 1e3; dur = (e - s) / 1e3
    tot_k += dur; tot_g += max(gap, 0.0)
    print("%8.2f gap %8.2f us  %s" % (gap, dur, n[:90]))
    prev_end = max(prev_end, e)
print("kernels %.1f us, gaps %.1f us, span %.1f us, %d dispatches" % (tot_k, tot_g, (rows[b][1] - rows[a][1]) / 1e3, b - a))
