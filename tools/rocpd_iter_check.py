#!/usr/bin/env python3
"""Cross-check of a bench line against its own kernel trace (rocprofv3 rocpd .db of `bench.py --steps K --warmup W`): the K iterations of the TIMED region are the
dispatches from the (W+1)-th launch of the rollout kernel to the (W+K+1)-th (the first launch of the instrumented pass).  Prints, per iteration: the span on the
device clock, the time at least one kernel was running (union of the dispatch intervals) and the plain sum of the kernel durations (which counts the overlap of a
kernel's tail with its successor's start twice).   usage: rocpd_iter_check.py <db> <warmup> <steps> [marker=k_rollout]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); W, K = int(sys.argv[2]), int(sys.argv[3])
marker = sys.argv[4] if len(sys.argv) > 4 else 'k_rollout'
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = idx[W], idx[W + K]
span = rows[b][1] - rows[a][1]
busy = 0; tot = 0; cur_s, cur_e = None, None
for n, s, e in rows[a:b]:
    tot += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("timed region: %d iterations, %d dispatches; per iteration: span %.4f ms, busy (union of kernel intervals) %.4f ms, sum of kernel durations %.4f ms"
      % (K, b - a, span / K / 1e6, busy / K / 1e6, tot / K / 1e6))
