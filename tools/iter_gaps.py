#!/usr/bin/env python3
"""GPU idle time inside a bench iteration: from a rocprofv3 --kernel-trace of `bench.py --config C`, the gaps between consecutive kernels
(and copies are not in the kernel trace: a gap may hide one), per iteration = from one rollout kernel's start to the next one's.
usage (GPU box): python tools/iter_gaps.py [config]"""
import sys, os, subprocess, sqlite3, glob, tempfile, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C1'
d = tempfile.mkdtemp(dir='/tmp')
subprocess.run(['rocprofv3', '--kernel-trace', '-d', d, '-o', 't', '--', sys.executable, os.path.join(root, 'bench.py'), '--config', cfg, '--no-cpu-baseline', '--steps', '12', '--warmup', '5'],
               capture_output=True, text=True, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
db = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
roll = [i for i, r in enumerate(rows) if 'k_rollout' in r[0]]
its = list(zip(roll[6:-1], roll[7:]))              # timed iterations only
tot = busy = 0
gaps = {}
for a, b in its:
    seg = rows[a:b + 1]
    tot += seg[-1][1] - seg[0][1]
    busy += sum(e - s for _, s, e in seg[:-1])
    for (n0, s0, e0), (n1, s1, e1) in zip(seg[:-1], seg[1:]):
        key = (n0.split('(')[0].replace('void ', '')[:28], n1.split('(')[0].replace('void ', '')[:28])
        g = gaps.setdefault(key, [0, 0]); g[0] += max(0, s1 - e0); g[1] += 1
n = len(its)
print('%s: %d iterations, %.1f us per iteration, kernels busy %.1f us, idle %.1f us (%.1f %%)' % (cfg, n, tot / n / 1e3, busy / n / 1e3, (tot - busy) / n / 1e3, 100.0 * (tot - busy) / tot))
for key, (g, cnt) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print('   %7.1f us per iteration in %5.1f gaps of %5.2f us   %s -> %s' % (g / n / 1e3, cnt / n, g / cnt / 1e3, key[0], key[1]))
shutil.rmtree(d, ignore_errors=True)
