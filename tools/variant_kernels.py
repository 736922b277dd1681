#!/usr/bin/env python3
"""Average duration (us) of the policy-update kernels of bench.py's C1 iteration under rocprofv3 --kernel-trace, for the shipped library
and each pre-built variant named on the command line (run on the GPU box)."""
import sys, os, shutil, subprocess, sqlite3, glob, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, 'me-trpo_amd', 'libmetrpo.so')
shutil.copy(lib, lib + '.orig')
try:
    for name in ['shipped'] + sys.argv[1:]:
        shutil.copy(lib + '.orig' if name == 'shipped' else os.path.join(root, 'tools', '_variants', name + '.so'), lib)
        d = tempfile.mkdtemp(dir='/tmp')
        subprocess.run(['rocprofv3', '--kernel-trace', '-d', d, '-o', 't', '--', sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline',
                        '--steps', '4', '--warmup', '1'], capture_output=True, text=True, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
        db = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
        if not db:
            print(name, 'no trace'); continue
        rows = sqlite3.connect(db[0]).execute("select name, count(*), avg(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
        pick = [(n, c, a) for n, c, a in rows if any(k in n for k in (os.environ.get('VK_KERNELS') or 'k_policy_mfma,k_finalize,k_rollout').split(','))]
        print('%-8s ' % name + '  '.join('%s x%d %.1f us' % (n.split('(')[0].replace('void ', '')[:34], c, a / 1e3) for n, c, a in pick), flush=True)
        shutil.rmtree(d, ignore_errors=True)
finally:
    shutil.copy(lib + '.orig', lib); os.remove(lib + '.orig')
