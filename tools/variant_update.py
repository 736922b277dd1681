#!/usr/bin/env python3
"""C1 policy-update time (bench.py's roofline.update.ms) and iteration time for each pre-built library variant named on the command line."""
import sys, os, shutil, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, 'me-trpo_amd', 'libmetrpo.so')
shutil.copy(lib, lib + '.orig')
try:
    for name in ['shipped'] + sys.argv[1:]:
        shutil.copy(lib + '.orig' if name == 'shipped' else os.path.join(root, 'tools', '_variants', name + '.so'), lib)
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline', '--steps', '60', '--warmup', '15'], capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print('%-10s iteration %.3f ms  rollout %.3f ms  update %.3f ms' % (name, d['ms_per_step'], d['rollout']['ms'], d['roofline']['update']['ms']), flush=True)
        except Exception:
            print(name, 'failed', out.stderr[-300:], flush=True)
finally:
    shutil.copy(lib + '.orig', lib); os.remove(lib + '.orig')
