#!/usr/bin/env python3
"""Dev tool: run big_sweep.py and train_sweep.py for pre-built library variants (tools/_variants/<name>.so)."""
import sys, os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:]:
    shutil.copy(os.path.join(root, 'tools', '_variants', name + '.so'), os.path.join(root, 'me-trpo_amd', 'libmetrpo.so'))
    print('==== ' + name, flush=True)
    for tool in ('big_sweep.py', 'train_sweep.py'):
        out = subprocess.run([sys.executable, os.path.join(root, 'tools', tool)], capture_output=True, text=True)
        print('\n'.join(l for l in out.stdout.splitlines() if 'ms/step' in l), flush=True)
