#!/usr/bin/env python3
"""Rollout time (swimmer K=5 2x64, H=100) at B = 4096 / 5000 / 8192 for each pre-built library variant in tools/_variants named on the command line."""
import sys, os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, torch
sys.path.insert(0, %r)
import metrpo_amd
from metrpo_amd import synthetic
eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
Ws, bs, norm = synthetic.make_dynamics('swimmer', 5, (64, 64), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool('swimmer'), device='cuda')
res = []
for B in (4096, 5000, 8192):
    out = eng.alloc_trajectory(B, 100, 100)
    for _ in range(3): eng.rollout(B, 100, 100, 'step_rand', pool, seed=1, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): eng.rollout(B, 100, 100, 'step_rand', pool, seed=2 + i, out=out)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 10)
print(' '.join('%%.3f' %% r for r in res))
''' % root
lib = os.path.join(root, 'me-trpo_amd', 'libmetrpo.so')
shutil.copy(lib, lib + '.orig')
try:
    for name in ['shipped'] + sys.argv[1:]:
        shutil.copy(lib + '.orig' if name == 'shipped' else os.path.join(root, 'tools', '_variants', name + '.so'), lib)
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
        print('%-12s %s' % (name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
finally:
    shutil.copy(lib + '.orig', lib); os.remove(lib + '.orig')
