#!/usr/bin/env python3
"""Dev tool: time the fused rollout for pre-built library variants (tools/_variants/<name>.so), one subprocess each."""
import sys, os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
import metrpo_amd
from metrpo_amd import synthetic
env, K, H = 'swimmer', 5, 100
eng = metrpo_amd.Engine(env, K, (64, 64), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (64, 64), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
res = []
for B in (4096, 5000, 8192):
    out = eng.alloc_trajectory(B, H, H)
    for i in range(3): eng.rollout(B, H, H, 'step_rand', pool, seed=i, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): eng.rollout(B, H, H, 'step_rand', pool, seed=5 + i, out=out)
    e1.record(); torch.cuda.synchronize()
    res.append('B=%%d %%.3f ms' %% (B, e0.elapsed_time(e1) / 10))
print('  '.join(res))
''' % root
for name in sys.argv[1:]:
    shutil.copy(os.path.join(root, 'tools', '_variants', name + '.so'), os.path.join(root, 'me-trpo_amd', 'libmetrpo.so'))
    out = subprocess.run([sys.executable, '-c', CODE], capture_output=True, text=True)
    print('%-12s %s' % (name, out.stdout.strip().split('\n')[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
