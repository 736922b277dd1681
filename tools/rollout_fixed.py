#!/usr/bin/env python3
"""Cooperative rollout kernel time vs horizon at C1's batch: T(H) = fixed + H * per-step; the fixed part is prologue (weight image), first
loads and the trajectory tail.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
env, K = 'swimmer', 5
eng = metrpo_amd.Engine(env, K, (64, 64), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (64, 64), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
for B in (4096, 5000):
    res = []
    for H in (1, 2, 5, 10, 25, 50, 100, 200):
        out = eng.alloc_trajectory(B, H, H)
        for i in range(3): eng.rollout(B, H, H, 'step_rand', pool, seed=i, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10): eng.rollout(B, H, H, 'step_rand', pool, seed=i, out=out)
        e1.record(); torch.cuda.synchronize()
        res.append((H, e0.elapsed_time(e1) * 100))
    print('B=%d: ' % B + '  '.join('H=%d %.1f us' % r for r in res), flush=True)
