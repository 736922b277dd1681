#!/usr/bin/env python3
"""Cooperative rollout kernel at K = 1 ... 10 heads (K > 5: one workgroup per CU only; `two per CU` / `head per wave` then show what the dispatcher falls to): what the launch rule picks (variant 0) against two co-resident workgroups per CU (variant 2) and the
head-per-wave kernel (variant 1), B = 5000 / 8192, T = 100.  usage: coop_heads.py [env]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
env = sys.argv[1] if len(sys.argv) > 1 else 'swimmer'
for B in (5000, 8192):
    for K in (tuple(int(k) for k in sys.argv[2].split(',')) if len(sys.argv) > 2 else (1, 2, 3, 4, 5)):
        row = []
        for var in (0, 2, 1):
            eng = metrpo_amd.Engine(env, K, (64, 64), (32, 32))
            Ws, bs, norm = synthetic.make_dynamics(env, K, (64, 64), seed=0)
            eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
            eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
            eng.set_rollout_variant(var)
            pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
            out = eng.alloc_trajectory(B, 100, 100)
            for i in range(3):
                eng.rollout(B, 100, 100, 'step_rand', pool, seed=i, out=out)
            torch.cuda.synchronize(); t0 = time.time()
            for i in range(20):
                eng.rollout(B, 100, 100, 'step_rand', pool, seed=10 + i, out=out)
            torch.cuda.synchronize()
            row.append((time.time() - t0) / 20 * 1e3); kern = eng.last_rollout_kernel() if var == 1 else None
            del eng
        print('%-12s B %5d K %2d: launch rule %.3f ms | two per CU %.3f ms | head per wave %.3f ms (%s)' % (env, B, K, row[0], row[1], row[2], kern))
