#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
env, K, H = 'swimmer', 5, 100
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng = metrpo_amd.Engine(env, K, (64, 64), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (64, 64), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
eng.set_rollout_variant(variant)
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
for B in (4096, 8192, 5000):
    out = eng.alloc_trajectory(B, H, H)
    for i in range(3):
        eng.rollout(B, H, H, 'step_rand', pool, seed=i, out=out)
torch.cuda.synchronize()
