#!/usr/bin/env python3
"""Three launches of the resident rollout kernel at the params-file shape (K = 5, 2 x 512, B = 100, 3 rounds of 200 steps) for profilers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
env, K, B, H, R = 'swimmer', 5, 100, 200, 3
eng = metrpo_amd.Engine(env, K, (512, 512), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (512, 512), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
out = eng.alloc_trajectory(B, R * H, H)
for i in range(3):
    eng.rollout(B, R * H, H, 'step_rand', pool, seed=i, out=out)
torch.cuda.synchronize()
assert eng.last_rollout_kernel() == 'resident'
