#!/usr/bin/env python3
"""One shipped-shape validation-cost call (profiling target): half-cheetah, K=5, dynamics 2x1024, B=500, T=100."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metrpo_amd
from metrpo_amd import synthetic
env, K, dh, B, T = 'half_cheetah', 5, (1024, 1024), 500, 100
eng = metrpo_amd.Engine(env, K, dh, (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, dh, seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
x0 = torch.as_tensor(synthetic.make_pool(env)[:B].astype(np.float32), device='cuda')
for _ in range(3): eng.validation_cost(x0, T, 1.0)
torch.cuda.synchronize()
