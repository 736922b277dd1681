#!/usr/bin/env python3
"""Three launches of the resident rollout kernel at the params-file shape (K = 5, 2 x 512, B = 100, 3 rounds of 200 steps) for profilers;
METRPO_RES_ONE=wide: the 4-wave form at params-half-cheetah.json's shape (2 x 1024, 5 rounds of 100 steps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
env, K, B, H, R, hid = 'swimmer', 5, 100, 200, 3, 512
if os.environ.get('METRPO_RES_ONE') == 'wide':
    env, H, R, hid = 'half_cheetah', 100, 5, 1024
eng = metrpo_amd.Engine(env, K, (hid, hid), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (hid, hid), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
out = eng.alloc_trajectory(B, R * H, H)
for i in range(3):
    eng.rollout(B, R * H, H, 'step_rand', pool, seed=i, out=out)
torch.cuda.synchronize()
assert eng.last_rollout_kernel() == 'resident'
