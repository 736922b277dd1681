#!/usr/bin/env python3
"""Rollout time at the shapes of the reference's own params files (K = 5, B = 100 envs, trpo batch 50 000 -> rounds of H steps):
usage: params_shapes.py [env ...]   ->  ms per rollout, us per step, kernel family."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
SHAPES = {'swimmer': ((512, 512), 200), 'half_cheetah': ((1024, 1024), 100), 'hopper': ((1024, 1024), 100), 'snake': ((1024, 1024), 200)}
for env in (sys.argv[1:] or list(SHAPES)):
    hid, H = SHAPES[env]
    K, B = 5, 100
    R = -(-50000 // (B * H))
    eng = metrpo_amd.Engine(env, K, hid, (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, K, hid, seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
    pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
    out = eng.alloc_trajectory(B, R * H, H)
    for i in range(3):
        eng.rollout(B, R * H, H, 'step_rand', pool, seed=i, out=out)
    torch.cuda.synchronize()
    n = 10
    t0 = time.time()
    for i in range(n):
        eng.rollout(B, R * H, H, 'step_rand', pool, seed=10 + i, out=out)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / n * 1e3
    print('%-13s dyn %s H %d rounds %d: %.3f ms per rollout, %.2f us per step, %s, finite %s' % (
        env, hid, H, R, ms, ms * 1e3 / (R * H), eng.last_rollout_kernel(), bool(torch.isfinite(out.rew).all())))
