#!/usr/bin/env python3
"""Time the fused rollout kernel across B / K to expose tile-quantisation and occupancy effects."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import metrpo_amd
from metrpo_amd import synthetic

def run(env, K, B, H, reps=5, dh=(64, 64)):
    eng = metrpo_amd.Engine(env, K, dh, (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, K, dh, seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
    pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
    out = eng.alloc_trajectory(B, H, H)
    for _ in range(2):
        eng.rollout(B, H, H, 'step_rand', pool, seed=1, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        eng.rollout(B, H, H, 'step_rand', pool, seed=2 + i, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("env=%s K=%d B=%6d H=%d  %8.3f ms  %7.2f us/step  %6.2f G env-steps/s" % (env, K, B, H, ms, ms * 1e3 / H, K * B * H / ms / 1e6), flush=True)

if __name__ == '__main__':
    for B in (2048, 4096, 5000, 8192, 16384, 65536):
        run('swimmer', 5, B, 100)
    run('half_cheetah', 5, 10000, 200)
    run('hopper', 5, 5000, 100)
