#!/usr/bin/env python3
"""Small-batch multi-round rollout (the reference's own shape: B = 100, 3 rounds of 200 steps, 2x512 nets): concurrent rounds vs
METRPO_SEQ_ROUNDS=1 (sequential step loop)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
env, K, dh, B, H, R = 'swimmer', 5, (512, 512), 100, 200, 3
eng = metrpo_amd.Engine(env, K, dh, (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, dh, seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
out = eng.alloc_trajectory(B, R * H, H)
for mode in ('par', 'seq', 'par', 'seq'):
    if mode == 'seq': os.environ['METRPO_SEQ_ROUNDS'] = '1'
    else: os.environ.pop('METRPO_SEQ_ROUNDS', None)
    for i in range(2): eng.rollout(B, R * H, H, 'step_rand', pool, seed=i, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5): eng.rollout(B, R * H, H, 'step_rand', pool, seed=i, out=out)
    t1 = time.perf_counter()                      # host enqueue time
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%s: %.2f ms per rollout (host enqueue %.2f ms)' % (mode, (t2 - t0) / 5 * 1e3, (t1 - t0) / 5 * 1e3), flush=True)
