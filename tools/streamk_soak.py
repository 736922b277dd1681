#!/usr/bin/env python3
"""Soak test of the stream-K step (hand-overs between workgroups, XCD-aware ranges, merged post/pre launch): n rollouts of a few steps at a shape whose tiles
do not divide by the CUs, every one compared bit for bit with the first of its seed; error cell checked.   python tools/streamk_soak.py [n] [env K hidden B]
SOAK_KERNEL=streamk-persistent SOAK_T=40: the persistent launch (arrivals, closers, ready flags over many steps) instead of the launch-per-step path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
env, K, hid, B = (sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ('ant', 10, 512, 2500)
T = int(os.environ.get('SOAK_T', '6'))
expect = os.environ.get('SOAK_KERNEL', 'gemm-streamk')
if expect == 'gemm-streamk':
    pass
pol = (100, 50, 25) if env == 'humanoid' else (32, 32)
dh = (hid, hid, hid) if env == 'humanoid' else (hid, hid)
eng = metrpo_amd.Engine(env, K, dh, pol)
Ws, bs, norm = synthetic.make_dynamics(env, K, dh, seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, pol, eng.na))
if expect == 'gemm-streamk':
    eng.set_option('NO_PERSIST', '1')
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
ref = {}
t0 = time.time()
for i in range(n):
    seed = i % 4
    tr = eng.rollout(B, T, max(T // 3, 2) if env == 'ant' else T, 'step_rand', pool, seed=seed)
    cur = [x.clone() for x in (tr.obs, tr.rew, tr.mean, tr.done)]
    if seed not in ref:
        ref[seed] = cur; assert eng.last_rollout_kernel() == expect, eng.last_rollout_kernel()
    else:
        for a, b in zip(ref[seed], cur):
            assert torch.equal(a, b), 'launch %d differs from the first launch of seed %d' % (i, seed)
    if i % 100 == 99:
        eng.comm_check()
eng.comm_check()
print('%s K=%d %s B=%d: %d rollouts of %d steps in %.1f s, every one bit for bit the first of its seed, no time-out' % (env, K, dh, B, n, T, time.time() - t0))
