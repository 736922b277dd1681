#!/usr/bin/env python3
"""Host enqueue time vs GPU time of the step-loop (GEMM-path) entry points: is the loop launch-bound?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metrpo_amd
from metrpo_amd import synthetic
for env, K, B, T, dh in (('swimmer', 5, 500, 100, (512, 512)), ('swimmer', 5, 100, 100, (512, 512)), ('half_cheetah', 5, 2500, 50, (1024, 1024))):
    eng = metrpo_amd.Engine(env, K, dh, (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, K, dh, seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
    x0 = torch.as_tensor(synthetic.make_pool(env)[:B].astype(np.float32), device='cuda')
    pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
    out = eng.alloc_trajectory(B, T, T)
    for name, fn in (('validation_cost', lambda: eng.validation_cost(x0, T, 1.0)), ('bptt_grad', lambda: eng.bptt_grad(x0, T, 1.0)),
                     ('rollout', lambda: eng.rollout(B, T, T, 'step_rand', pool, seed=3, out=out))):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("%-13s dyn=%s B=%5d T=%3d %-16s enqueue %7.2f ms   total %7.2f ms" % (env, dh, B, T, name, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
