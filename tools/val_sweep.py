#!/usr/bin/env python3
"""Time metrpo_validation_cost (per-model policy costs, model_based_rl.py:1160-1163) on the MFMA and the generic kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metrpo_amd
from metrpo_amd import synthetic
for env, K, B, T, dh in (('swimmer', 5, 500, 100, (64, 64)), ('swimmer', 5, 5000, 100, (64, 64)), ('half_cheetah', 5, 5000, 200, (64, 64)), ('ant', 5, 5000, 100, (64, 64)),
                         ('swimmer', 5, 500, 100, (512, 512)), ('half_cheetah', 5, 500, 100, (1024, 1024))):
    eng = metrpo_amd.Engine(env, K, dh, (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, K, dh, seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
    x0 = torch.as_tensor(synthetic.make_pool(env)[:B].astype(np.float32), device='cuda')
    res = []
    for path in (True, False):
        eng.set_det_path(path)
        for _ in range(2): eng.validation_cost(x0, T, 1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.validation_cost(x0, T, 1.0)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    print("%-13s K=%d dyn=%s B=%5d T=%4d: fast path %8.3f ms (%7.1f M model-steps/s)   generic %8.3f ms" % (env, K, dh, B, T, res[0], K * B * T / res[0] / 1e3, res[1]), flush=True)
