#!/usr/bin/env python3
"""Soak of the resident kernel's validation mode: N evaluations at the params-file shapes (K = 5, 500 start states), error cell checked every 200,
every result compared bit for bit with the first."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metrpo_amd
from metrpo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for env, hid, T in (('swimmer', 512, 200), ('half_cheetah', 1024, 100), ('ant', 1024, 100)):
    eng = metrpo_amd.Engine(env, 5, (hid, hid), (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, 5, (hid, hid), seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
    x0 = torch.as_tensor(synthetic.make_pool(env)[:500].astype(np.float32), device='cuda')
    first = eng.validation_cost(x0, T, 0.99).clone()
    t0 = time.time()
    for i in range(n):
        c = eng.validation_cost(x0, T, 0.99)
        if i % 200 == 199:
            eng.comm_check()
            assert torch.equal(c, first)
    eng.comm_check()
    assert torch.equal(eng.validation_cost(x0, T, 0.99), first) and bool(torch.isfinite(first).all())
    print('%-13s 2x%d T=%d: %d evaluations, %.3f ms each, all equal, no time-out' % (env, hid, T, n, (time.time() - t0) / n * 1e3), flush=True)
