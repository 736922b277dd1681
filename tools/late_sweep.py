#!/usr/bin/env python3
"""Stream-K below one tile per CU (SkArgs::late, mlp_streamk.h) against the tile GEMMs on the step-wise path: ms per rollout of T steps for ensembles / batches
between the params files' shapes and one tile per CU.  usage: late_sweep.py   (prints a table; option STREAMK_LATE = 0 keeps the tile GEMMs)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
CASES = [('swimmer', 5, (512, 512), 300), ('swimmer', 5, (512, 512), 600), ('swimmer', 5, (512, 512), 1000), ('swimmer', 5, (1024, 1024), 200),
         ('half_cheetah', 5, (1024, 1024), 100), ('half_cheetah', 5, (1024, 1024), 300), ('half_cheetah', 5, (1024, 1024), 800), ('ant', 10, (512, 512), 400),
         ('humanoid', 5, (1024, 1024), 100), ('humanoid', 5, (1024, 1024), 250), ('humanoid', 5, (512, 512), 500), ('humanoid', 5, (1024, 1024, 1024), 500)]
T = 40
for env, K, hid, B in CASES:
    pol = (100, 50, 25) if env == 'humanoid' else (32, 32)
    res = []
    for late in ('0', None):
        eng = metrpo_amd.Engine(env, K, hid, pol)
        Ws, bs, norm = synthetic.make_dynamics(env, K, hid, seed=0)
        eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
        eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, pol, eng.na))
        eng.set_option('NO_RESIDENT', '1'); eng.set_option('STREAMK_LATE', late)
        pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
        out = eng.alloc_trajectory(B, T, T)
        for i in range(2):
            eng.rollout(B, T, T, 'step_rand', pool, seed=i, out=out)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(5):
            eng.rollout(B, T, T, 'step_rand', pool, seed=10 + i, out=out)
        torch.cuda.synchronize()
        res.append(((time.time() - t0) / 5 * 1e3, eng.last_rollout_kernel()))
        del eng
    print('%-13s K %2d dyn %-18s B %5d: tile GEMMs %8.3f ms (%s)   default %8.3f ms (%s)   %+.1f %%' % (
        env, K, hid, B, res[0][0], res[0][1], res[1][0], res[1][1], (res[1][0] / res[0][0] - 1) * 100))
