#!/usr/bin/env python3
"""Wall time of metrpo_trpo_update at the params files' sample counts (N = 50 000 - 60 000), per value of METRPO_UPD_TILES_PER_WAVE (blocks of the MFMA
update kernels = partial rows of k_finalize).  Run on the GPU box:  python tools/upd_small.py"""
import os, sys, subprocess, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, root)
    import torch, metrpo_amd
    env, N = sys.argv[2], int(sys.argv[3])
    ns, na = {'swimmer': (10, 2), 'ant': (29, 8), 'half_cheetah': (17, 6)}[env]
    eng = metrpo_amd.Engine(env, 5, (64, 64), (32, 32))
    eng.set_policy(metrpo_amd.xavier_policy_theta(ns, (32, 32), na))
    g = torch.Generator(device='cuda').manual_seed(0)
    obs = torch.randn(N, ns, device='cuda', generator=g) * 0.5; adv = torch.randn(N, device='cuda', generator=g)
    theta0 = eng.get_policy().clone()
    act, mean = eng.policy_actions(obs, torch.randn(N, na, device='cuda', generator=g))
    b = eng.make_batch(obs, act, adv, mean, eng.get_policy()[-na:])
    ts = []
    for it in range(60):
        eng.set_policy(theta0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = eng.trpo_update(b)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[10:])
    print('%-12s N=%6d tiles/wave>=%s  update median %.1f us  min %.1f us  accepted=%s kl=%.2e' % (env, N, os.environ.get('METRPO_UPD_TILES_PER_WAVE', '1'), ts[len(ts) // 2] * 1e6, ts[0] * 1e6, r['accepted'], r['kl']), flush=True)
    sys.exit(0)
for env, N in (('swimmer', 60000), ('ant', 50000), ('swimmer', 500000)):
    for m in (1, 2, 3, 4, 6, 8, 12):
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', env, str(N)], env=dict(os.environ, METRPO_UPD_TILES_PER_WAVE=str(m)))
