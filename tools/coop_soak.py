#!/usr/bin/env python3
"""Soak of the cooperative rollout kernel's round-6 forms: K = 6 ... 10 heads (one workgroup per CU, tiles migrating through stamped hand-over slots when there are
more tiles than CUs) and the zero-padded narrow nets: n launches per shape at random batch sizes, every launch repeated with the same seed and compared bit for bit,
the sticky rollout-error cell checked (a hand-over that timed out).   python tools/coop_soak.py [n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metrpo_amd
from metrpo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(0)
for env, K, hid in (('swimmer', 10, (64, 64)), ('hopper', 9, (64, 64)), ('snake', 8, (64, 64)), ('ant', 8, (64, 64)), ('half_cheetah', 7, (64, 64)), ('swimmer', 6, (64, 64)),
                    ('swimmer', 5, (48, 40)), ('ant', 3, (20, 64))):
    eng = metrpo_amd.Engine(env, K, hid, (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, K, hid, seed=1)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
    pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
    t0 = time.time(); kinds = set()
    for i in range(n):
        B = int(rng.randint(4100, 9000)) if i % 3 else int(rng.randint(16, 4096))
        T = int(rng.randint(3, 25))
        a = eng.rollout(B, T, 7, 'step_rand', pool, seed=1000 + i)
        keep = (a.obs.clone(), a.rew.clone(), a.done.clone())
        b = eng.rollout(B, T, 7, 'step_rand', pool, seed=1000 + i)
        torch.cuda.synchronize()
        assert torch.equal(keep[0], b.obs) and torch.equal(keep[1], b.rew) and torch.equal(keep[2], b.done), (env, K, hid, B, T, i)
        assert bool(torch.isfinite(b.obs).all())
        kinds.add(eng.last_rollout_kernel())
    eng.comm_check()
    print('%-13s K=%2d %-9s: %d launches x 2 (B 16 ... 9000, T 3 ... 24) in %.1f s on %s, every pair bit for bit, no time-out' % (env, K, hid, n, time.time() - t0, sorted(kinds)), flush=True)
