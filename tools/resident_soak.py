#!/usr/bin/env python3
"""Soak test of the resident rollout kernel: N launches at the params-file shape with changing seeds, error cell checked every 250,
first and last launch of the same seed compared bit for bit (stamps wrap nowhere near: 603 per launch of 2^32)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
env, K, B, H, R, hid = 'swimmer', 5, 100, 200, 3, 512
if len(sys.argv) > 2 and sys.argv[2] == 'wide':            # params-half-cheetah.json shape: the 4-wave workgroups, five rounds dealt to three columns
    env, H, R, hid = 'half_cheetah', 100, 5, 1024
if len(sys.argv) > 2 and sys.argv[2] in ('ant', 'wide1'):   # one round of 7 tiles on three columns: the rotating deal + sentinel wait (Ant also ends episodes on the state)
    env, H, R, hid = ('ant' if sys.argv[2] == 'ant' else 'half_cheetah'), 100, 1, 1024
mode = sys.argv[3] if len(sys.argv) > 3 else 'step_rand'   # 'one_model' / 'eps_rand': heads nobody selects run un-throttled
eng = metrpo_amd.Engine(env, K, (hid, hid), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (hid, hid), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
out = eng.alloc_trajectory(B, R * H, H)
eng.rollout(B, R * H, H, mode, pool, seed=12345, out=out)
first = [x.clone() for x in (out.obs, out.act, out.rew, out.done)]
t0 = time.time()
for i in range(n):
    eng.rollout(B, R * H, H, mode, pool, seed=i, out=out)
    if i % 250 == 249:
        eng.comm_check()
        assert bool(torch.isfinite(out.rew).all())
eng.rollout(B, R * H, H, mode, pool, seed=12345, out=out)
eng.comm_check()
assert eng.last_rollout_kernel() == 'resident'
for a, b in zip(first, (out.obs, out.act, out.rew, out.done)):
    assert torch.equal(a, b)
print('%d launches in %.1f s (%.3f ms each), no time-out, first == last for the repeated seed' % (n, time.time() - t0, (time.time() - t0) / n * 1e3))
