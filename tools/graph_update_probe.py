#!/usr/bin/env python3
"""Experiment: would a hipGraph shorten the policy update's chain of ~30 dependent launches?  The first half of the update (metrpo_trpo_update_begin: gradient,
CG with 10 Fisher-vector products, two speculative line-search trials; no synchronisation) is captured with torch.cuda.graph and replayed; GPU time per update
between stream events, captured graph vs plain launches.  (The replayed launches carry the capture's publish stamp: a timing probe, not a product path.)
    python tools/graph_update_probe.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
ns, na = 10, 2
def make():
    eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
    eng.set_policy(metrpo_amd.xavier_policy_theta(ns, (32, 32), na))
    g = torch.Generator(device='cuda').manual_seed(0)
    obs = torch.randn(N, ns, device='cuda', generator=g) * 0.5; adv = torch.randn(N, device='cuda', generator=g)
    act, mean = eng.policy_actions(obs, torch.randn(N, na, device='cuda', generator=g))
    return eng, eng.make_batch(obs, act, adv, mean, eng.get_policy()[-na:]), (obs, act, adv, mean)
eng, batch, keep = make()
theta0 = eng.get_policy().clone()
def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        eng.set_policy(theta0); torch.cuda.synchronize()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[5:]); return ts[len(ts) // 2]
def plain():
    eng.trpo_update(batch, spec_trials=2); eng.trpo_update_end()
for _ in range(3): plain()
t_plain = timed(lambda: eng.trpo_update(batch, spec_trials=2) or None, reps=1) if False else None
# plain: events around begin only (the end half only reads back / launches nothing when a speculated trial was accepted)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(30):
    eng.set_policy(theta0); torch.cuda.synchronize()
    e0.record(); eng.trpo_update(batch, spec_trials=2); e1.record(); r = eng.trpo_update_end(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts = sorted(ts[5:]); t_plain = ts[len(ts) // 2]
print('N = %d: plain launches      %.1f us per update (begin half; accepted=%s n_backtrack=%s)' % (N, t_plain, r['accepted'], r['n_backtrack']))
# captured
eng2, batch2, keep2 = make()
for _ in range(3):
    eng2.trpo_update(batch2, spec_trials=2); eng2.trpo_update_end()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        eng2.trpo_update(batch2, spec_trials=2)
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:300]); sys.exit(0)
ts = []
for _ in range(30):
    torch.cuda.synchronize()             # (theta is left to evolve: eng2's update stays open, set_policy would try to close it)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts = sorted(ts[5:])
print('N = %d: one captured graph  %.1f us per update' % (N, ts[len(ts) // 2]))
sys.stdout.flush(); os._exit(0)      # (eng2 still has an update open that never ran: skip the destructors)
