#!/usr/bin/env python3
"""Time the step-wise GEMM rollout on the large BASELINE shapes (per-GPU shares), few steps each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
def flops(env, dh):
    ns, na, nd = synthetic.ENV_SPECS[env]
    d = [ns + na - nd] + list(dh) + [ns]
    return 2 * sum(d[i] * d[i + 1] for i in range(len(d) - 1))
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, env, K, dh, ph, B, T in [('hc-params-B100', 'half_cheetah', 5, (1024, 1024), (32, 32), 100, 20), ('hc-params-B500', 'half_cheetah', 5, (1024, 1024), (32, 32), 500, 20),
                                   ('C0-params', 'swimmer', 5, (512, 512), (32, 32), 100, 20), ('C2-2x1024', 'half_cheetah', 5, (1024, 1024), (32, 32), 2500, 10),
                                   ('C3', 'ant', 10, (512, 512), (32, 32), 2500, 10), ('C4', 'humanoid', 20, (1024, 1024, 1024), (100, 50, 25), 6250, 5)]:
    if only and only != name: continue
    eng = metrpo_amd.Engine(env, K, dh, ph)
    Ws, bs, norm = synthetic.make_dynamics(env, K, dh, seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, ph, eng.na))
    pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
    out = eng.alloc_trajectory(B, T, T)
    eng.rollout(B, T, T, 'step_rand', pool, seed=1, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.rollout(B, T, T, 'step_rand', pool, seed=2, out=out); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    f = K * B * T * flops(env, dh)
    print("%-10s env=%s K=%d dyn=%s B=%d: %.3f ms/step  %.2f M env-steps/s  %.1f TFLOP/s (variant %d)" % (name, env, K, dh, B, ms / T, K * B * T / ms / 1e3, f / ms / 1e9, eng.set_rollout_variant(0)), flush=True)
