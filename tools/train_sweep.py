#!/usr/bin/env python3
"""Time one ensemble-dynamics training step (forward + backward + Adam for all K heads) at the reference's shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metrpo_amd
from metrpo_amd import synthetic

def flops(env, dh, bs, K):
    ns, na, nd = synthetic.ENV_SPECS[env]
    d = [ns + na - nd] + list(dh) + [ns]
    fwd = 2 * sum(d[i] * d[i + 1] for i in range(len(d) - 1))
    bwd = fwd + 2 * sum(d[i] * d[i + 1] for i in range(1, len(d) - 1))     # dW for every layer + dH for all but the first
    return K * bs * (fwd + bwd)

for name, env, K, dh, bs in [('C1 2x64', 'swimmer', 5, (64, 64), 1000), ('params-swimmer 2x512', 'swimmer', 5, (512, 512), 1000),
                             ('params-half-cheetah 2x1024', 'half_cheetah', 5, (1024, 1024), 1000), ('params-ant 2x1024?', 'ant', 5, (1024, 1024), 1000),
                             ('C4 3x1024 K=20', 'humanoid', 20, (1024, 1024, 1024), 1000)]:
    eng = metrpo_amd.Engine(env, K, dh, (32, 32))
    Ws, bs_, norm = synthetic.make_dynamics(env, K, dh, seed=0)
    eng.set_dynamics_layers(Ws, bs_, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    ns, na, _ = synthetic.ENV_SPECS[env]
    x = torch.randn(bs * K, ns + na, device='cuda'); y = x[:, :ns] + 0.1 * torch.randn(bs * K, ns, device='cuda')
    eng.train_reset()
    for _ in range(3): eng.train_step(x, y, bs, 1e-3, want_loss=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): eng.train_step(x, y, bs, 1e-3, want_loss=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    f = flops(env, dh, bs, K)
    print("%-28s K=%2d bs=%d: %7.3f ms/step  %6.2f TFLOP/s  %.2f M samples/s" % (name, K, bs, ms, f / ms / 1e9, K * bs / ms / 1e3), flush=True)
