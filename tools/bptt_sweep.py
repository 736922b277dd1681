#!/usr/bin/env python3
"""Time one BPTT gradient (forward sweep + reverse sweep + parameter reduction) and one full step; CPU oracle beside it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import metrpo_amd
from metrpo_amd import synthetic


def flops(ns, na, n_drop, dh, ph):
    d = [ns + na - n_drop] + list(dh) + [ns]; p = [ns] + list(ph) + [na]
    fd = 2 * sum(d[i] * d[i + 1] for i in range(len(d) - 1)); fp = 2 * sum(p[i] * p[i + 1] for i in range(len(p) - 1))
    # forward sweep 1x(dyn+pol), reverse sweep: recompute 1x + input-VJP 1x, parameter reduction: policy recompute + backprop + outer products ~3x pol
    return 3 * fd + 6 * fp


def run(env, K, dh, ph, B, T, cpu=False):
    eng = metrpo_amd.Engine(env, K, dh, ph)
    Ws, bs, norm = synthetic.make_dynamics(env, K, dh, seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, ph, eng.na))
    x0 = torch.as_tensor(synthetic.make_pool(env)[:B].astype(np.float32), device='cuda')
    for _ in range(2): eng.bptt_grad(x0, T, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        c, g = eng.bptt_grad(x0, T, 1.0)
        eng.policy_adam_step(g, 1e-4, clip_val=1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ns, na, n_drop = synthetic.ENV_SPECS[env]
    fl = flops(ns, na, n_drop, dh, ph) * K * B * T
    line = "%-13s K=%d dyn=%s pol=%s B=%5d T=%4d: %8.3f ms/step  %6.2f TFLOP/s  %7.2f M (k,b,t)/s" % (env, K, dh, ph, B, T, ms, fl / ms / 1e9, K * B * T / ms / 1e3)
    if cpu:
        from oracle import metrpo_oracle as O, bptt_oracle as Bp
        from threadpoolctl import threadpool_limits
        dm, theta, pdims, pool = O.make_problem(env, K=K, dyn_hidden=dh, pol_hidden=ph, seed=0)
        Bc, Tc = min(B, 500), min(T, 20)
        with threadpool_limits(limits=1):
            t0 = time.perf_counter(); Bp.policy_costs_and_grad(dm, theta, pdims, env, pool[:Bc], Tc, 1.0); dt = time.perf_counter() - t0
        line += "   | CPU oracle (NumPy f64, 1 thread, B=%d T=%d): %.2f M (k,b,t)/s" % (Bc, Tc, K * Bc * Tc / dt / 1e6)
    print(line, flush=True)


if __name__ == '__main__':
    run('swimmer', 5, (64, 64), (32, 32), 500, 100, cpu=True)
    run('swimmer', 5, (64, 64), (32, 32), 5000, 100)
    run('swimmer', 5, (512, 512), (32, 32), 500, 100, cpu=True)
    run('half_cheetah', 5, (1024, 1024), (32, 32), 500, 100)
    run('half_cheetah', 5, (64, 64), (32, 32), 2000, 200)
    run('ant', 5, (64, 64), (32, 32), 2000, 100)
