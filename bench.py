#!/usr/bin/env python3
"""Benchmark of the ME-TRPO inner loop on MI355X.

One "step" = one iteration of the reference's TRPO loop body (model_based_rl.py:1174-1179):
    algo.start_worker(); paths = algo.obtain_samples(j); samples = algo.process_samples(j, paths)
    algo.optimize_policy(j, samples)
on the configuration BASELINE.json quotes the metric on (C1: Swimmer, K=5 models 2x64, policy 2x32,
B=5000 imagined envs, H=100, step_rand, TRPO max-KL 0.01), synthetic weights/initial states.
`value` = K*B*H*n_gpus imagined env-steps per second over the WHOLE iteration (rollout + GAE/baseline +
TRPO update), all K heads evaluated per env-step as the reference does (env_helpers.py:612).
B is per GPU (weak scaling); the only cross-rank traffic is the small sum all-reduces of parallel.py.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch


def flops_per_unit(ns, na, n_drop, dyn_hidden, pol_hidden):
    """Algorithmic FLOPs: per evaluated (k,b,h) dynamics forward, and per (b,h) policy forward."""
    d = [ns + na - n_drop] + list(dyn_hidden) + [ns]
    p = [ns] + list(pol_hidden) + [na]
    return (2 * sum(d[i] * d[i + 1] for i in range(len(d) - 1)), 2 * sum(p[i] * p[i + 1] for i in range(len(p) - 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='C1')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-H', type=int, default=50, help='horizon of the bounded CPU-baseline sample')
    args = ap.parse_args()

    import metrpo_amd
    from metrpo_amd import synthetic
    # test hooks (tests/test_gpu_api.py): METRPO_BENCH_BACKEND=gloo + METRPO_BENCH_DEVICE=0 run N ranks on ONE GPU so the
    # multi-rank control flow (collectives, barriers, max-over-ranks timing) is exercised on a 1-GPU box
    comm = metrpo_amd.Comm.init_from_env(os.environ.get('METRPO_BENCH_BACKEND', 'nccl'))
    assert comm.world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = int(os.environ.get('METRPO_BENCH_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)

    cfg = synthetic.CONFIGS[args.config]
    env, K, B, H = cfg['env'], cfg['K'], cfg['B'], cfg['H']
    ns, na, n_drop = synthetic.ENV_SPECS[env]
    eng = metrpo_amd.Engine(env, K, cfg['dyn_hidden'], cfg['pol_hidden'], device=dev)
    Ws, bs, norm = synthetic.make_dynamics(env, K, cfg['dyn_hidden'], seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=0)
    baseline = metrpo_amd.LinearFeatureBaseline()
    init = metrpo_amd.InitStatePool(synthetic.make_pool(env), na)
    nne = metrpo_amd.NeuralNetEnv(env=init, inner_env=None, cost_np=env, dynamics_in=None, dynamics_outs=eng,
                                  sam_mode='step_rand')
    algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=baseline, batch_size=B * H, max_path_length=H,
                           discount=1.0, step_size=0.01, sampler_args=dict(n_envs=B), comm=comm, seed=0)
    algo.defer_baseline_fit = True            # host solve of the 24x24 baseline system overlaps the next rollout
    algo.reuse_trajectory_buffers = True      # one set of [T,B,.] tensors, overwritten every iteration

    ev_roll = []

    def step(j, timed):
        algo.rollout_events = ev_roll if timed else None      # HIP events recorded around the rollout launch itself
        algo.start_worker()
        paths = algo.obtain_samples(j)
        samples = algo.process_samples(j, paths)
        algo.optimize_policy(j, samples)

    for j in range(args.warmup):
        step(j, False)
    comm.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(args.steps):
        step(args.warmup + j, True)
    comm.barrier(); torch.cuda.synchronize()
    dt = comm.max_float(time.perf_counter() - t0, device='cuda')

    roll_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_roll])) if ev_roll else float('nan')
    roll_ms = comm.max_float(roll_ms, device='cuda')
    units_per_step = K * B * H * comm.world
    ms_per_step = dt / args.steps * 1e3
    f_dyn, f_pol = flops_per_unit(ns, na, n_drop, cfg['dyn_hidden'], cfg['pol_hidden'])
    flops_launch = K * B * H * f_dyn + B * H * f_pol            # one rollout launch on one GPU
    achieved = flops_launch / (roll_ms * 1e-3) / 1e12
    PEAK_F32 = 157.3                                            # TFLOP/s, dense f32 MFMA = f32 vector peak (MI355X_MICROARCH.md)
    traffic = None                                              # HBM bytes per launch from rocprofv3 PMC (profiles/, measured offline)
    tpath = os.path.join(REPO, 'profiles', 'r01_rollout_traffic.json')
    if args.config == 'C1' and eng.has_mfma_path and os.path.exists(tpath):
        traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
    out = {
        "metric": "imagined env-steps/sec (KxBxH) over the full TRPO iteration", "value": units_per_step / (dt / args.steps),
        "unit": "env-steps/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s Swimmer-class rollout+GAE+TRPO: env=%s K=%d dyn=%s policy=%s B=%d/GPU H=%d sam_mode=step_rand "
                               "all-K-heads-evaluated max_kl=0.01 cg_iters=10" % (args.config, env, K, list(cfg['dyn_hidden']),
                                                                              list(cfg['pol_hidden']), B, H),
                   "parallelism": "B-sharded x%d, sum all-reduce of g/FVP/scalars" % comm.world},
        "trpo_iter_ms": ms_per_step,
        "rollout": {"ms": roll_ms, "env_steps_per_s": units_per_step / (roll_ms * 1e-3),
                    "kernel": {3: "gemm-stepwise", 2: "mfma-cooperative", 1: "mfma-head-per-wave", 0: "generic"}[eng.set_rollout_variant(0)]},
        "roofline": {"bound": "mfma", "kernel": "rollout", "achieved": achieved, "peak": PEAK_F32, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F32, "traffic": traffic,
                     "hbm_frac_unfused_88B": (K * B * H * (2 * ns + na) * 4) / (roll_ms * 1e-3) / 8e12},
    }
    if comm.rank == 0 and comm.world == 1 and not args.no_cpu_baseline:
        try:
            from threadpoolctl import threadpool_limits
            from oracle import cpu_baseline
            with threadpool_limits(limits=1):
                cb = cpu_baseline.run_iteration(env, K, cfg['dyn_hidden'], cfg['pol_hidden'], B=B, H=args.cpu_H, seed=0)
            out["cpu_baseline"] = {"value": cb['units'] / cb['seconds'], "unit": "env-steps/s", "cores": 1, "kind": "port",
                                   "sample": "one full iteration (obtain_samples+process_samples+optimize_policy) at B=%d, H=%d "
                                             "(N=%d samples), float32 NumPy, 1 thread; breakdown_s=%s"
                                             % (B, args.cpu_H, B * args.cpu_H, {k: round(v, 3) for k, v in cb['breakdown'].items()})}
        except Exception as e:                                  # the baseline is a report, never a reason to lose the GPU line
            out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
    if comm.rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
