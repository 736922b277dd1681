#!/usr/bin/env python3
"""Benchmark of the ME-TRPO inner loop on MI355X.

One "step" = one iteration of the reference's TRPO loop body (model_based_rl.py:1174-1179):
    algo.start_worker(); paths = algo.obtain_samples(j); samples = algo.process_samples(j, paths)
    algo.optimize_policy(j, samples)
Default workload = C1, the configuration BASELINE.json quotes the metric on (Swimmer, K=5 models 2x64, policy 2x32,
B=5000 imagined envs, H=100, step_rand, TRPO max-KL 0.01), synthetic weights / initial states.  `--config C2|C2s|C3|C4|C0|C0p`
runs the other BASELINE configs at their per-GPU share (B / gpus the config is quoted on; me-trpo_amd/synthetic.py).
`value` = K*B*steps*n_gpus imagined env-steps per second over the WHOLE iteration (rollout + GAE/baseline + TRPO update),
all K heads evaluated per env-step as the reference does (env_helpers.py:612).  B is per GPU (weak scaling); the only
cross-rank traffic is the small sum all-reduces of parallel.py.

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

PEAK_F32 = 157.3            # TFLOP/s, dense f32 MFMA = f32 vector peak (MI355X_MICROARCH.md)
PEAK_HBM = 8.0e12           # B/s


def flops_per_unit(ns, na, n_drop, dyn_hidden, pol_hidden):
    """Algorithmic FLOPs: per evaluated (k,b,h) dynamics forward, and per (b,h) policy forward."""
    d = [ns + na - n_drop] + list(dyn_hidden) + [ns]
    p = [ns] + list(pol_hidden) + [na]
    return (2 * sum(d[i] * d[i + 1] for i in range(len(d) - 1)), 2 * sum(p[i] * p[i + 1] for i in range(len(p) - 1)))


def cpu_baseline_block(env, K, dyn_hidden, pol_hidden, B, H):
    """The CPU restatement (oracle/cpu_baseline.py) on the host cores: 1 thread (what the reference configures,
    utils.py:229-232) and all hardware threads."""
    import multiprocessing
    from threadpoolctl import threadpool_limits
    from oracle import cpu_baseline
    with threadpool_limits(limits=1):
        cb = cpu_baseline.run_iteration(env, K, dyn_hidden, pol_hidden, B=B, H=H, seed=0)
    block = {"value": cb['units'] / cb['seconds'], "unit": "env-steps/s", "cores": 1, "kind": "port",
             "sample": "one full iteration (obtain_samples+process_samples+optimize_policy) at B=%d, H=%d (N=%d samples), float32 "
                       "NumPy, 1 thread; breakdown_s=%s" % (B, H, B * H, {k: round(v, 3) for k, v in cb['breakdown'].items()})}
    ncpu = multiprocessing.cpu_count()
    try:
        ca = cpu_baseline.run_iteration(env, K, dyn_hidden, pol_hidden, B=B, H=max(10, H // 4), seed=0)       # BLAS free to use every core
        block["all_cores"] = {"value": ca['units'] / ca['seconds'], "unit": "env-steps/s", "cores": ncpu,
                              "sample": "same iteration at H=%d with the BLAS thread pool unrestricted (%d hardware threads)" % (max(10, H // 4), ncpu)}
    except Exception as e:
        block["all_cores"] = {"value": None, "cores": ncpu, "sample": "failed: %r" % (e,)}
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='C1')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-H', type=int, default=100, help='horizon of the bounded CPU-baseline sample')
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
    env, K, H = cfg['env'], cfg['K'], cfg['H']
    B = cfg['B'] // cfg['gpus']                               # per-GPU share of the config's B (weak scaling keeps it fixed)
    ns, na, n_drop = synthetic.ENV_SPECS[env]
    eng = metrpo_amd.Engine(env, K, cfg['dyn_hidden'], cfg['pol_hidden'], device=dev)
    Ws, bs, norm = synthetic.make_dynamics(env, K, cfg['dyn_hidden'], seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=0)
    baseline = metrpo_amd.LinearFeatureBaseline()
    init = metrpo_amd.InitStatePool(synthetic.make_pool(env), na)
    nne = metrpo_amd.NeuralNetEnv(env=init, inner_env=None, cost_np=env, dynamics_in=None, dynamics_outs=eng,
                                  sam_mode='step_rand')
    algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=baseline, batch_size=cfg.get('batch_size', B * H), max_path_length=H,
                           discount=1.0, step_size=0.01, sampler_args=dict(n_envs=B), comm=comm, seed=0)
    rccl_in_ctx = False                       # N > 1 over RCCL: the ctx owns the communicator, all-reduces issued from C
    if comm.world > 1 and os.environ.get('METRPO_BENCH_NO_CTX_COMM', '0') != '1':
        try:
            rccl_in_ctx = bool(comm.attach_engine(eng))
        except Exception as e:                # never lose the multi-GPU line: fall back to torch.distributed through the host callback
            sys.stderr.write('rank %d: ctx-owned RCCL communicator unavailable (%r); using the torch.distributed callback\n' % (comm.rank, e))
            comm.engine = None
    algo.defer_baseline_fit = True            # host solve of the 24x24 baseline system overlaps the next rollout
    algo.reuse_trajectory_buffers = True      # one set of [T,B,.] tensors, overwritten every iteration

    ev_roll, ev_upd, steps_run, n_valid = [], [], [], []

    def step(j, timed):
        algo.rollout_events = ev_roll if timed else None      # HIP events recorded around the rollout launch(es) themselves
        algo.start_worker()
        paths = algo.obtain_samples(j)
        samples = algo.process_samples(j, paths)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        algo.optimize_policy(j, samples)
        if timed:
            e1.record(); ev_upd.append((e0, e1))
            steps_run.append(paths.traj.T); n_valid.append(samples['n_valid_global'])

    for j in range(args.warmup):
        step(j, False)
    comm.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(args.steps):
        step(args.warmup + j, True)
    comm.barrier(); torch.cuda.synchronize()
    dt = comm.max_float(time.perf_counter() - t0, device='cuda')

    roll_ms = comm.max_float(float(np.mean([a.elapsed_time(b) for a, b in ev_roll])) if ev_roll else float('nan'), device='cuda')
    upd_ms = comm.max_float(float(np.mean([a.elapsed_time(b) for a, b in ev_upd])), device='cuda')
    T_mean = float(np.mean(steps_run))                          # env steps per rollout (= H except for early-terminating Ant)
    units_per_step = K * B * T_mean * comm.world
    ms_per_step = dt / args.steps * 1e3
    f_dyn, f_pol = flops_per_unit(ns, na, n_drop, cfg['dyn_hidden'], cfg['pol_hidden'])
    flops_launch = K * B * T_mean * f_dyn + B * T_mean * f_pol  # one rollout on one GPU
    achieved = flops_launch / (roll_ms * 1e-3) / 1e12
    # TRPO update, algorithmic FLOPs by SURVEY 8d: gradient = 3 x forward, each Hessian-vector product = 4 x forward, each
    # line-search evaluation = 1 x forward, per sample.  The diagnostics say how many of each this run did.
    diag = algo.optimizer.last_diag or {}
    n_hvp = int(diag.get('cg_iters_run', 10))
    n_ls = int(diag.get('n_backtrack', 0)) + 1
    N_local = float(np.mean(n_valid)) / comm.world
    upd_flops = (3 + 4 * n_hvp + n_ls) * f_pol * N_local
    upd_achieved = upd_flops / (upd_ms * 1e-3) / 1e12
    variant = eng.rollout_path()
    upd_traffic, upd_traffic_src = None, None                   # HBM bytes per policy update (all of its launches), from the OFFLINE per-sample PMC figures
    upath = os.path.join(REPO, 'profiles', 'r02_update_traffic.json')
    if args.config in ('C0', 'C0p', 'C1') and eng.update_path(int(N_local)) == 'mfma' and os.path.exists(upath):
        bps = json.load(open(upath)).get('hbm_bytes_per_sample', {})
        if all(k in bps for k in ('fvp', 'grad', 'losskl')):
            upd_traffic = float(N_local) * (n_hvp * bps['fvp'] + bps['grad'] + n_ls * bps['losskl'])
            upd_traffic_src = 'profiles/r02_update_traffic.json (rocprofv3 --pmc bytes per sample of each kernel, offline, x this run\'s launch counts)'
    traffic, traffic_src = None, None                           # HBM bytes per rollout launch: rocprofv3 PMC, measured OFFLINE (profiles/)
    tpath = os.path.join(REPO, 'profiles', 'r02_rollout_traffic.json')
    if args.config == 'C1' and variant == 2 and os.path.exists(tpath):
        traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        traffic_src = 'profiles/r02_rollout_traffic.json (rocprofv3 --pmc, offline run of the same launch)'
    out = {
        "metric": "imagined env-steps/sec (KxBxH) over the full TRPO iteration", "value": units_per_step / (dt / args.steps),
        "unit": "env-steps/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s rollout+GAE+TRPO: env=%s K=%d dyn=%s policy=%s B=%d/GPU (config B=%d on %d GPUs) H=%d sam_mode=step_rand "
                               "all-K-heads-evaluated max_kl=0.01 cg_iters=10; env steps per rollout %.1f"
                               % (args.config, env, K, list(cfg['dyn_hidden']), list(cfg['pol_hidden']), B, cfg['B'], cfg['gpus'], H, T_mean),
                   "parallelism": "B-sharded x%d, sum all-reduce of g/FVP/scalars" % comm.world},
        "trpo_iter_ms": ms_per_step,
        "rollout": {"ms": roll_ms, "env_steps_per_s": units_per_step / (roll_ms * 1e-3),
                    "kernel": {3: "gemm-stepwise", 2: "mfma-cooperative", 1: "mfma-head-per-wave", 0: "generic"}[variant]},
        "roofline": {"bound": "mfma", "kernel": "rollout", "achieved": achieved, "peak": PEAK_F32, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F32, "traffic": traffic, "traffic_source": traffic_src,
                     "hbm_frac_unfused_88B": (K * B * T_mean * (2 * ns + na) * 4) / (roll_ms * 1e-3) / PEAK_HBM,
                     "update": {"kernel": "policy update (1 gradient + %d Fisher-vector products + %d line-search evaluations, N=%d)"
                                          % (n_hvp, n_ls, int(N_local)),
                                "path": eng.update_path(int(N_local)),
                                "ms": upd_ms, "achieved": upd_achieved, "peak": PEAK_F32, "unit": "TFLOP/s", "frac": upd_achieved / PEAK_F32,
                                "traffic": upd_traffic, "traffic_source": upd_traffic_src}},
    }
    if comm.world > 1:                                          # latency of the exchanges of the path: P and 2 float64 values
        lat = {}
        for n_el in (eng.P, 2):
            buf = torch.zeros(n_el, dtype=torch.float64, device='cuda')
            for _ in range(10):
                comm.allreduce_sum_(buf)
            torch.cuda.synchronize(); comm.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                comm.allreduce_sum_(buf)
            e1.record(); torch.cuda.synchronize()
            lat["%d_f64" % n_el] = comm.max_float(e0.elapsed_time(e1) * 10.0, device='cuda')     # us per all-reduce
        out["allreduce_us"] = dict(lat, transport="rccl-in-ctx (ncclAllReduce issued by libmetrpo.so)" if rccl_in_ctx
                                   else "torch.distributed %s via host callback" % os.environ.get('METRPO_BENCH_BACKEND', 'nccl'))
    if comm.rank == 0 and comm.world == 1 and not args.no_cpu_baseline:
        try:
            # bounded sample: ~10-30 s of 1-thread CPU work (about 0.4 TFLOP of dynamics forwards), same per-step structure
            Hc = min(H, args.cpu_H) if max(cfg['dyn_hidden']) <= 64 else 10
            Bc = max(50, min(B, 5000, int(4e11 / (K * Hc * f_dyn))))
            out["cpu_baseline"] = cpu_baseline_block(env, K, cfg['dyn_hidden'], cfg['pol_hidden'], B=Bc, H=Hc)
        except Exception as e:                                  # the baseline is a report, never a reason to lose the GPU line
            out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
    if comm.rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
