#!/usr/bin/env python3
"""Benchmark of the ME-TRPO inner loop on MI355X.

One "step" = one iteration of the reference's TRPO loop body (model_based_rl.py:1174-1179):
    algo.start_worker(); paths = algo.obtain_samples(j); samples = algo.process_samples(j, paths)
    algo.optimize_policy(j, samples)
Default workload = C1, the configuration BASELINE.json quotes the metric on (Swimmer, K=5 models 2x64, policy 2x32,
B=5000 imagined envs, H=100, step_rand, TRPO max-KL 0.01), synthetic weights / initial states.  `--config C2|C2s|C3|C4|C0|C0p|C0hc|C0ho|C0sn|C0an|C0hu`
runs the other BASELINE configs at their per-GPU share (B / gpus the config is quoted on; me-trpo_amd/synthetic.py).
`value` = K*B*steps*n_gpus imagined env-steps per second over the WHOLE iteration (rollout + GAE/baseline + TRPO update),
all K heads evaluated per env-step as the reference does (env_helpers.py:612).  B is per GPU (weak scaling); the only
cross-rank traffic is the small sum all-reduces of parallel.py.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process is one rank; invoked
plainly, bench.py re-executes itself under `python -m torch.distributed.run --nproc-per-node N` on 127.0.0.1 and rank 0 prints the
one JSON line.  The sum all-reduces of the path go through libmetrpo.so's one-shot direct all-reduce (peer-mapped receive regions
over xGMI; comm.hip), RCCL from C if that is unavailable, torch.distributed as the last resort -- `allreduce_us.transport` says
which.  On a box with fewer GPUs than ranks the ranks share the devices over gloo (`oversubscribed`: a functional run, not a
scaling measurement).
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


def newest_profile(suffix):
    """profiles/rNN_<...><suffix> of the highest round NN ('' when there is none): the offline PMC figures a bench line quotes are the newest committed ones."""
    import re
    best, best_r = '', -1
    pdir = os.path.join(REPO, 'profiles')
    for fn in (os.listdir(pdir) if os.path.isdir(pdir) else []):
        m = re.match(r'r(\d+)' + re.escape(suffix) + '$', fn)
        if m and int(m.group(1)) > best_r:
            best, best_r = os.path.join(pdir, fn), int(m.group(1))
    return best


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
    try:
        block["all_cores"] = cpu_all_cores_block(env, K, dyn_hidden, pol_hidden, B=B, H=max(10, H // 4))
    except Exception as e:
        block["all_cores"] = {"value": None, "cores": multiprocessing.cpu_count(), "sample": "failed: %r" % (e,)}
    return block


def cpu_all_cores_block(env, K, dyn_hidden, pol_hidden, B, H):
    """All host cores the way the path itself shards: one single-threaded process per B-shard (no BLAS pool oversubscription),
    all started together; throughput = total units / slowest shard.  The cross-shard sums of the update are NOT exchanged (each
    shard optimises on its own samples), so this is an upper bound for a sharded CPU run."""
    import multiprocessing
    import subprocess
    ncpu = multiprocessing.cpu_count()
    nproc = max(1, min(ncpu // 2 if ncpu >= 4 else ncpu, B // 8))       # physical cores (SMT siblings share the FP pipes)
    Bs = [B // nproc + (1 if i < B % nproc else 0) for i in range(nproc)]
    env_ = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1', PYTHONPATH=REPO)
    procs = [subprocess.Popen([sys.executable, '-m', 'oracle.cpu_baseline', env, str(K), ','.join(map(str, dyn_hidden)),
                               ','.join(map(str, pol_hidden)), str(b), str(H), str(i)], stdout=subprocess.PIPE, env=env_, cwd=REPO)
             for i, b in enumerate(Bs)]
    secs, units = [], 0
    for pr in procs:
        out, _ = pr.communicate(timeout=600)
        rec = json.loads(out.decode().strip().splitlines()[-1])
        secs.append(rec['seconds']); units += rec['units']
    return {"value": units / max(secs), "unit": "env-steps/s", "cores": nproc,
            "sample": "%d single-threaded processes, each one full iteration on its B-shard (%d..%d envs, H=%d), started together; "
                      "total units / slowest shard (%.2f s); %d hardware threads on the host" % (nproc, min(Bs), max(Bs), H, max(secs), ncpu)}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same flags>`."""
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')           # dmabuf IPC (RCCL and the one-shot transport both need it)
    os.environ.setdefault('OMP_NUM_THREADS', '4')
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)          # the first ~8 iterations after start-up run 5-10 % slow (clocks, allocator): 3 warm-up steps left two of them in a 20-step mean
    ap.add_argument('--config', default='C1')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak (default): every GPU runs the per-GPU share of the config (B / gpus the config is quoted on), total work grows with --gpus; '
                         'strong: the config\'s WHOLE batch B is divided over the --gpus ranks (B / N envs each: the literal reading of BASELINE.json\'s "K=5, B=5000 ... at 1/2/4/8")')
    ap.add_argument('--params', default=None, help='one of the reference\'s params/params-*.json files: run ITS shapes (overrides --config; metrpo_amd.shapes_from_params)')
    ap.add_argument('--B', type=int, default=None, help='per-GPU env count instead of the config\'s share (tools/scaling_model.py times the strong-scaling shares on ONE GPU with it; the line says so)')
    ap.add_argument('--no-strong', action='store_true', help='skip the second (strong-scaling) timed region of a --gpus N > 1 run')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-H', type=int, default=100, help='horizon of the bounded CPU-baseline sample')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)

    import metrpo_amd
    from metrpo_amd import synthetic
    # test hooks (tests/test_gpu_api.py): METRPO_BENCH_BACKEND=gloo + METRPO_BENCH_DEVICE=0 run N ranks on ONE GPU so the
    # multi-rank control flow (collectives, barriers, max-over-ranks timing) is exercised on a 1-GPU box
    n_dev = torch.cuda.device_count()
    oversub = args.gpus > n_dev                               # fewer devices than ranks: share them (RCCL refuses that; gloo + IPC works)
    backend = os.environ.get('METRPO_BENCH_BACKEND', 'gloo' if oversub else 'nccl')
    dev = int(os.environ.get('METRPO_BENCH_DEVICE', int(os.environ.get('LOCAL_RANK', '0')) % max(n_dev, 1)))
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        os.environ['LOCAL_RANK'] = str(dev)
    comm = metrpo_amd.Comm.init_from_env(backend)
    preflight_rep = None
    assert comm.world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (comm.world, args.gpus)
    if comm.world > 1 and os.environ.get('METRPO_BENCH_NO_PREFLIGHT') is None:
        # first contact with a multi-GPU box made cheap (tools/multi_gpu_preflight.py): peer access, the agreed transport, exact sums, bit-identical
        # theta after one sharded update -- on stderr, before anything is timed; a failure ends the run with its reason instead of a hang or a wrong number
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
        from multi_gpu_preflight import preflight
        try:
            ok, preflight_rep = preflight(comm.world, comm.rank, dev, latency_table=False, out=lambda *a: print(*a, file=sys.stderr, flush=True))
        except Exception as e:                                   # the check itself must not cost the run its number
            ok, preflight_rep = False, {'reason': 'preflight raised %s: %s' % (type(e).__name__, e)}
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=('cuda' if backend == 'nccl' else 'cpu'))
        torch.distributed.all_reduce(okt, op=torch.distributed.ReduceOp.MIN)
        preflight_rep['ok'] = bool(int(okt.item()) == 1)
        if not preflight_rep['ok']:
            print('[bench rank %d] multi-GPU preflight FAILED on some rank (%s): the line below carries "preflight": {"ok": false}; '
                  'METRPO_BENCH_STRICT_PREFLIGHT=1 makes this fatal' % (comm.rank, preflight_rep.get('reason', 'another rank')), file=sys.stderr, flush=True)
            if os.environ.get('METRPO_BENCH_STRICT_PREFLIGHT') is not None:
                sys.exit(3)

    if args.params:
        cfg = synthetic.config_from_params(args.params)
        args.config = 'params:' + os.path.basename(args.params)
    else:
        cfg = synthetic.CONFIGS[args.config]
    env, K, H = cfg['env'], cfg['K'], cfg['H']
    if args.scaling == 'strong':                              # the config's whole batch over the ranks of THIS run ("B/G envs each")
        B = cfg['B'] // args.gpus
        assert B >= 16, "--scaling strong: %d envs over %d ranks leaves fewer than one 16-env tile per rank" % (cfg['B'], args.gpus)
    else:
        B = cfg['B'] // cfg['gpus']                           # per-GPU share of the config's B (weak scaling keeps it fixed)
    if args.B is not None:
        B = args.B                                            # a share chosen by the caller (scaling model: what ONE rank of an N-rank strong run computes)
    ns, na, n_drop = synthetic.ENV_SPECS[env]
    def build_leg(B):
        """Engine + the reference's objects (policy, baseline, NeuralNetEnv, TRPO) for B envs on this rank, attached to the exchange transport."""
        eng = metrpo_amd.Engine(env, K, cfg['dyn_hidden'], cfg['pol_hidden'], device=dev)
        if oversub:
            eng.set_exclusive(False)                          # several ranks per device: no kernel whose workgroups wait on each other inside one launch
        Ws, bs, norm = synthetic.make_dynamics(env, K, cfg['dyn_hidden'], seed=0)
        eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
        policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=0)
        baseline = metrpo_amd.LinearFeatureBaseline()
        init = metrpo_amd.InitStatePool(synthetic.make_pool(env), na)
        nne = metrpo_amd.NeuralNetEnv(env=init, inner_env=None, cost_np=env, dynamics_in=None, dynamics_outs=eng,
                                      sam_mode='step_rand')
        algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=baseline, batch_size=cfg.get('batch_size', B * H), max_path_length=H,
                               discount=1.0, step_size=0.01, sampler_args=dict(n_envs=B), comm=comm, seed=0)
        transport = False                     # N > 1: all-reduces issued from C (one-shot exchange over peer-mapped regions, else RCCL)
        if comm.world > 1 and os.environ.get('METRPO_BENCH_NO_CTX_COMM', '0') != '1':
            try:
                transport = comm.attach_engine(eng)
            except Exception as e:            # never lose the multi-GPU line: fall back to torch.distributed through the host callback
                sys.stderr.write('rank %d: ctx-owned transport unavailable (%r); using the torch.distributed callback\n' % (comm.rank, e))
                comm.engine = None
        algo.defer_baseline_fit = True        # host solve of the 24x24 baseline system overlaps the next rollout
        algo.reuse_trajectory_buffers = True  # one set of [T,B,.] tensors, overwritten every iteration
        algo.device_baseline_fit = os.environ.get('METRPO_BENCH_HOST_BASELINE_FIT') != '1'   # the 24x24 solve of the baseline fit as a kernel: coefficients stay on the device
        algo.async_line_search = os.environ.get('METRPO_BENCH_SYNC_LINESEARCH') != '1'     # update enqueued without a host round trip per trial; closed after the next rollout is enqueued
        return eng, algo, transport

    eng, algo, transport = build_leg(B)

    from metrpo_amd.tracing import timing_event          # HIP events with a device-scope release: a torch.cuda.Event record costs the next kernel 6-15 us
    ev_roll, ev_upd, ev_iter, steps_run, n_valid = [], [], [], [], []

    def step(j, timed, events=False):
        # `events`: HIP events around the rollout launch(es), the update and the iteration.  They are NOT recorded in the timed region: every record is a
        # marker packet between two dependent kernels (3-6 us each, five per iteration = 1-1.5 % of a C1 iteration); the per-phase figures of the line come
        # from a second, instrumented pass over the same iterations behind the timed one
        algo.rollout_events = ev_roll if events else None     # HIP events recorded around the rollout launch(es) themselves
        if events:
            ei = timing_event(); ei.record(); ev_iter.append(ei)
        algo.start_worker()
        paths = algo.obtain_samples(j)
        samples = algo.process_samples(j, paths)
        if events:
            e0, e1 = timing_event(), timing_event()
            e0.record()
        algo.optimize_policy(j, samples)
        if events:
            e1.record(); ev_upd.append((e0, e1))
        if timed:
            steps_run.append(paths.traj.T); n_valid.append(samples['n_valid_global'])

    # The host meets the device once per iteration (the line search reads its trial back), so a stop-the-world pass of Python's cyclic
    # collector lands 1:1 on an iteration: one 2.2-2.4 ms iteration in every 20-step run (per-iteration events).  The iteration allocates no
    # reference cycles worth collecting; the collector is paused from before the warm-up steps to the end of the timed region
    # (what Python's own `timeit` does by default; METRPO_BENCH_GC=1 leaves it on) -- collecting right before the timed region would leave the GPU idle for tens of ms and the first
    # iterations after an idle gap run 10-30 % slow.
    import gc
    gc_was = gc.isenabled()
    if os.environ.get('METRPO_BENCH_GC') != '1':
        gc.collect(); gc.disable()
    for j in range(args.warmup):
        step(j, False)
    comm.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(args.steps):
        step(args.warmup + j, True)
    algo.optimizer.finish()                                   # the last update's line search is closed inside the timed region
    comm.barrier(); torch.cuda.synchronize()
    dt = comm.max_float(time.perf_counter() - t0, device='cuda' if backend == 'nccl' else 'cpu')
    # instrumented pass (untimed): the same iterations with HIP events around rollout / update / iteration.  Iterations of tens of ms and more
    # (C2 / C3 / C4) do not feel the events: a few of them are enough
    n_inst = args.steps if dt / args.steps < 0.05 else max(1, min(args.steps, 3))
    for j in range(n_inst):
        step(args.warmup + args.steps + j, False, events=True)
    algo.optimizer.finish()
    ei = timing_event(); ei.record(); ev_iter.append(ei)
    comm.barrier(); torch.cuda.synchronize()
    if gc_was:
        gc.enable()
    # the time-dominant kernel of the update, measured live: one more (untimed) iteration with HIP events around every Fisher-vector-product KERNEL of
    # the CG solve (option TIME_FVP; the events sit on the update's own stream, between the kernel and its reduction)
    fvp_us, fvp_n = float('nan'), 0
    try:
        eng.set_option('TIME_FVP', '1')
        step(args.warmup + args.steps + n_inst, False)
        algo.optimizer.finish()
        torch.cuda.synchronize()
        fvp_us, fvp_n = eng.fvp_kernel_us()
    except Exception as e:                                     # a diagnostics leg must not cost the run its line
        sys.stderr.write('rank %d: FVP timing leg failed: %r\n' % (comm.rank, e))
    finally:
        eng.set_option('TIME_FVP', None)
    side = 'cuda' if backend == 'nccl' else 'cpu'             # where the bench's own bookkeeping reductions live
    iter_ms = [a.elapsed_time(b) for a, b in zip(ev_iter[:-1], ev_iter[1:])]        # per-iteration times on the stream (median of >= 20)

    roll_ms = comm.max_float(float(np.mean([a.elapsed_time(b) for a, b in ev_roll])) if ev_roll else float('nan'), device=side)
    upd_ms = comm.max_float(float(np.mean([a.elapsed_time(b) for a, b in ev_upd])), device=side)
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
    # the same update by the work the kernels EXECUTE: inside a CG solve the Fisher-vector products read the hidden activations the
    # gradient kernel cached (policy_mfma.hip MODE_FVPC) and skip the forward pass: tangent + back-prop + weight gradient = 3 x forward
    fvp_mult = 3 if eng.update_path(int(N_local)) == 'mfma' else 4
    upd_exec = (3 + fvp_mult * n_hvp + n_ls) * f_pol * N_local / (upd_ms * 1e-3) / 1e12
    variant = eng.rollout_path()
    upd_traffic, upd_traffic_src = None, None                   # HBM bytes per policy update (all of its launches), from the OFFLINE per-sample PMC figures
    upath = newest_profile('_update_traffic.json')
    if args.config in ('C0', 'C0p', 'C1') and eng.update_path(int(N_local)) == 'mfma' and os.path.exists(upath):
        bps = json.load(open(upath)).get('hbm_bytes_per_sample', {})
        if all(k in bps for k in ('fvp', 'grad', 'losskl')):
            upd_traffic = float(N_local) * (n_hvp * bps['fvp'] + bps['grad'] + n_ls * bps['losskl'])
            upd_traffic_src = 'profiles/' + os.path.basename(upath) + ' (rocprofv3 --pmc bytes per sample of each kernel, offline, x this run\'s launch counts)'
    traffic, traffic_src = None, None                           # HBM bytes per rollout launch: rocprofv3 PMC, measured OFFLINE (profiles/)
    tpath = newest_profile('_rollout_traffic.json')
    if args.config == 'C1' and variant == 2 and os.path.exists(tpath):
        traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        traffic_src = 'profiles/' + os.path.basename(tpath) + ' (rocprofv3 --pmc, offline run of the same launch)'
    rpath = newest_profile('_resident_traffic.json')
    if args.config == 'C0p' and eng.last_rollout_kernel() == 'resident' and os.path.exists(rpath):
        traffic = json.load(open(rpath)).get('hbm_bytes_per_launch')
        traffic_src = 'profiles/' + os.path.basename(rpath) + ' (rocprofv3 --pmc, offline run of the same launch; mostly the uncached step hand-over packets)'
    spath = newest_profile('_streamk_traffic.json')
    if eng.last_rollout_kernel() in ('gemm-streamk', 'streamk-persistent') and os.path.exists(spath):      # (the per-GPU share does not depend on --gpus: weak scaling)
        ent = json.load(open(spath)).get(args.config)
        if ent:                                                 # per-step bytes of the stream-K launches (the per-GPU share the file was measured at) x the steps of this rollout
            traffic = float(ent['hbm_bytes_per_step']) * T_mean
            traffic_src = 'profiles/' + os.path.basename(spath) + ' (rocprofv3 --pmc, offline run of the same launches at this per-GPU share: (2 x FETCH_SIZE + WRITE_SIZE) per step x %.0f steps; algorithmic %.3g B per step)' % (T_mean, ent['algorithmic_bytes_per_step'])
    # what bounds the rollout when it is not the matrix pipe: a 16-env tile is a chain of T dependent steps, and with fewer tiles than CUs the chip is not filled
    n_tiles = (B + 15) // 16
    kern = eng.last_rollout_kernel() or ''
    if 'cooperative' in kern or variant == 2:
        note = ('one launch; a 16-env tile is a chain of %d dependent steps (3.6 us each at 2x64), %d tiles on 256 CUs' % (int(T_mean), n_tiles)
                + (': chain-latency-bound by construction, the fraction of the matrix peak says nothing about the kernel here' if n_tiles < 200 else ''))
    elif kern == 'resident':
        note = 'one launch for the whole time loop, weights register-resident; bound by matrix-instruction issue of the compute workgroups + the step hand-over latency (B = %d: %d env tiles per round)' % (B, n_tiles)
    elif kern == 'streamk-persistent':
        note = ('one launch per rollout chunk (mlp_persist.h): whole 128 x 256 tiles of all steps in one sequence, 248 compute workgroups + closing workgroups that close a row '
                'block\'s step (reward / done / reset / policy / next input rows) as soon as its tiles have arrived; f32 MFMA issue bound')
    elif kern == 'gemm-streamk':
        note = 'per step: one stream-K launch for the K-head forward (two for three hidden layers) + one pre/post launch; f32 MFMA issue bound'
    else:
        note = 'step-wise tile GEMMs: %d rows per head and step (fewer tiles than CUs: launch-chain latency bound)' % B
    out = {
        "metric": "imagined env-steps/sec (KxBxH) over the full TRPO iteration", "value": units_per_step / (dt / args.steps),
        "unit": "env-steps/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_median": comm.max_float(float(np.median(iter_ms)), device=side), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s rollout+GAE+TRPO: env=%s K=%d dyn=%s policy=%s B=%d/GPU (%s scaling; config B=%d on %d GPUs) H=%d sam_mode=step_rand "
                               "all-K-heads-evaluated max_kl=0.01 cg_iters=10; env steps per rollout %.1f; reuse_trajectory_buffers=1 (one set of "
                               "[T,B,.] tensors overwritten every iteration) device_baseline_fit=%d (1: the fit's normal equations are solved by a kernel, float64, coefficients stay on the device; "
                               "0: host lstsq deferred behind the next rollout, defer_baseline_fit) async_line_search=%d (the accept test of the first two line-search trials runs on "
                               "the device and the host closes the update after it has enqueued the next rollout: same trials, same rule, same results)"
                               % (args.config, env, K, list(cfg['dyn_hidden']), list(cfg['pol_hidden']), B, args.scaling, cfg['B'], cfg['gpus'], H, T_mean, int(bool(algo.device_baseline_fit)), int(bool(algo.async_line_search))),
                   "parallelism": "B-sharded x%d, sum all-reduce of g/FVP/scalars" % comm.world,
                   "B_override": args.B},                      # not None: --B chose the per-GPU share (tools/scaling_model.py), not the config
        "trpo_iter_ms": ms_per_step,
        "instrumented_pass": {"iterations": n_inst, "ms_per_step": comm.max_float(float(np.mean(iter_ms)), device=side),
                              "note": "rollout.ms, roofline.update.ms and ms_per_step_median come from this second pass over the same iterations with HIP events "
                                      "around the rollout launch, the update and the iteration; the timed region (ms_per_step, value) records no events"},
        "preflight": preflight_rep,
        "rollout": {"ms": roll_ms, "env_steps_per_s": units_per_step / (roll_ms * 1e-3),
                    "kernel": eng.last_rollout_kernel() or {3: "gemm-stepwise", 2: "mfma-cooperative", 1: "mfma-head-per-wave", 0: "generic"}[variant]},
        "roofline": {"bound": "mfma", "kernel": "rollout", "achieved": achieved, "peak": PEAK_F32, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F32, "note": note, "traffic": traffic, "traffic_source": traffic_src,
                     "hbm_frac_unfused_88B": (K * B * T_mean * (2 * ns + na) * 4) / (roll_ms * 1e-3) / PEAK_HBM,
                     "update": {"kernel": "policy update (1 gradient + %d Fisher-vector products + %d line-search evaluations, N=%d)"
                                          % (n_hvp, n_ls, int(N_local)),
                                "path": eng.update_path(int(N_local)),
                                "ms": upd_ms, "achieved": upd_exec, "peak": PEAK_F32, "unit": "TFLOP/s", "frac": upd_exec / PEAK_F32,
                                "flop_count": "executed: gradient 3 x, Fisher-vector product %d x, evaluation 1 x the forward FLOPs per sample" % fvp_mult,
                                "achieved_survey_8d": upd_achieved, "frac_survey_8d": upd_achieved / PEAK_F32,
                                "traffic": upd_traffic, "traffic_source": upd_traffic_src,
                                "fvp": (None if not fvp_n else {
                                    "kernel": "Fisher-vector-product kernel of one CG iteration (the update's time-dominant kernel: %d launches per update), HIP events around the kernel itself" % n_hvp,
                                    "launches_timed": fvp_n, "us": fvp_us,
                                    "achieved": fvp_mult * f_pol * N_local / (fvp_us * 1e-6) / 1e12, "peak": PEAK_F32, "unit": "TFLOP/s",
                                    "frac": fvp_mult * f_pol * N_local / (fvp_us * 1e-6) / 1e12 / PEAK_F32,
                                    "frac_survey_8d": 4 * f_pol * N_local / (fvp_us * 1e-6) / 1e12 / PEAK_F32,
                                    "share_of_update": n_hvp * fvp_us * 1e-3 / upd_ms})}},
    }
    if comm.rank == 0:
        try:                                                    # measured peaks of THIS device next to the nominal denominators
            mf, hb = eng.probe_peaks()
            out["roofline"]["measured_peaks"] = {"f32_mfma_tflops": mf, "hbm_copy_gbs": hb, "frac_of_measured": achieved / mf,
                                                 "hbm_copy_gbs_guide": 6290.0, "note": "register-resident v_mfma_f32_32x32x2_f32 issue loop / 1 GiB streaming copy, best of 12 loop forms (csrc/probe.hip; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy); "
                                                         "`frac` keeps the nominal peak (the stricter denominator)"}
        except Exception as e:
            out["roofline"]["measured_peaks"] = {"error": repr(e)}
    out["n_devices"] = n_dev
    if oversub:
        out["oversubscribed"] = "%d ranks share %d device(s): functional run of the sharded path, not a scaling measurement" % (comm.world, n_dev)
    if comm.world > 1:                                          # latency of the exchanges of the path: 1+P, P and 2 float64 values
        lat = {}
        for n_el in (eng.P + 1, eng.P, 2):
            buf = torch.zeros(n_el, dtype=torch.float64, device='cuda')
            for _ in range(10):
                comm.allreduce_sum_(buf)
            torch.cuda.synchronize(); comm.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                comm.allreduce_sum_(buf)
            e1.record(); torch.cuda.synchronize()
            lat["%d_f64" % n_el] = comm.max_float(e0.elapsed_time(e1) * 10.0, device=side)     # us per all-reduce (stand-alone kernel; inside the update it rides in k_finalize)
        out["allreduce_us"] = dict(lat, transport={'one-shot': "one-shot direct all-reduce over peer-mapped receive regions (libmetrpo.so, comm.hip)",
                                                   'rccl': "rccl-in-ctx (ncclAllReduce issued by libmetrpo.so)"}.get(
                                                       transport, "torch.distributed %s via host callback" % backend),
                                   one_shot_error=getattr(comm, 'one_shot_error', None))
    # Second timed region of the same command: STRONG scaling (BASELINE.json reads "K=5, B=5000 ... at 1/2/4/8": the config's whole batch divided over the
    # ranks of this run, cfg['B'] / N envs each), behind the weak-scaling one above (per-GPU share fixed).  Same K steps / W warm-up, same barrier +
    # synchronize bracket, max over ranks; no events, no instrumented pass.  The driver runs one command per N and finds both scalings on its line.
    if args.scaling == 'weak' and args.B is None and not args.no_strong:
        B_s = cfg['B'] // comm.world
        if B_s == B:                                            # N = the GPU count the config is quoted on (C1 at N = 1, C4 at N = 8 ...): the same run
            out["strong"] = {"ms_per_step": ms_per_step, "value": out["value"], "unit": "env-steps/s", "B_per_gpu": B, "B_total": B * comm.world,
                             "note": "config B = %d over %d rank(s) is the weak-scaling share: the timed region above IS the strong-scaling point" % (cfg['B'], comm.world)}
        elif B_s < 16:
            out["strong"] = {"ms_per_step": None, "value": None, "unit": "env-steps/s", "B_per_gpu": B_s, "note": "fewer than one 16-env tile per rank: not run"}
        elif B_s * H * max(ns, 1) >= 2 ** 31:                   # [T, B, ns] tensors of more than 2^31 elements (C4's whole batch on ONE GPU: 2.75e9) are beyond what any parity test has exercised
            out["strong"] = {"ms_per_step": None, "value": None, "unit": "env-steps/s", "B_per_gpu": B_s,
                             "note": "config B = %d on %d rank(s): %d x %d x %d trajectory elements (> 2^31): a share no parity test covers, not run on one GPU (profiles/r06_scaling_model.json extrapolates it from B = 25 000 / 12 500)" % (cfg['B'], comm.world, H, B_s, ns)}
        else:
            # The weak-scaling result above must never be lost to the second region: if the strong leg has not finished within a generous multiple of the first
            # region's time (a hang in a collective cannot be caught as an exception), every rank leaves through the watchdog and rank 0 prints the line without it.
            import threading
            limit_s = max(180.0, 6.0 * dt * (args.steps + args.warmup) / max(args.steps, 1) + 120.0)

            def give_up():
                if comm.rank == 0:
                    out["strong"] = {"ms_per_step": None, "value": None, "unit": "env-steps/s", "B_per_gpu": B_s,
                                     "note": "strong-scaling region did not finish within %.0f s: abandoned, the line carries the weak-scaling region only" % limit_s}
                    print(json.dumps(out), flush=True)
                os._exit(0)
            watchdog = threading.Timer(limit_s, give_up)
            watchdog.daemon = True
            watchdog.start()
            try:
                if transport == 'one-shot':
                    eng.comm_ipc_detach()
                elif transport == 'rccl':
                    eng.comm_destroy()
                comm.engine = None
                algo.optimizer.finish()
                torch.cuda.synchronize()
                eng_s, algo_s, transport_s = build_leg(B_s)
                T_s = []

                def step_s(j, timed):
                    algo_s.rollout_events = None
                    algo_s.start_worker()
                    paths = algo_s.obtain_samples(j)
                    samples = algo_s.process_samples(j, paths)
                    algo_s.optimize_policy(j, samples)
                    if timed:
                        T_s.append(paths.traj.T)
                gc.collect()
                if os.environ.get('METRPO_BENCH_GC') != '1':
                    gc.disable()
                for j in range(args.warmup):
                    step_s(j, False)
                comm.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for j in range(args.steps):
                    step_s(args.warmup + j, True)
                algo_s.optimizer.finish()
                comm.barrier(); torch.cuda.synchronize()
                dt_s = comm.max_float(time.perf_counter() - t0, device=side)
                if gc_was:
                    gc.enable()
                out["strong"] = {"ms_per_step": dt_s / args.steps * 1e3, "value": K * B_s * float(np.mean(T_s)) * comm.world / (dt_s / args.steps),
                                 "unit": "env-steps/s", "B_per_gpu": B_s, "B_total": B_s * comm.world, "steps": args.steps, "warmup": args.warmup,
                                 "rollout_kernel": eng_s.last_rollout_kernel(), "transport": ("single rank: no exchange" if comm.world == 1 else (transport_s or ("torch.distributed %s via host callback" % backend))),
                                 "note": "second timed region of this command: config B = %d divided over %d ranks (strong scaling; `value` / `ms_per_step` of the line are the "
                                         "weak-scaling region, %d envs per rank)" % (cfg['B'], comm.world, B)}
            except Exception as e:                              # the second region must not cost the run its line
                out["strong"] = {"ms_per_step": None, "value": None, "unit": "env-steps/s", "B_per_gpu": B_s, "note": "strong-scaling region failed: %r" % (e,)}
            watchdog.cancel()
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
