"""GPU tests of the drop-in host interface (NeuralNetEnv / VecSimpleEnv / GaussianMLPPolicy /
VectorizedSampler / TRPO / early stopping) against the oracle's restatement of the reference."""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
import helpers as Hh
import tolerances as TOL

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


def build_algo(env='swimmer', K=5, B=64, H=10, batch=None, sam_mode='step_rand', seed=0, gamma=0.99, lam=0.95, dyn_hidden=(64, 64)):
    import metrpo_amd
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, dyn_hidden, (32, 32), seed=seed)
    policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=seed)
    eng.set_policy(theta)
    init = metrpo_amd.InitStatePool(pool, dm.na)
    nne = metrpo_amd.NeuralNetEnv(env=init, inner_env=None, cost_np=env, dynamics_in=None, dynamics_outs=eng, sam_mode=sam_mode)
    algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=metrpo_amd.LinearFeatureBaseline(), batch_size=batch or B * H,
                           max_path_length=H, discount=gamma, gae_lambda=lam, step_size=0.01, sampler_args=dict(n_envs=B))
    return algo, eng, dm, theta, pdims, pool


def test_vec_env_step_api_matches_reference_restatement():
    """VecSimpleEnv.reset/step with the reference's own np.random call pattern: seeding np.random identically
    for the device env and the oracle env must give the same trajectory (fp32 vs fp64 tolerance)."""
    import metrpo_amd
    for env, sam_mode in (('swimmer', 'step_rand'), ('ant', 'eps_rand'), ('swimmer', 'model_mean_std')):
        K, B, H, T = 4, 12, 4, 9
        eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (64, 64), (32, 32), seed=41)
        if env == 'ant':
            pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
            eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
        pool32 = pool.astype(np.float32).astype(np.float64)
        actions = np.random.RandomState(1).randn(T, B, dm.na) * 0.8
        np.random.seed(77)
        nne = metrpo_amd.NeuralNetEnv(metrpo_amd.InitStatePool(pool32, dm.na), None, env, None, eng, sam_mode)
        assert nne.vectorized and nne.observation_space.shape == (dm.ns,)
        ve = nne.vec_env_executor(n_envs=B, max_path_length=H)
        first = ve.reset()
        got = [ve.step(actions[t]) for t in range(T)]
        np.random.seed(77)
        pr = Hh_pool_reset(pool32)
        ref_env = O.VecEnvOracle(env, lambda s, a: O.dynamics_forward_all(dm, s, a), K, B, dm.ns, H, sam_mode, pr)
        ref_first = ref_env.reset()
        np.testing.assert_allclose(first, ref_first, atol=1e-6)
        for t in range(T):
            s, r, d, _ = ref_env.step(actions[t])
            np.testing.assert_allclose(got[t][0], s, **TOL.FREE_RUN)
            np.testing.assert_allclose(got[t][1], r, **TOL.FREE_RUN)
            assert np.array_equal(got[t][2], d)
            assert np.array_equal(ve.ts, ref_env.ts) or t < T - 1
        assert ve.num_envs == B and got[0][3] == {}
        ve.terminate()


def Hh_pool_reset(pool):
    from conftest import PoolReset
    return PoolReset(pool)


def test_policy_get_actions_surface():
    algo, eng, dm, theta, pdims, pool = build_algo()
    obs = pool[:7]
    np.random.seed(5)
    a, info = algo.policy.get_actions(obs)
    np.random.seed(5)
    eps = np.random.normal(size=(7, dm.na))
    ra, rinfo = O.policy_get_actions(theta.astype(np.float32).astype(np.float64), pdims, obs.astype(np.float32).astype(np.float64), eps)
    np.testing.assert_allclose(a, ra, **TOL.ACTION)
    np.testing.assert_allclose(info['mean'], rinfo['mean'], **TOL.STEP)
    np.testing.assert_allclose(info['log_std'], rinfo['log_std'], atol=1e-6)
    assert algo.policy.vectorized and not algo.policy.recurrent
    th = algo.policy.get_param_values()
    algo.policy.set_param_values(th * 0.5)
    np.testing.assert_allclose(algo.policy.get_param_values(), (th * 0.5).astype(np.float32), atol=0)
    algo.policy.reset_log_std()
    assert np.allclose(algo.policy.get_param_values()[-dm.na:], 0.0)


@pytest.mark.parametrize('env,batch', [('swimmer', 640), ('swimmer', 1500), ('ant', 900)])
def test_sampler_process_samples_pipeline(env, batch):
    """start_worker / obtain_samples / process_samples vs BaseSampler.process_samples restated by the oracle,
    through the list-of-paths view of the device trajectory (completion order, whole rounds, dropped tails)."""
    B, H = 64, 10
    algo, eng, dm, theta, pdims, pool = build_algo(env, B=B, H=H, batch=batch)
    if env == 'ant':
        pool[::3, 2] = 0.21
        algo.env.env.states[::3, 2] = 0.21; algo.env.env._dev = None
    prev = np.random.RandomState(3).randn(2 * dm.ns + 4) * 0.05
    algo.baseline.set_param_values(prev.copy())
    algo.start_worker()
    paths = algo.obtain_samples(0)
    rounds = -(-batch // (B * H))
    plist = paths.to_paths()
    if env != 'ant':
        assert paths.traj.T == rounds * H
    else:      # step-granular stop (vectorized_sampler.py:60,104): the LAST step is the first at which completed samples >= batch
        done, tp = cpu(paths.traj.done), cpu(paths.traj.tpath)
        per_step = (done * (tp + 1)).sum(axis=1).cumsum()
        assert per_step[-1] >= batch and (paths.traj.T == 1 or per_step[-2] < batch)
    assert sum(len(p['rewards']) for p in plist) >= batch                 # quirk 6: n_samples counts completed paths
    if env != 'ant':
        assert len(plist) == rounds * B and all(len(p['rewards']) == H for p in plist)
    else:
        assert len({len(p['rewards']) for p in plist}) > 1
    samples = algo.process_samples(0, paths)
    base = O.LinearFeatureBaselineOracle(); base._coeffs = prev.copy()
    ref = O.process_samples([dict(p) for p in plist], base, algo.discount, algo.gae_lambda, center_adv=True)
    v = cpu(samples['valids']).astype(bool)
    assert v.sum() == len(ref['advantages']) == samples['n_valid_global']
    # same multiset of samples: compare after sorting by (reward, return) -- time-major vs path-major order differ
    def srt(ret, adv):
        o = np.lexsort((adv, ret)); return ret[o], adv[o]
    gr, ga = srt(cpu(samples['returns'])[v], cpu(samples['advantages'])[v])
    rr, ra = srt(ref['returns'], ref['advantages'])
    np.testing.assert_allclose(gr, rr, **TOL.RETURNS)
    np.testing.assert_allclose(ga, ra, **TOL.ADVANTAGE_CENTRED)
    # baseline refit AFTER the advantages were computed (quirk 8)
    feat_pred = lambda c: np.concatenate([O.LinearFeatureBaselineOracle.features(p) for p in plist]) @ c
    np.testing.assert_allclose(feat_pred(algo.baseline.coeffs), feat_pred(base._coeffs), rtol=0,
                               atol=TOL.BASELINE_FIT * max(1.0, np.abs(feat_pred(base._coeffs)).max()))
    assert not np.allclose(algo.baseline.coeffs, prev)


def test_params_file_shape_iteration_on_the_resident_kernel():
    """The reference's own params-swimmer.json shape (K = 5, 2 x 512 nets, 100 envs, several rounds per batch) through the host interface:
    obtain_samples lands on the resident rollout kernel (all rounds in one launch), the list-of-paths view, process_samples and the
    TRPO step behave as on the other kernels."""
    B, H, batch = 100, 10, 2500                                              # -> 3 rounds of 100 envs x 10 steps
    algo, eng, dm, theta, pdims, pool = build_algo('swimmer', B=B, H=H, batch=batch, dyn_hidden=(512, 512))
    prev = np.random.RandomState(3).randn(2 * dm.ns + 4) * 0.05
    algo.baseline.set_param_values(prev.copy())
    algo.start_worker()
    paths = algo.obtain_samples(0)
    assert eng.last_rollout_kernel() == 'resident'
    assert paths.traj.T == 3 * H
    plist = paths.to_paths()
    assert len(plist) == 3 * B and all(len(p['rewards']) == H for p in plist)
    samples = algo.process_samples(0, paths)
    base = O.LinearFeatureBaselineOracle(); base._coeffs = prev.copy()
    ref = O.process_samples([dict(p) for p in plist], base, algo.discount, algo.gae_lambda, center_adv=True)
    v = cpu(samples['valids']).astype(bool)
    assert v.sum() == len(ref['advantages']) == samples['n_valid_global'] == 3 * B * H
    srt = lambda ret, adv: tuple(x[np.lexsort((adv, ret))] for x in (ret, adv))
    gr, ga = srt(cpu(samples['returns'])[v], cpu(samples['advantages'])[v])
    rr, ra = srt(ref['returns'], ref['advantages'])
    np.testing.assert_allclose(gr, rr, **TOL.RETURNS)
    np.testing.assert_allclose(ga, ra, **TOL.ADVANTAGE_CENTRED)
    assert algo.optimize_policy(0, samples) == dict()
    d = algo.optimizer.last_diag
    assert np.isfinite(d['loss_before']) and (not d['accepted'] or (d['kl'] <= 0.01 and d['loss'] < d['loss_before']))
    # the next iteration rolls out under the updated policy, again in one launch
    algo.start_worker()
    paths2 = algo.obtain_samples(1)
    assert eng.last_rollout_kernel() == 'resident' and not torch.equal(paths2.traj.act, paths.traj.act) if d['accepted'] else True


def test_trpo_iteration_and_early_stopping_loop():
    import metrpo_amd
    algo, eng, dm, theta, pdims, pool = build_algo('swimmer', B=256, H=20, gamma=1.0, lam=1.0)
    th0 = eng.get_policy().clone()
    for j in range(3):
        algo.start_worker()
        paths = algo.obtain_samples(j)
        samples = algo.process_samples(j, paths)
        assert algo.optimize_policy(j, samples) == dict()
        d = algo.optimizer.last_diag
        assert d['accepted'] and d['kl'] <= 0.01 and d['loss'] < d['loss_before']
    assert not torch.equal(th0, eng.get_policy())
    val0 = pool[:128]
    out = metrpo_amd.early_stop.optimize_policy(algo, val0, T=15, gamma=1.0, mode='estimated', whole=True, log_every=2,
                                               num_iters_threshold=4, max_iters=8, reset_log_std=True)
    assert 0 <= out['best_index'] <= out['last_index'] <= 8 and len(out['history']) >= 1
    assert out['min_validation_costs']['estimated'].shape == (5,)
    # after the loop the policy is the best snapshot: its validation cost equals the recorded minimum when whole=True
    final = cpu(eng.validation_cost(val0, 15, 1.0))
    if out['best_index'] > 0:
        np.testing.assert_allclose(final, out['min_validation_costs']['estimated'], rtol=1e-5, atol=1e-6)
    # trpo_mean mode runs the determ=True sampling path (model_based_rl.py:1221)
    out2 = metrpo_amd.early_stop.optimize_policy(algo, val0, T=15, gamma=1.0, mode='trpo_mean', log_every=1,
                                                num_iters_threshold=2, max_iters=3, reset_log_std=False)
    assert out2['last_index'] <= 3


def test_allreduce_callback_through_rccl_world1():
    """The fused C driver calls back into Python for every all-reduce; here the callback runs a REAL RCCL all-reduce
    (process group of size 1) on a zero-copy view of the library-owned float64 buffers.  Result must be bitwise equal
    to the run without the callback."""
    import os, socket
    import torch.distributed as dist
    import metrpo_amd
    from test_gpu_engine import _update_problem
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=4000, seed=33)
    batch = eng.make_batch(obs, act, adv, om, ols)
    theta0 = eng.get_policy().clone()
    ref = eng.trpo_update(batch, want_vectors=True)
    theta_ref = eng.get_policy().clone()
    eng.set_policy(theta0)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        comm = metrpo_amd.Comm(always_reduce=True)
        calls = []
        def ar(t):
            assert t.dtype == torch.float64 and t.is_cuda and t.numel() in (eng.P + 1, eng.P, 2)
            calls.append(t.numel())
            comm.allreduce_sum_(t)
        out = eng.trpo_update(batch, allreduce=ar, want_vectors=True)
        torch.cuda.synchronize()
        # loss+gradient, 10 CG products (the step scale comes from the CG recurrence; explicit_final_hvp=True adds the 11th), line search
        assert calls[0] == eng.P + 1 and calls.count(eng.P) == 10 and calls[-1] == 2
        assert torch.equal(eng.get_policy(), theta_ref)
        assert torch.equal(out['g'], ref['g']) and torch.equal(out['d'], ref['d'])
        # the object-level path: sampler statistics + optimizer through Comm
        stats = torch.ones(3, dtype=torch.float64, device='cuda')
        assert torch.equal(comm.allreduce_sum_(stats), torch.ones(3, dtype=torch.float64, device='cuda'))
    finally:
        dist.destroy_process_group()


def test_attached_rccl_communicator_world1():
    """metrpo_comm_init attaches an RCCL communicator to the ctx (here of size 1 -- a 1-GPU box): metrpo_trpo_update then issues
    ncclAllReduce itself at every exchange point, with no host callback.  Must be bitwise equal to the single-rank fused run."""
    import metrpo_amd
    from test_gpu_engine import _update_problem
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=4000, seed=34)
    batch = eng.make_batch(obs, act, adv, om, ols)
    theta0 = eng.get_policy().clone()
    ref = eng.trpo_update(batch, want_vectors=True)
    theta_ref = eng.get_policy().clone()
    eng.set_policy(theta0)
    eng.comm_init(metrpo_amd.Engine.comm_unique_id(), 1, 0)
    try:
        assert eng.comm_world == 1
        out = eng.trpo_update(batch, want_vectors=True)
        assert torch.equal(eng.get_policy(), theta_ref) and torch.equal(out['g'], ref['g']) and torch.equal(out['d'], ref['d'])
        assert out['n_backtrack'] == ref['n_backtrack'] and out['beta'] == ref['beta']
        t = torch.arange(5, dtype=torch.float64, device='cuda')
        assert torch.equal(eng.allreduce_sum_(t.clone()), t)
        with pytest.raises(metrpo_amd._lib.MetrpoError, match='already attached'):
            eng.comm_init(metrpo_amd.Engine.comm_unique_id(), 1, 0)
    finally:
        eng.comm_destroy()
    with pytest.raises(metrpo_amd._lib.MetrpoError, match='no communicator'):
        eng.allreduce_sum_(torch.zeros(2, dtype=torch.float64, device='cuda'))


@pytest.mark.parametrize('path', ['mfma', 'gemm'])
def test_two_ranks_equal_one_rank_fused_update(path, tmp_path):
    """theta(2 ranks x N/2 samples) == theta(1 rank x N samples) for the fused HIP update (metrpo_trpo_update with the all-reduce
    hook, two processes on this GPU over gloo): same g, d, beta, backtrack index and theta up to the float64 summation order."""
    import os, subprocess, sys
    from test_gpu_engine import _update_problem
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_file = str(tmp_path / 'two_rank.npz')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29519', os.path.join(root, 'tests', '_two_rank_update.py'), out_file, path]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR='127.0.0.1'), cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    two = np.load(out_file)
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=6000, seed=29)
    eng.set_update_path({'mfma': True, 'gemm': 'gemm'}[path])
    one = eng.trpo_update(eng.make_batch(obs, act, adv, om, ols), want_vectors=True)
    assert list(two['calls'][:1]) == [eng.P + 1] and list(two['calls']).count(eng.P) == 10
    # the kernels sum float32 partials over different sample groupings on 1 and 2 ranks: float32-level (1e-7 relative) differences
    # in g, amplified by the 10 CG iterations in d (SURVEY 8d allows rel-L2 1e-3 there)
    np.testing.assert_allclose(two['g'], cpu(one['g']), rtol=0, atol=TOL.MULTI_RANK_GRAD * np.abs(cpu(one['g'])).max())
    rel = np.linalg.norm(two['d'] - cpu(one['d'])) / np.linalg.norm(cpu(one['d']))
    assert rel < 1e-3
    assert abs(float(two['beta']) - one['beta']) < 1e-3 * one['beta'] and int(two['n_backtrack']) == one['n_backtrack']
    assert bool(two['accepted']) and one['accepted']
    step = np.abs(cpu(eng.get_policy()) - th).max()
    np.testing.assert_allclose(two['theta'], cpu(eng.get_policy()), rtol=0, atol=TOL.MULTI_RANK_THETA * step + 1e-7)


def test_bench_two_ranks_on_one_gpu():
    """bench.py launched the way the driver launches it for N=2 (torch.distributed.run, one process per rank), with both
    ranks on cuda:0 over gloo: exercises the B-sharded control flow, every all-reduce of the TRPO driver, the barrier /
    max-over-ranks timing and the rank-0 JSON line.  (RCCL itself needs two GPUs; world size 1 over RCCL is tested above.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, METRPO_BENCH_BACKEND='gloo', METRPO_BENCH_DEVICE='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29517', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 3 and rec['scaling'] == 'weak' and rec['value'] > 0
    assert 'cpu_baseline' not in rec                       # rank 0 at N=1 only
    assert abs(rec['value'] - 2 * 5 * 5000 * 100 / (rec['ms_per_step'] * 1e-3)) / rec['value'] < 1e-9
    # the same command carries the strong-scaling region (verdict r5 item 2b): C1's B = 5000 divided over the 2 ranks, same K / W
    st = rec['strong']
    assert st['B_per_gpu'] == 2500 and st['B_total'] == 5000 and st['steps'] == 3 and st['ms_per_step'] > 0, st
    assert abs(st['value'] - 5 * 5000 * 100 / (st['ms_per_step'] * 1e-3)) / st['value'] < 1e-9


def test_bench_strong_block_at_one_gpu_is_the_timed_region():
    """N = 1: C1 is quoted on one GPU, so the strong-scaling point IS the line's timed region (no second region is run)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert rec['strong']['value'] == rec['value'] and rec['strong']['B_per_gpu'] == 5000 and rec['config']['B_override'] is None


def test_rollout_at_a_larger_batch_does_not_wait_for_other_streams():
    """SURVEY 8b Threading / verdict r5 item 5: a launch entry point neither allocates with a device-wide wait nor synchronises.  A one-thread spin kernel keeps a
    SECOND stream busy for ~0.4 s; meanwhile metrpo_rollout + metrpo_gae are called at a batch size the context has never seen (the cooperative kernel's migration slots,
    B > 16 x CUs, and the GAE partials grow: until round 6 a hipFree + hipMalloc pair, and hipFree waits for every stream of the device).  The calls must return while
    the other stream is still busy; the trajectory equals a fresh context's at the same seed, bit for bit."""
    import time
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 5, (64, 64), (32, 32), seed=3)
    dev = eng.device
    pool_t = torch.tensor(pool, dtype=torch.float32, device=dev)
    H = T = 12
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    B1, B2 = 16 * (n_cu + 4), 16 * (2 * n_cu + 24)                 # both past one tile per CU (migration slots in use), the second larger
    out1, out2 = eng.alloc_trajectory(B1, T, H), eng.alloc_trajectory(B2, T, H)
    eng.rollout(B1, T, H, 'step_rand', pool_t, seed=5, out=out1)
    eng.gae(out1, None, 0.99, 0.95)
    torch.cuda.synchronize()
    # calibrate the spin kernel: cycles for ~0.4 s
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20000000); e1.record(); torch.cuda.synchronize()
    cycles = int(20000000 * 400.0 / max(e0.elapsed_time(e1), 1e-3))
    side = torch.cuda.Stream(device=dev)
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        torch.cuda._sleep(cycles)
        done.record(side)
    t0 = time.perf_counter()
    eng.rollout(B2, T, H, 'step_rand', pool_t, seed=6, out=out2)
    adv, ret, valid, stats = eng.gae(out2, None, 0.99, 0.95)
    dt = time.perf_counter() - t0
    still_busy = not done.query()
    torch.cuda.synchronize()
    assert still_busy and dt < 0.2, "the launch entry points waited for the other stream: returned after %.3f s, other stream busy at return: %s" % (dt, still_busy)
    n_ret, b_ret = eng.retired_workspaces()
    assert n_ret >= 1 and b_ret > 0, (n_ret, b_ret)                # the outgrown migration slots / partials were retired, not freed
    assert 'cooperative' in (eng.last_rollout_kernel() or ''), eng.last_rollout_kernel()
    eng_b = Hh.make_engine('swimmer', 5, (64, 64), (32, 32), seed=3)[0]
    ref = eng_b.rollout(B2, T, H, 'step_rand', pool_t, seed=6)
    adv_b = eng_b.gae(ref, None, 0.99, 0.95)[0]
    torch.cuda.synchronize()
    assert torch.equal(out2.obs, ref.obs) and torch.equal(out2.rew, ref.rew) and torch.equal(out2.act, ref.act)
    assert torch.equal(adv, adv_b)
    # the sweep the library runs by itself past 4 GB of retired buffers, forced: nothing retired is still in use -- the next (again larger) calls work and repeat bitwise
    assert eng.retired_workspaces(sweep=True)[0] == n_ret and eng.retired_workspaces() == (0, 0)
    B3 = B2 + 16 * 40
    c1 = eng.rollout(B3, T, H, 'step_rand', pool_t, seed=7)
    keep = c1.obs.clone()
    c2 = eng.rollout(B3, T, H, 'step_rand', pool_t, seed=7)
    torch.cuda.synchronize()
    assert torch.equal(keep, c2.obs) and eng.retired_workspaces()[0] >= 1


def test_end_to_end_outer_loop_example():
    """examples/me_trpo_loop.py: collect (surrogate real env) -> split / normalise -> train the ensemble -> TRPO with validation-cost
    early stopping, two outer iterations at tiny sizes: the rows interoperate on one context without host copies of the weights."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('me_trpo_loop', os.path.join(root, 'examples', 'me_trpo_loop.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    hist = mod.main(['--outer', '2', '--K', '3', '--T', '20', '--traj', '40', '--n-envs', '200', '--policy-iters', '6', '--model-passes', '6',
                     '--quiet'])
    assert len(hist) == 2 and hist[1]['n_data'] > hist[0]['n_data'] > 0
    assert all(np.isfinite([h['real_cost'], h['model_val'], h['est_cost']]).all() for h in hist)
    assert hist[1]['model_val'] < 10 * hist[0]['model_val'] + 1.0          # training did not blow up on the grown buffer


def test_from_params_builds_the_loop_of_a_params_file():
    """metrpo_amd.from_params on the reference's own params-swimmer.json key set (tests/golden/params_swimmer.json): the objects of
    training.py:297-372 / model_based_rl.py:373-380 with the file's shapes, and two iterations of the reference's loop body on them."""
    import os
    import metrpo_amd
    from metrpo_amd import synthetic
    s = metrpo_amd.from_params(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'params_swimmer.json'))
    eng, algo = s.engine, s.algo
    assert (eng.K, eng.dyn_hidden, eng.pol_hidden, eng.n_drop) == (5, [512, 512], [32, 32], 2)
    assert (algo.batch_size, algo.max_path_length, algo.discount, algo.step_size) == (50000, 200, 1.0, 0.01)
    assert s.optimize_policy_kwargs['T'] == 200 and s.optimize_policy_kwargs['log_every'] == 5 and s.optimize_policy_kwargs['reset_log_std'] is True
    Ws, bs, norm = synthetic.make_dynamics('swimmer', 5, (512, 512), seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    for j in range(2):
        algo.start_worker()
        paths = algo.obtain_samples(j)
        assert (paths.traj.B, paths.traj.T) == (100, 600)                 # the sampler's 100-env clamp, 3 rounds of T = 200: 60 000 >= 50 000 samples
        algo.optimize_policy(j, algo.process_samples(j, paths))
    assert eng.last_rollout_kernel() == 'resident' and eng.rollout_note() == ''
    assert np.isfinite(cpu(eng.get_policy())).all()


def test_rollout_note_names_shapes_off_the_fast_table():
    """metrpo_rollout reports the kernel family it chose, and says why when a shape falls off the fast dispatch table (once per context on stderr)."""
    for K, hid, family, needle in ((5, (64, 64), 'mfma-cooperative', ''), (3, (64, 64), 'mfma-cooperative', ''), (7, (64, 64), 'mfma-cooperative', ''), (12, (64, 64), 'gemm-stepwise', 'K = 12'), (5, (96, 96), 'gemm-stepwise', '96x96'),       # round 5: widths below 128 take the tile GEMMs, not the thread-per-env kernel
                                    (5, (8, 8), 'mfma-cooperative', ''), (5, (8, 12), 'mfma-cooperative', ''), (5, (8, 12, 8), 'generic', '8x12x8')):       # round 6: two hidden layers of at most 64 units run zero-padded on the fused kernel
        eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', K, hid, (32, 32), seed=3)
        eng.set_option('QUIET', '1')
        assert eng.get_option('quiet') == '1' and eng.get_option('STREAMK') is None and 'STREAMK' in eng.option_names()
        eng.rollout(64, 4, 4, 'step_rand', pool, seed=1)
        assert eng.last_rollout_kernel() == family, (K, hid, eng.last_rollout_kernel())
        note = eng.rollout_note()
        assert (note == '') if not needle else (needle in note), note
    with pytest.raises(Exception, match='unknown key'):
        eng.set_option('NO_SUCH_SWITCH', '1')
    with pytest.raises(Exception, match='unknown key'):          # (an unknown key is an error, an unset one is None)
        eng.get_option('NO_SUCH_SWITCH')
