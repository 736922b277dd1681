"""Full-size runs of BASELINE.json's configs (C1 whole, C2/C3/C4 at their per-GPU share) checked through size-independent
properties: path/done structure, step indices, reward recomputed from the stored (obs, action) of the next step where the
path continues, bitwise repeatability of rollout and update, TRPO invariants (accepted => loss decreased and KL <= delta),
centred advantages, Ant's step-granular stop rule."""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
import tolerances as TOL

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: env, K, dyn hidden, policy hidden, B (per GPU), H
    'C1': ('swimmer', 5, (64, 64), (32, 32), 5000, 100),
    'C2': ('half_cheetah', 5, (1024, 1024), (32, 32), 2500, 200),
    'C2-2x64': ('half_cheetah', 5, (64, 64), (32, 32), 2500, 200),
    'C3': ('ant', 10, (512, 512), (32, 32), 2500, 500),
    'C4': ('humanoid', 20, (1024, 1024, 1024), (100, 50, 25), 6250, 1000),
    # round 6: shapes that moved to another kernel family
    'C1-K10': ('swimmer', 10, (64, 64), (32, 32), 5000, 100),            # ten heads on the cooperative kernel (one workgroup per CU, tiles migrate)
    'C1-48x40': ('swimmer', 5, (48, 40), (32, 32), 5000, 100),           # narrow nets on the cooperative kernel over zero-padded weights
    'C2-strong8': ('half_cheetah', 5, (1024, 1024), (32, 32), 1250, 200),  # C2's share of an 8-GPU strong-scaling run: per-step stream-K launches instead of the persistent one
    'C3-strong16': ('ant', 10, (512, 512), (32, 32), 1250, 500),
}


# the kernel family each shape is expected on (an exclusive 256-CU device): a silent change of the dispatch rules shows here
FAMILY = {'C1': 'mfma-cooperative', 'C2': 'streamk-persistent', 'C2-2x64': 'mfma-cooperative', 'C3': 'streamk-persistent', 'C4': 'gemm-streamk',
          'C1-K10': 'mfma-cooperative', 'C1-48x40': 'mfma-cooperative', 'C2-strong8': 'gemm-streamk', 'C3-strong16': 'gemm-streamk'}


def build(name, seed=0):
    import metrpo_amd
    env, K, dh, ph, B, H = CONFIGS[name]
    dm, theta, pdims, pool = O.make_problem(env, K=K, dyn_hidden=dh, pol_hidden=ph, seed=seed, n_pool=4096, dtype=np.float32)
    eng = metrpo_amd.Engine(env, K, dh, ph)
    eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=seed)
    eng.set_policy(theta)
    nne = metrpo_amd.NeuralNetEnv(env=metrpo_amd.InitStatePool(pool, dm.na), inner_env=None, cost_np=env, dynamics_in=None,
                                  dynamics_outs=eng, sam_mode='step_rand')
    algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=metrpo_amd.LinearFeatureBaseline(), batch_size=B * H, max_path_length=H,
                           discount=1.0, gae_lambda=1.0, step_size=0.01, sampler_args=dict(n_envs=B), seed=3)
    return algo, eng, dm, theta


@pytest.mark.parametrize('name', list(CONFIGS))
def test_full_size_iteration_properties(name):
    env, K, dh, ph, B, H = CONFIGS[name]
    algo, eng, dm, theta = build(name)
    algo.start_worker()
    paths = algo.obtain_samples(0)
    tr = paths.traj
    T = tr.T
    done, tpath = tr.done.bool(), tr.tpath.long()
    if torch.cuda.get_device_properties(eng.device).multi_processor_count == 256:
        assert eng.last_rollout_kernel() == FAMILY[name], (name, eng.last_rollout_kernel(), eng.rollout_note())
    # --- step indices and done structure
    assert int(tpath.min()) == 0 and int(tpath.max()) <= H - 1
    assert bool((tpath[0] == 0).all())
    cont = ~done[:-1]
    assert bool((tpath[1:][cont] == tpath[:-1][cont] + 1).all()) and bool((tpath[1:][~cont] == 0).all())
    assert bool(done[tpath == H - 1].all())                               # horizon reached => done (env_helpers.py:604)
    if env != 'ant':
        assert T == H and bool(done[-1].all()) and int(done.sum()) == B   # every path has length H
    else:
        lens = (done * (tpath + 1)).sum(dim=1).cumsum(0)                   # completed-path samples after each step
        assert int(lens[-1]) >= B * H and int(lens[-2]) < B * H           # vectorized_sampler.py:60,104
        assert int((done & (tpath < H - 1)).sum()) > 0                    # early terminations happened
        assert H <= T < 2 * H
    # --- continuing steps: next stored observation is a state whose reward we can recompute from (action, next obs)
    ns = eng.ns
    t_s = torch.randint(0, T - 1, (2000,), device=eng.device); b_s = torch.randint(0, B, (2000,), device=eng.device)
    keep = ~done[t_s, b_s]
    t_s, b_s = t_s[keep], b_s[keep]
    xn = tr.obs[t_s + 1, b_s].double().cpu().numpy(); x = tr.obs[t_s, b_s].double().cpu().numpy()
    u = np.clip(tr.act[t_s, b_s].double().cpu().numpy(), -1, 1)
    np.testing.assert_allclose(tr.rew[t_s, b_s].double().cpu().numpy(), -O.cost_np_vec(env, x, u, xn), **TOL.REWARD)
    assert bool(torch.isfinite(tr.obs).all()) and bool(torch.isfinite(tr.rew).all())
    # --- bitwise repeat of the rollout (same launch counter -> same Philox key)
    algo.sampler._itr_seed -= 1
    again = algo.obtain_samples(0).traj
    assert again.T == T and torch.equal(again.obs, tr.obs) and torch.equal(again.rew, tr.rew) and torch.equal(again.done, tr.done)
    del again
    # --- process_samples + TRPO update
    samples = algo.process_samples(0, paths)
    v = samples['valids'].bool()
    adv = samples['advantages'][v].double()
    assert abs(float(adv.mean())) < 1e-4 and abs(float(adv.std(unbiased=False)) - 1.0) < 1e-3
    assert int(v.sum()) == samples['n_valid_global']
    theta0 = eng.get_policy().clone()
    batch = eng.make_batch(samples['observations'], samples['actions'], samples['advantages'], samples['agent_infos']['mean'],
                           samples['agent_infos']['log_std'], valid=samples['valids'], n_global=samples['n_valid_global'])
    out = eng.trpo_update(batch)
    theta1 = eng.get_policy().clone()
    assert np.isfinite(out['loss_before']) and out['accepted']
    assert out['loss'] < out['loss_before'] and out['kl'] <= 0.01 + 1e-9
    lk = eng.loss_kl(batch).cpu().numpy()                                  # independent re-evaluation at the accepted theta
    np.testing.assert_allclose(lk, [out['loss'], out['kl']], rtol=1e-6, atol=1e-9)
    eng.set_policy(theta0)
    out2 = eng.trpo_update(batch)
    assert torch.equal(eng.get_policy(), theta1) and out2['n_backtrack'] == out['n_backtrack']      # bitwise repeat of the update


@pytest.mark.parametrize('env,H,mode', [('swimmer', 100, 'step_rand'), ('ant', 12, 'eps_rand'), ('half_cheetah', 37, 'model_mean_std')])
def test_migrating_tiles_equal_whole_tiles(env, H, mode):
    """B = 5000 is 313 tiles on 256 CUs: the cooperative kernel deals the tile-steps out evenly and hands tiles from one workgroup
    to the next in the middle of a trajectory (rollout_coop.hip).  The trajectories must be bitwise those of the same environments
    run as two launches small enough that every tile stays on one workgroup (same Philox streams through stream_offset)."""
    import metrpo_amd
    B, T = 5000, H + 3                                                      # T > H: horizon resets inside the launch as well
    dm, theta, pdims, pool = O.make_problem(env, K=5, dyn_hidden=(64, 64), pol_hidden=(32, 32), seed=4, n_pool=512, dtype=np.float32)
    eng = metrpo_amd.Engine(env, 5, (64, 64), (32, 32))
    eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    eng.set_policy(theta)
    if eng.rollout_path() != 2:
        pytest.skip('cooperative kernel not selected')
    pool_d = torch.as_tensor(pool, device=eng.device)
    ls = lambda n: (torch.empty(n, dtype=torch.int32, device=eng.device), torch.empty(n, dtype=torch.int32, device=eng.device))
    whole_ls = ls(B)
    whole = eng.rollout(B, T, H, mode, pool_d, seed=11, last_state=whole_ls)
    B1 = 2496
    parts, parts_ls = [], []
    for off, n in ((0, B1), (B1, B - B1)):
        parts_ls.append(ls(n))
        parts.append(eng.rollout(n, T, H, mode, pool_d, seed=11, stream_offset=off, last_state=parts_ls[-1]))
    torch.cuda.synchronize()
    for name in ('obs', 'act', 'rew', 'mean', 'done', 'tpath'):
        cat = torch.cat([getattr(p, name) for p in parts], dim=1)
        assert torch.equal(getattr(whole, name), cat), name
    assert torch.equal(whole.last_obs, torch.cat([p.last_obs for p in parts], dim=0))
    for i in range(2):
        assert torch.equal(whole_ls[i], torch.cat([p[i] for p in parts_ls]))
    # and a second launch on the same context (hand-over flags are epoch-stamped, not cleared)
    again = eng.rollout(B, T, H, mode, pool_d, seed=11)
    assert torch.equal(again.obs, whole.obs) and torch.equal(again.rew, whole.rew)


def test_migrating_schedule_random_shapes_equal_two_per_cu():
    """Random (B, T, H, env, sam_mode) in the window where tiles migrate between workgroups: bitwise the trajectories of the
    two-workgroups-per-CU instantiation of the same kernel (test hook 2), which runs every tile on one workgroup from start to end."""
    import metrpo_amd
    rs = np.random.RandomState(77)
    engines = {}
    for case in range(14):
        env = ['swimmer', 'hopper', 'snake', 'half_cheetah', 'ant'][case % 5]
        B = int(rs.randint(4097, 6337)); H = int(rs.randint(3, 40)); T = int(rs.randint(1, 2 * H + 2))
        mode = ['step_rand', 'eps_rand', 'model_mean', 'model_med', 'model_mean_std', 'one_model'][int(rs.randint(6))]
        if env not in engines:
            dm, theta, pdims, pool = O.make_problem(env, K=5, dyn_hidden=(64, 64), pol_hidden=(32, 32), seed=case, n_pool=300, dtype=np.float32)
            eng = metrpo_amd.Engine(env, 5, (64, 64), (32, 32))
            eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
            eng.set_policy(theta)
            engines[env] = (eng, torch.as_tensor(pool, device=eng.device))
        eng, pool_d = engines[env]
        if eng.set_rollout_variant(2) != 2:
            pytest.skip('cooperative kernel not selected')
        ref = eng.rollout(B, T, H, mode, pool_d, seed=100 + case)
        ref = {k: getattr(ref, k).clone() for k in ('obs', 'act', 'rew', 'mean', 'done', 'tpath', 'last_obs')}
        eng.set_rollout_variant(0)
        got = eng.rollout(B, T, H, mode, pool_d, seed=100 + case)
        for k, v in ref.items():
            assert torch.equal(getattr(got, k), v), (case, env, B, T, H, mode, k)


def test_chunked_ant_sampling_on_migrating_schedule_equals_two_per_cu():
    """Early-terminating env (Ant, 2x64 nets) at n_envs = 5000: obtain_samples rolls in chunks that continue from the last state and stop
    at the reference's step-granular rule; every chunk launch runs the migrating one-workgroup-per-CU schedule.  Same samples, bitwise,
    as with the two-workgroups-per-CU instantiation (test hook 2), whose tiles never change workgroup."""
    import metrpo_amd
    env, K, B, H = 'ant', 5, 5000, 40
    dm, theta, pdims, pool = O.make_problem(env, K=K, dyn_hidden=(64, 64), pol_hidden=(32, 32), seed=2, n_pool=2048, dtype=np.float32)
    pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02                          # some paths fall early
    outs = []
    for variant in (2, 0):
        eng = metrpo_amd.Engine(env, K, (64, 64), (32, 32))
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
        policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=0)
        eng.set_policy(theta)
        if eng.set_rollout_variant(variant) != 2:
            pytest.skip('cooperative kernel not selected')
        nne = metrpo_amd.NeuralNetEnv(env=metrpo_amd.InitStatePool(pool, dm.na), inner_env=None, cost_np=env, dynamics_in=None,
                                      dynamics_outs=eng, sam_mode='step_rand')
        algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=metrpo_amd.LinearFeatureBaseline(), batch_size=B * H, max_path_length=H,
                               discount=0.99, gae_lambda=0.97, step_size=0.01, sampler_args=dict(n_envs=B), seed=5)
        algo.start_worker()
        tr = algo.obtain_samples(0).traj
        outs.append({k: getattr(tr, k).clone() for k in ('obs', 'act', 'rew', 'done', 'tpath')})
        assert int((tr.done.bool() & (tr.tpath < H - 1)).sum()) > 0       # early terminations happened
        assert tr.T > H                                                   # more than one chunk
    for k in outs[0]:
        assert outs[0][k].shape == outs[1][k].shape and torch.equal(outs[0][k], outs[1][k]), k
