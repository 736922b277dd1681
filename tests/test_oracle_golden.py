"""Pin the CPU oracle against golden vectors captured from the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
from conftest import load_golden, ReplayRNG, dm_from_golden, PoolReset
from oracle import metrpo_oracle as O

ENVS = list(O.ENV_SPECS)


@pytest.mark.parametrize('env', ENVS)
def test_reward_kat(env):
    d = load_golden('rewards_' + env)
    got = O.cost_np_vec(env, d['x'], d['u'], d['x_next'])
    np.testing.assert_allclose(got, d['cost'], rtol=0, atol=1e-14)


def test_ant_done_kat():
    d = load_golden('ant_done')
    got = O.is_done('ant', d['x_next'], d['x_next'])
    assert got.dtype == bool and np.array_equal(got, d['done'])
    assert not O.is_done('swimmer', d['x_next'][:, :10], d['x_next'][:, :10]).any()


VEC = [('swimmer', m) for m in O.SAM_MODES] + [('ant', 'step_rand'), ('ant', 'eps_rand')]


@pytest.mark.parametrize('env,sam_mode', VEC)
def test_vecenv_trace(env, sam_mode):
    d = load_golden('vecenv_%s_%s' % (env, sam_mode))
    dm = dm_from_golden(d, env)
    rng = ReplayRNG(d['rng_kinds'], d['rng_offs'], d['rng_flat'])
    K, B, H = int(d['K']), int(d['B']), int(d['H'])
    ve = O.VecEnvOracle(env, lambda s, a: O.dynamics_forward_all(dm, s, a), K, B, dm.ns, H, sam_mode,
                        PoolReset(d['pool']), rng=rng)
    first = ve.reset()
    np.testing.assert_array_equal(first, d['first_obs'])
    for t in range(d['actions'].shape[0]):
        s, r, dn, _ = ve.step(d['actions'][t])
        np.testing.assert_allclose(s, d['states'][t], rtol=0, atol=1e-13)
        np.testing.assert_allclose(r, d['rewards'][t], rtol=0, atol=1e-13)
        assert np.array_equal(dn, d['dones'][t])
        assert np.array_equal(ve.ts, d['ts'][t])
        assert np.array_equal(ve.cur_model_idx, d['cur_idx'][t])
    assert rng.exhausted()
    if env == 'ant':
        assert d['dones'].any() and not d['dones'].all()


SAMPLERS = ['sampler_0_swimmer_step_rand', 'sampler_1_swimmer_step_rand', 'sampler_2_swimmer_eps_rand',
            'sampler_3_ant_step_rand', 'sampler_4_half_cheetah_model_mean_std']


@pytest.mark.parametrize('name', SAMPLERS)
def test_sampler_and_process_samples(name):
    d = load_golden(name)
    env, sam_mode = str(d['env']), str(d['sam_mode'])
    dm = dm_from_golden(d, env)
    rng = ReplayRNG(d['rng_kinds'], d['rng_offs'], d['rng_flat'])
    K, B, H = int(d['K']), int(d['B']), int(d['H'])
    theta, pdims = d['theta'], [int(x) for x in d['pdims']]
    ve = O.VecEnvOracle(env, lambda s, a: O.dynamics_forward_all(dm, s, a), K, B, dm.ns, H, sam_mode,
                        PoolReset(d['pool']), rng=rng)

    def get_actions(obs):
        return O.policy_get_actions(theta, pdims, np.asarray(obs), rng.normal(size=(len(obs), pdims[-1])))

    paths = O.obtain_samples(ve, get_actions, int(d['batch_size']), determ=bool(d['determ']))
    assert rng.exhausted()
    assert len(paths) == int(d['n_paths'])
    assert [len(p['rewards']) for p in paths] == list(d['lengths'])
    cat = lambda k: np.concatenate([p[k] for p in paths])
    for k in ('observations', 'actions', 'rewards'):
        np.testing.assert_allclose(cat(k), d[k], rtol=0, atol=1e-13)
    np.testing.assert_allclose(np.concatenate([p['agent_infos']['mean'] for p in paths]), d['mean'], atol=1e-13)
    if bool(d['determ']):
        np.testing.assert_array_equal(cat('actions'), d['mean'])

    base = O.LinearFeatureBaselineOracle()
    if bool(d['has_coeffs']):
        base._coeffs = d['coeffs_before'].copy()
    s = O.process_samples(paths, base, float(d['gamma']), float(d['lam']), center_adv=True)
    np.testing.assert_allclose(s['returns'], d['s_returns'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(s['advantages'], d['s_advantages'], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(s['observations'], d['s_observations'], atol=1e-13)
    np.testing.assert_allclose(s['agent_infos']['log_std'], d['s_log_std'], atol=1e-13)
    assert abs(s['advantages'].mean()) < 1e-12 and abs(s['advantages'].std() - 1) < 1e-6
    # quirk 6: whole rounds, overshooting batch_size
    assert sum(d['lengths']) >= int(d['batch_size'])


def test_stop_logic():
    d = load_golden('stoplogic')
    stop = O.stop_critereon(0.10, 1e-5, 0.30)
    for o, n, v in zip(d['olds'], d['news'], d['vec']):
        assert bool(stop(o, n, mode='vector')) == bool(v)
    for o, n, v in zip(d['sc_old'], d['sc_new'], d['scal']):
        assert bool(stop(o, n)) == bool(v)
    modes = [str(m) for m in d['modes']]
    for i in range(len(d['done'])):
        mode = modes[int(d['mode_idx'][i])]
        mins = {'real': float(d['min_real'][i]), 'trpo_mean': float(d['min_tm'][i]), 'estimated': d['min_est'][i].copy()}
        cand = {'real': float(d['cand_real'][i]), 'trpo_mean': float(d['cand_tm'][i]), 'estimated': d['cand_est'][i].copy()}
        assert bool(O.policy_is_done(mode, stop, mins, cand)) == bool(d['done'][i])
        for whole, pre in ((False, 'upd0_'), (True, 'upd1_')):
            m2 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in mins.items()}
            O.update_stats(m2, cand, whole)
            assert m2['real'] == d[pre + 'real'][i] and m2['trpo_mean'] == d[pre + 'tm'][i]
            np.testing.assert_array_equal(m2['estimated'], d[pre + 'est'][i])
