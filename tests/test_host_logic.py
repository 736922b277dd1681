"""CPU tests of the host-side logic that mirrors the reference's interface (no GPU, no kernels):
early-stopping state machine vs the reference's own decisions (golden), baseline solve, the
reference-shaped CG loop with an injected evaluator, sharding helpers."""
import numpy as np
import pytest
import torch
from conftest import load_golden
from oracle import metrpo_oracle as O
import metrpo_amd
from metrpo_amd import early_stop
from metrpo_amd.optimizer import ConjugateGradientOptimizer, cg


def test_early_stop_matches_reference_golden():
    d = load_golden('stoplogic')
    stop = early_stop.stop_critereon(0.10, 1e-5, 0.30)
    for o, n, v in zip(d['olds'], d['news'], d['vec']):
        assert bool(stop(o, n, mode='vector')) == bool(v)
    for o, n, v in zip(d['sc_old'], d['sc_new'], d['scal']):
        assert bool(stop(float(o), float(n))) == bool(v)
    modes = [str(m) for m in d['modes']]
    for i in range(len(d['done'])):
        mode = modes[int(d['mode_idx'][i])]
        mins = {'real': float(d['min_real'][i]), 'trpo_mean': float(d['min_tm'][i]), 'estimated': d['min_est'][i].copy()}
        cand = {'real': float(d['cand_real'][i]), 'trpo_mean': float(d['cand_tm'][i]), 'estimated': d['cand_est'][i].copy()}
        assert bool(early_stop.is_done(mode, stop, mins, cand)) == bool(d['done'][i])
        for whole, pre in ((False, 'upd0_'), (True, 'upd1_')):
            m2 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in mins.items()}
            early_stop.update_stats(m2, cand, whole)
            assert m2['real'] == d[pre + 'real'][i] and m2['trpo_mean'] == d[pre + 'tm'][i]
            np.testing.assert_array_equal(m2['estimated'], d[pre + 'est'][i])


def test_baseline_solve_matches_lstsq_fit():
    rng = np.random.RandomState(0)
    paths = []
    for _ in range(20):
        p = dict(observations=rng.randn(15, 10) * 2, rewards=rng.randn(15))
        p['returns'] = O.discount_cumsum(p['rewards'], 0.99)
        paths.append(p)
    ref = O.LinearFeatureBaselineOracle(); ref.fit(paths)
    F = np.concatenate([O.LinearFeatureBaselineOracle.features(p) for p in paths])
    y = np.concatenate([p['returns'] for p in paths])
    b = metrpo_amd.LinearFeatureBaseline()
    assert b.coeffs is None and np.array_equal(b.predict(paths[0]), np.zeros(15))
    got = b.solve(F.T @ F, F.T @ y)
    np.testing.assert_allclose(got, ref._coeffs, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b.predict(paths[1]), ref.predict(paths[1]), rtol=1e-9, atol=1e-12)


class OracleEvaluator(object):
    """Test double for the kernels behind ConjugateGradientOptimizer (lives in tests/ only): this rank's
    pre-scaled share of loss/grad/Hx/loss+kl over its slice of the samples."""

    def __init__(self, theta, pdims, data, n_global):
        self.theta, self.pdims, self.d, self.n = theta.copy(), pdims, data, n_global

    def _w(self):
        return len(self.d[0]) / float(self.n)

    def loss_grad(self):
        l, g = O.surrogate_loss_grad(self.theta, self.pdims, *self.d)
        return torch.from_numpy(np.concatenate([[l], g]) * self._w())

    def hvp(self, v):
        return torch.from_numpy(O.fisher_vector_product(self.theta, self.pdims, self.d[0], np.asarray(v), reg_coeff=0.0) * self._w())

    def loss_constraint(self, theta):
        th = self.theta if theta is None else np.asarray(theta, dtype=np.float64)
        l, k = O.surrogate_loss_kl(th, self.pdims, *self.d)
        return torch.from_numpy(np.array([l, k]) * self._w())

    def get_params(self):
        return self.theta.copy()

    def set_params(self, th):
        self.theta = np.asarray(th, dtype=np.float64)


def make_update_problem(N=300, seed=4):
    dm, theta, pdims, pool = O.make_problem('swimmer', K=2, dyn_hidden=(8, 8), pol_hidden=(8, 8), seed=seed)
    rng = np.random.RandomState(seed)
    theta = (theta + rng.randn(theta.size) * 0.1).astype(np.float32).astype(np.float64)
    obs = rng.randn(N, 10)
    om = O.policy_mean(theta, pdims, obs)
    ols = np.broadcast_to(O.policy_log_std(theta, pdims), om.shape).copy()
    act = om + np.exp(ols) * rng.randn(*om.shape)
    adv = O.center_advantages(rng.randn(N))
    return theta, pdims, (obs, act, adv, om, ols)


def test_host_cg_loop_matches_oracle_optimize():
    theta, pdims, data = make_update_problem()
    ev = OracleEvaluator(theta, pdims, data, len(data[0]))
    opt = ConjugateGradientOptimizer(fused=False)
    opt.update_opt(leq_constraint=(None, 0.01))
    out = opt.optimize(ev)
    ref = O.cg_optimize(theta, pdims, *data, max_kl=0.01)
    np.testing.assert_allclose(out['g'], ref['g'], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out['d'], ref['d'], rtol=1e-9, atol=1e-12)
    assert out['n_backtrack'] == ref['n_backtrack'] and out['accepted'] == ref['accepted']
    np.testing.assert_allclose(ev.get_params(), ref['theta_new'].astype(np.float32), rtol=0, atol=1e-7)
    x = cg(lambda p: 3.0 * p, np.ones(4))
    np.testing.assert_allclose(x, np.ones(4) / 3.0)


def test_comm_shard_single_process():
    c = metrpo_amd.Comm()
    assert (c.rank, c.world) == (0, 1) and c.shard(10) == (0, 10)
    t = torch.ones(3, dtype=torch.float64)
    assert c.allreduce_sum_(t) is t and c.max_float(2.5) == 2.5


def test_paths_and_spaces_surface():
    pool = metrpo_amd.InitStatePool(np.arange(30.0).reshape(3, 10), na=2)
    assert pool.observation_space.shape == (10,) and np.array_equal(pool.action_space.bounds[1], np.ones(2))
    assert np.array_equal(pool.reset(), np.arange(10.0)) and np.array_equal(pool.reset(), np.arange(10.0, 20.0))
    assert pool.observation_space.flatten_n([np.zeros(10), np.ones(10)]).shape == (2, 10)
    with pytest.raises(NotImplementedError):
        metrpo_amd.NPO(env=None, policy=None, baseline=None)


def test_ring_replay_buffer_reproduces_reference_index_stream():
    """Product data_collection (preallocated ring, here on CPU tensors) against what the reference's own class produced:
    the add / get_next_batch / sample log of dyn_data.npz (exact np.random stream) and the reference's OWN buffer tests."""
    from metrpo_amd.dynamics_training import data_collection, combine_data_collections
    from test_oracle_dynamics import run_reference_buffer_tests
    to_np = lambda t: t.detach().cpu().numpy().astype(np.float64)
    d = load_golden('dyn_data')
    dc = data_collection(max_size=50, device='cpu')
    storage = None
    np.random.seed(int(d['seed']))
    ai = 0
    for i, (op, n) in enumerate(zip(d['ops'], d['ns'])):
        if op == 0:
            dc.add_data(d['addx%d' % ai], d['addy%d' % ai]); ai += 1
            xb, yb = dc.x, dc.y
            storage = storage or dc._xs.data_ptr()
            assert dc._xs.data_ptr() == storage and dc._xs.shape[0] == 50      # never re-allocated
        elif op == 1:
            xb, yb = dc.get_next_batch(int(n))
        else:
            xb, yb = dc.sample(int(n))
        np.testing.assert_array_equal(to_np(xb), d['x%d' % i].astype(np.float32))
        np.testing.assert_array_equal(to_np(yb), d['y%d' % i].astype(np.float32))
        assert dc.n_data == int(d['n_data'][i]) and dc.cur_idx == int(d['cur_idx'][i])
    assert dc._head != 0                                                       # the FIFO wrapped physically
    run_reference_buffer_tests(lambda m: data_collection(m, device='cpu'), combine_data_collections, to_np=to_np)
    big = data_collection(max_size=4, device='cpu')                            # one block larger than the ring
    xs = np.arange(14.0).reshape(7, 2)
    big.add_data(xs[:2], xs[:2]); big.add_data(xs, xs)
    np.testing.assert_array_equal(to_np(big.x), xs[3:]); assert big.n_data == 4 and big.cur_idx == 2 - 5


def test_engine_names_the_dynamics_variants_it_does_not_implement():
    """training.py:234-242 (use_logit_weights) and :259-268 (second_derivative, *_goal prediction types) exist in the reference but no six-env params
    file selects them and no kernel here evaluates them: the host mirror refuses them BY NAME before touching the device."""
    import metrpo_amd
    with pytest.raises(ValueError, match='second_derivative'):
        metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32), prediction_type='second_derivative')
    with pytest.raises(ValueError, match='state_change_goal'):
        metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32), prediction_type='state_change_goal')
    with pytest.raises(ValueError, match='use_logit_weights'):
        metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32), use_logit_weights=True)
