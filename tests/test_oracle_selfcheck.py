"""Self-consistency checks for the oracle pieces that NO reference fixture can pin (rllab/TF
arithmetic): gradient, Fisher-vector product, CG, line search, baseline, scan.
Independent derivations: torch CPU autograd (double back-prop = what rllab's PerlmutterHvp
computes), scipy.signal.lfilter, np.linalg.  CPU only."""
import numpy as np
import pytest
import torch
import scipy.signal
from oracle import metrpo_oracle as O


def _problem(N=200, seed=3, env='swimmer'):
    dm, theta, pdims, pool = O.make_problem(env, K=2, dyn_hidden=(8, 8), pol_hidden=(8, 6), seed=seed)
    rng = np.random.RandomState(seed)
    theta = theta + rng.randn(theta.size) * 0.1
    theta[-pdims[-1]:] = rng.randn(pdims[-1]) * 0.3        # log_std
    obs = rng.randn(N, pdims[0])
    old_mean = O.policy_mean(theta, pdims, obs)
    old_ls = np.broadcast_to(O.policy_log_std(theta, pdims), old_mean.shape).copy()
    act = old_mean + np.exp(old_ls) * rng.randn(*old_mean.shape)
    adv = rng.randn(N)
    return theta, pdims, obs, act, adv, old_mean, old_ls


def _torch_loss_kl(theta_t, pdims, obs, act, adv, old_mean, old_ls):
    o, h = 0, torch.from_numpy(obs)
    L = len(pdims) - 1
    for l in range(L):
        W = theta_t[o:o + pdims[l] * pdims[l + 1]].reshape(pdims[l], pdims[l + 1]); o += W.numel()
        b = theta_t[o:o + pdims[l + 1]]; o += pdims[l + 1]
        h = h @ W + b
        if l < L - 1:
            h = torch.tanh(h)
    ls = torch.clamp(theta_t[o:], min=float(O.LOG_MIN_STD))
    mean, std = h, torch.exp(ls)
    om, ols = torch.from_numpy(old_mean), torch.from_numpy(old_ls)
    ostd = torch.exp(ols)
    na = pdims[-1]
    ll_new = -ls.sum() - 0.5 * (((torch.from_numpy(act) - mean) / std) ** 2).sum(-1) - 0.5 * na * np.log(2 * np.pi)
    ll_old = -ols.sum(-1) - 0.5 * (((torch.from_numpy(act) - om) / ostd) ** 2).sum(-1) - 0.5 * na * np.log(2 * np.pi)
    lr = torch.exp(ll_new - ll_old)
    kl = (((om - mean) ** 2 + ostd ** 2 - std ** 2) / (2 * std ** 2 + 1e-8) + ls - ols).sum(-1)
    return -(lr * torch.from_numpy(adv)).mean(), kl.mean()


def test_loss_kl_grad_match_autograd():
    theta, pdims, obs, act, adv, om, ols = _problem()
    # evaluate away from theta_old so lr != 1 and kl != 0
    th2 = theta + np.random.RandomState(0).randn(theta.size) * 0.05
    t = torch.tensor(th2, requires_grad=True)
    loss_t, kl_t = _torch_loss_kl(t, pdims, obs, act, adv, om, ols)
    loss, kl = O.surrogate_loss_kl(th2, pdims, obs, act, adv, om, ols)
    assert abs(loss - loss_t.item()) < 1e-12 and abs(kl - kl_t.item()) < 1e-12 and kl > 1e-4
    g_t, = torch.autograd.grad(loss_t, t)
    loss2, g = O.surrogate_loss_grad(th2, pdims, obs, act, adv, om, ols)
    assert abs(loss2 - loss) < 1e-14
    np.testing.assert_allclose(g, g_t.numpy(), rtol=1e-9, atol=1e-12)


def test_fvp_matches_double_backprop():
    theta, pdims, obs, act, adv, om, ols = _problem()
    v = np.random.RandomState(1).randn(theta.size)
    t = torch.tensor(theta, requires_grad=True)
    _, kl_t = _torch_loss_kl(t, pdims, obs, act, adv, om, ols)
    gk, = torch.autograd.grad(kl_t, t, create_graph=True)
    hv, = torch.autograd.grad((gk * torch.from_numpy(v)).sum(), t)
    ref = hv.numpy() + 1e-5 * v
    got = O.fisher_vector_product(theta, pdims, obs, v, reg_coeff=1e-5)
    np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-10)


def test_fvp_small_std_eps_term():
    # with std ~ 1e-3 the 1e-8 in the KL denominator is no longer negligible; the exact form must still match
    theta, pdims, obs, act, adv, om, ols = _problem(seed=5)
    theta[-pdims[-1]:] = np.log(1e-3)
    om = O.policy_mean(theta, pdims, obs)
    ols = np.broadcast_to(O.policy_log_std(theta, pdims), om.shape).copy()
    v = np.random.RandomState(2).randn(theta.size)
    t = torch.tensor(theta, requires_grad=True)
    _, kl_t = _torch_loss_kl(t, pdims, obs, act, adv, om, ols)
    gk, = torch.autograd.grad(kl_t, t, create_graph=True)
    hv, = torch.autograd.grad((gk * torch.from_numpy(v)).sum(), t)
    got = O.fisher_vector_product(theta, pdims, obs, v, reg_coeff=0.0)
    np.testing.assert_allclose(got, hv.numpy(), rtol=1e-6, atol=1e-6 * np.abs(hv.numpy()).max())


def test_cg_solves_and_trpo_invariants():
    theta, pdims, obs, act, adv, om, ols = _problem(N=400)
    adv = O.center_advantages(adv)
    out = O.cg_optimize(theta, pdims, obs, act, adv, om, ols, max_kl=0.01)
    Hx = lambda x: O.fisher_vector_product(theta, pdims, obs, x, 1e-5)
    # 10 CG iterations reduce the residual substantially on a P=~150 problem
    r = Hx(out['d']) - out['g']
    assert np.linalg.norm(r) < 0.5 * np.linalg.norm(out['g'])
    assert out['accepted']
    assert out['loss'] < out['loss_before'] and out['kl'] <= 0.01
    step = theta - out['theta_new']
    np.testing.assert_allclose(step, (0.8 ** out['n_backtrack']) * out['beta'] * out['d'], rtol=1e-12, atol=1e-15)
    # quadratic model: 0.5 * step^T H step == max_kl at ratio 1
    full = out['beta'] * out['d']
    assert abs(0.5 * full.dot(Hx(full)) - 0.01) < 1e-6


def test_cg_exact_on_spd():
    rng = np.random.RandomState(0)
    A = rng.randn(6, 6); A = A @ A.T + 6 * np.eye(6)
    b = rng.randn(6)
    x = O.cg(lambda p: A @ p, b, cg_iters=10)
    np.testing.assert_allclose(A @ x, b, atol=1e-8)


def test_rejected_step_restores_theta():
    theta, pdims, obs, act, adv, om, ols = _problem()
    out = O.cg_optimize(theta, pdims, obs, act, adv * 0.0, om, ols)   # zero advantages: g = 0 -> nan step
    np.testing.assert_array_equal(out['theta_new'], theta)
    assert not out['accepted']


@pytest.mark.parametrize('c', [1.0, 0.99, 0.95 * 0.99, 0.0])
def test_discount_cumsum_is_lfilter(c):
    x = np.random.RandomState(0).randn(37)
    ref = scipy.signal.lfilter([1], [1, float(-c)], x[::-1], axis=0)[::-1]
    np.testing.assert_allclose(O.discount_cumsum(x, c), ref, rtol=1e-12, atol=1e-12)


def test_linear_baseline_fit_predict():
    rng = np.random.RandomState(0)
    paths = []
    w = rng.randn(24) * 0.1
    for _ in range(30):
        p = dict(observations=rng.randn(20, 10) * 3, rewards=np.zeros(20))
        p['returns'] = O.LinearFeatureBaselineOracle.features(p) @ w
        paths.append(p)
    b = O.LinearFeatureBaselineOracle()
    assert np.array_equal(b.predict(paths[0]), np.zeros(20))        # zeros before the first fit
    b.fit(paths)
    np.testing.assert_allclose(b.predict(paths[3]), paths[3]['returns'], atol=1e-4)
    f = O.LinearFeatureBaselineOracle.features(dict(observations=np.full((3, 2), 20.0), rewards=np.zeros(3)))
    assert f.shape == (3, 8) and f[0, 0] == 10.0 and f[2, 4] == 0.02 and f[2, 7] == 1.0   # clip, t/100, bias


def test_validation_cost_equals_stepwise_rollout():
    """Known-answer relation of env_helpers.py:271-305 (test_policy_cost): the unrolled-graph cost
    equals a step-by-step rollout with cost_np_vec, here for eps_rand-with-fixed-model VecEnv."""
    env = 'swimmer'
    dm, theta, pdims, pool = O.make_problem(env, K=3, dyn_hidden=(8, 8), pol_hidden=(8, 8), seed=11)
    s0, T, gamma = pool[:16], 7, 0.97
    costs = O.validation_costs(dm, theta, pdims, env, s0, T, gamma)
    for k in range(dm.K):
        x, tot = s0.copy(), 0.0
        for t in range(T):
            u = np.clip(O.policy_mean(theta, pdims, x), -1, 1)
            xn = O.select_next(O.dynamics_forward_all(dm, x, u), 'eps_rand', np.full(len(x), k))
            tot += gamma ** t * np.mean(O.cost_np_vec(env, x, u, xn))
            x = xn
        assert abs(tot - costs[k]) < 1e-12


def test_rms_floor():
    m, s = O.rms_mean_std(np.array([2.0, 0.0]), np.array([4.0, 0.0]) + 1e-2, 1.0 + 1e-2)
    assert s[1] == pytest.approx(0.1) and s[0] == pytest.approx(np.sqrt(max(4.01 / 1.01 - (2 / 1.01) ** 2, 1e-2)))
