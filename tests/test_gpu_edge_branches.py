"""Reference edge branches of the update half that have kernel code of their own, each through the C ABI against the oracle:

  * `log_std` below log(1e-6)         [rllab] GaussianMLPPolicy(min_std=1e-6): the clamp, its zeroed gradient / Fisher entries   oracle/metrpo_oracle.py: surrogate_loss_grad, fisher_vector_product
  * krylov.cg's `residual_tol` exit   [rllab] krylov.cg:  if rdotr < residual_tol: break                                         cg, cg_optimize
  * accept_violation=True             [rllab] ConjugateGradientOptimizer.optimize: keep the last trial although it violates      cg_optimize
  * positive_adv=True                 algos/batch_polopt.py:33, samplers/base.py:85-86: shift_advantages_to_positive              process_samples
  * the x10 regulariser retry         [rllab] LinearFeatureBaseline.fit: reg_coeff *= 10 while the solve is not finite            LinearFeatureBaselineOracle.fit
"""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
import helpers as Hh
import tolerances as TOL
from test_gpu_engine import _update_problem, rel_l2

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _clamped_problem(env, pol_hidden, N, seed=41):
    """An update problem whose policy has log_std[0] = -15, BELOW log(1e-6); one other dim at -0.7, the rest free.  Actions are drawn from the clamped
    distribution (sigma_0 = 1e-6), so z = (a - mean) / sigma stays O(1) -- which needs a - mean to 1e-6 * 1e-7: the output unit of action dim 0 is
    given zero weights and bias, its mean is EXACTLY 0 in fp32 and fp64 alike (with a generic mean the fp32 rounding of the mean, 6e-8, would be 6 %
    of sigma and no fp32 / fp64 comparison of the ratio would mean anything)."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 2, (64, 64), pol_hidden, seed=seed)
    na = dm.na
    Ws, bs, ls = O.policy_unflatten(theta.astype(np.float32).astype(np.float64), pdims)
    Ws = [w.copy() for w in Ws]; bs = [b.copy() for b in bs]; ls = ls.copy()
    Ws[-1][:, 0] = 0.0; bs[-1][0] = 0.0
    ls[0] = -15.0
    ls[1] = -0.7
    th = O.policy_flatten(Ws, bs, ls).astype(np.float32).astype(np.float64)
    eng.set_policy(th.astype(np.float32))
    rng = np.random.RandomState(seed)
    obs = (rng.randn(N, dm.ns) * 0.5).astype(np.float32).astype(np.float64)
    old_mean = O.policy_mean(th, pdims, obs).astype(np.float32).astype(np.float64)
    assert (old_mean[:, 0] == 0.0).all()
    lsc = O.policy_log_std(th, pdims)
    assert lsc[0] == O.LOG_MIN_STD and th[-na] < O.LOG_MIN_STD
    old_ls = np.broadcast_to(lsc, old_mean.shape).copy()
    act = (old_mean + np.exp(old_ls) * rng.randn(*old_mean.shape)).astype(np.float32).astype(np.float64)
    adv = O.center_advantages(rng.randn(N)).astype(np.float32).astype(np.float64)
    return eng, th, pdims, obs, act, adv, old_mean, old_ls


@pytest.mark.parametrize('use_mfma', [True, False, 'gemm'])
@pytest.mark.parametrize('env,pol_hidden,N', [('swimmer', (32, 32), 3000), ('half_cheetah', (32, 32), 2111), ('humanoid', (100, 50, 25), 1500)])
def test_log_std_below_the_clamp_on_every_update_path(env, pol_hidden, N, use_mfma):
    """sigma = exp(max(log_std, log 1e-6)): KL and the likelihood ratio use the clamped sigma; the log_std entry of the gradient and of the
    Fisher-vector product is ZERO where the raw parameter sits below the bound (the max() has no slope there) and non-zero elsewhere."""
    eng, th, pdims, obs, act, adv, om, ols = _clamped_problem(env, pol_hidden, N)
    active = eng.set_update_path(use_mfma)
    if use_mfma is True and not active:
        pytest.skip('no fused MFMA update kernels for this policy shape (generic / GEMM paths cover it)')
    na, P = pdims[-1], th.size
    batch = eng.make_batch(obs, act, adv, om, ols)
    out = cpu(eng.loss_grad(batch))
    loss, g = O.surrogate_loss_grad(th, pdims, obs, act, adv, om, ols)
    assert np.isfinite(out).all() and np.isfinite(g).all()
    assert out[1 + P - na] == 0.0 and g[P - na] == 0.0                               # clamped dim: exactly zero, both sides
    assert abs(g[P - na + 1]) > 0 and abs(out[1 + P - na + 1]) > 0                   # a free dim keeps its slope
    assert abs(out[0] - loss) <= TOL.LOSS_RTOL * max(1.0, abs(loss))
    # the clamped action dim's output weights see z / sigma = O(1e6) per sample: the gradient spans 6 decades.  Whole vector by rel-L2 (dominated by
    # those entries), the free log_std entries on their own
    assert rel_l2(out[1:], g) <= 10 * TOL.GRAD_REL_L2
    np.testing.assert_allclose(out[1 + P - na + 1:], g[P - na + 1:], rtol=1e-4, atol=1e-7)
    v = np.random.RandomState(1).randn(eng.P)
    hv = cpu(eng.fvp(batch, v))
    ref = O.fisher_vector_product(th, pdims, obs, v, reg_coeff=0.0)
    assert hv[P - na] == 0.0 and ref[P - na] == 0.0 and abs(hv[P - na + 1]) > 0
    assert rel_l2(hv, ref) <= 10 * TOL.FVP_REL_L2                                    # 1 / (sigma^2 + eps / 2) = 2e8 on the clamped column: same arithmetic, larger dynamic range
    np.testing.assert_allclose(hv[P - na + 1:], ref[P - na + 1:], rtol=1e-5, atol=1e-9)
    # KL / ratio at a trial theta: every parameter moves except the clamped unit's (its mean stays exactly 0); the clamped log_std is pushed further
    # down -- sigma stays 1e-6 on both sides, so that dim adds 0 to the KL (with sigma = exp(-20) it would add 1e5)
    Ws, bs, ls = O.policy_unflatten(th, pdims)
    rng = np.random.RandomState(2)
    Ws2 = [w + rng.randn(*w.shape) * 0.02 for w in Ws]; bs2 = [b + rng.randn(*b.shape) * 0.02 for b in bs]
    Ws2[-1][:, 0] = 0.0; bs2[-1][0] = 0.0
    ls2 = ls + rng.randn(na) * 0.02; ls2[0] = -20.0
    th2 = O.policy_flatten(Ws2, bs2, ls2).astype(np.float32).astype(np.float64)
    lk = cpu(eng.loss_kl(batch, th2.astype(np.float32)))
    l2, k2 = O.surrogate_loss_kl(th2, pdims, obs, act, adv, om, ols)
    assert np.isfinite(lk).all() and 0 < k2 < 1.0
    assert abs(lk[0] - l2) <= TOL.LOSS_RTOL * max(1.0, abs(l2)) and abs(lk[1] - k2) <= max(TOL.KL_ATOL, TOL.KL_RTOL * k2)
    lk0 = cpu(eng.loss_kl(batch))
    assert abs(lk0[1]) < TOL.KL_ATOL and abs(lk0[0] - loss) < 1e-6                   # at theta_old: KL = 0 with the clamped sigma (it would be < 0 with exp(-15))


@pytest.mark.parametrize('use_mfma', [True, False, 'gemm'])
def test_cg_leaves_early_at_residual_tol_device_and_host_agree(use_mfma):
    """krylov.cg breaks once r.r < residual_tol.  The tolerance is placed (geometric mean) in the widest drop of the oracle's r.r sequence -- between
    an iterate and the smallest one before it, at least a factor 2 apart, so fp32 rounding cannot move the exit -- and the device-fused solve, the
    host-driven solve (all-reduce callback: k_cg_step kernels) and the oracle must agree on the iteration count and on the direction."""
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=6000, seed=27)
    assert eng.set_update_path(use_mfma) == use_mfma
    full = O.cg_optimize(th, pdims, obs, act, adv, om, ols, max_kl=0.01)
    rr = full['rdotr']
    assert len(rr) == 10
    cand = [(min(rr[:i]) / rr[i], i) for i in range(1, 9) if rr[i] < 0.5 * min(rr[:i])]      # (CG's Euclidean residual is not monotone)
    assert cand, rr
    i_exit = max(cand)[1]
    tol = float(np.sqrt(rr[i_exit] * min(rr[:i_exit])))
    n_run = i_exit + 1
    ref = O.cg_optimize(th, pdims, obs, act, adv, om, ols, max_kl=0.01, residual_tol=tol)
    assert ref['cg_iters_run'] == n_run < 10
    theta0 = eng.get_policy().clone()
    batch = eng.make_batch(obs, act, adv, om, ols)
    dev = eng.trpo_update(batch, max_kl=0.01, residual_tol=tol, want_vectors=True)
    eng.set_policy(theta0)
    host = eng.trpo_update(batch, max_kl=0.01, residual_tol=tol, want_vectors=True, allreduce=lambda t: t)     # single rank: the sum is the identity
    assert dev['cg_iters_run'] == host['cg_iters_run'] == n_run
    for out in (dev, host):
        d = cpu(out['d'])
        assert rel_l2(d, ref['d']) <= TOL.CG_REL_L2
        assert abs(out['beta'] - ref['beta']) <= TOL.STEP_SCALE_RTOL * ref['beta']
        assert out['accepted'] == ref['accepted'] and out['n_backtrack'] == ref['n_backtrack']
    assert rel_l2(cpu(dev['d']), cpu(host['d'])) <= 1e-6
    assert rel_l2(cpu(dev['d']), full['d']) > 10 * TOL.CG_REL_L2                    # and it IS a different direction from the 10-iteration one


@pytest.mark.parametrize('use_mfma', [True, False, 'gemm'])
def test_accept_violation_keeps_the_last_trial(use_mfma):
    """With too few backtracks for the constraint (max_backtracks = 2 at a max_kl the quadratic model overshoots) the search ends on a violating
    trial: accept_violation=False restores theta_old, accept_violation=True keeps that last trial's theta -- [rllab] optimize()."""
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=4000, seed=29)
    assert eng.set_update_path(use_mfma) == use_mfma
    pick = None
    for max_kl in (2.0, 5.0, 10.0, 20.0):
        r0 = O.cg_optimize(th, pdims, obs, act, adv, om, ols, max_kl=max_kl, max_backtracks=2)
        margin = min(abs(r0['kl'] - max_kl) / max_kl, 1.0) if np.isfinite(r0['kl']) else 0.0
        if not r0['accepted'] and np.isfinite(r0['loss']) and np.isfinite(r0['kl']) and (r0['kl'] > 1.05 * max_kl or r0['loss'] > r0['loss_before'] + 1e-3) and margin > 0.05:
            pick = max_kl
            break
    assert pick is not None, "no violating configuration found"
    ref = O.cg_optimize(th, pdims, obs, act, adv, om, ols, max_kl=pick, max_backtracks=2, accept_violation=True)
    assert ref['accepted'] and ref['n_backtrack'] == 1 and not np.allclose(ref['theta_new'], th)
    theta0 = eng.get_policy().clone()
    batch = eng.make_batch(obs, act, adv, om, ols)
    rej = eng.trpo_update(batch, max_kl=pick, max_backtracks=2, accept_violation=False)
    assert not rej['accepted'] and torch.equal(eng.get_policy(), theta0)
    out = eng.trpo_update(batch, max_kl=pick, max_backtracks=2, accept_violation=True)
    assert out['accepted'] and out['n_backtrack'] == 1
    step_ref = ref['theta_new'] - th
    assert rel_l2(cpu(eng.get_policy()) - th.astype(np.float32).astype(np.float64), step_ref) <= TOL.THETA_STEP_REL_L2
    assert abs(out['kl'] - ref['kl']) <= TOL.POST_UPDATE_RTOL * abs(ref['kl']) and abs(out['loss'] - ref['loss']) <= TOL.POST_UPDATE_RTOL * max(1.0, abs(ref['loss']))
    # the asynchronous form (line search decided on the device) takes the same branch
    eng.set_policy(theta0)
    assert eng.trpo_update(batch, max_kl=pick, max_backtracks=2, accept_violation=True, spec_trials=2) is None
    end = eng.trpo_update_end()
    assert end['accepted'] and end['n_backtrack'] == 1
    assert rel_l2(cpu(eng.get_policy()) - th.astype(np.float32).astype(np.float64), step_ref) <= TOL.THETA_STEP_REL_L2


@pytest.mark.parametrize('env,batch', [('swimmer', 640), ('ant', 900)])
def test_positive_adv_through_process_samples(env, batch):
    """BatchPolopt(positive_adv=True): after centring, the advantages are shifted to be positive (minimum over the VALID samples becomes 1e-8)."""
    from test_gpu_api import build_algo
    B, H = 64, 10
    algo, eng, dm, theta, pdims, pool = build_algo(env, B=B, H=H, batch=batch)
    algo.positive_adv = True
    if env == 'ant':
        pool[::3, 2] = 0.21
        algo.env.env.states[::3, 2] = 0.21; algo.env.env._dev = None
    prev = np.random.RandomState(3).randn(2 * dm.ns + 4) * 0.05
    algo.baseline.set_param_values(prev.copy())
    algo.start_worker()
    paths = algo.obtain_samples(0)
    plist = paths.to_paths()
    samples = algo.process_samples(0, paths)
    base = O.LinearFeatureBaselineOracle(); base._coeffs = prev.copy()
    ref = O.process_samples([dict(p) for p in plist], base, algo.discount, algo.gae_lambda, center_adv=True, positive_adv=True)
    v = cpu(samples['valids']).astype(bool)
    got = np.sort(cpu(samples['advantages'])[v])
    want = np.sort(ref['advantages'])
    assert got.size == want.size and got.min() > 0 and abs(got.min() - 1e-8) < 1e-9 and want.min() == 1e-8
    np.testing.assert_allclose(got, want, **TOL.ADVANTAGE_CENTRED)
    if env == 'ant':
        assert (~v).any()                                                           # the minimum was taken over valid samples only (dropped tails hold 0)


@pytest.mark.parametrize('env', ['swimmer', 'humanoid'])
def test_baseline_solve_regulariser_retry(env):
    """LinearFeatureBaseline.fit escalates reg_coeff x10 (up to 5 times) while the solve is not finite.  The device solves by elimination (no lstsq), so an
    EXACTLY singular A + reg I (pivot -reg + reg = 0) gives a non-finite first attempt and the second attempt solves (A + 10 reg I) x = b; non-finite
    input stays non-finite after all five attempts, as the oracle's coefficients do."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 2, (64, 64), (32, 32) if env != 'humanoid' else (100, 50, 25), seed=3)
    F = 2 * dm.ns + 4                                                               # 24: the one-wave solver; 114: the workgroup solver
    rng = np.random.RandomState(5)
    M = rng.randn(4 * F, F)
    A = M.T @ M / (4 * F)
    b = rng.randn(F)
    reg = 1e-5
    A[0, :] = 0.0; A[:, 0] = 0.0; A[0, 0] = -reg                                    # pivot 0 of A + reg I is exactly 0.0
    gram = torch.tensor(np.concatenate([A.reshape(-1), b]), dtype=torch.float64, device=eng.device)
    got = cpu(eng.baseline_solve(gram, reg_coeff=reg))
    assert np.isfinite(got).all()
    want = np.linalg.solve(A + 10 * reg * np.eye(F), b)                             # the second attempt's system
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-9 * np.abs(want).max())
    # a regular system: one attempt, the oracle's lstsq solution of (A + reg I) x = b
    A2 = M.T @ M / (4 * F)
    gram2 = torch.tensor(np.concatenate([A2.reshape(-1), b]), dtype=torch.float64, device=eng.device)
    want2 = np.linalg.lstsq(A2 + reg * np.eye(F), b, rcond=None)[0]
    np.testing.assert_allclose(cpu(eng.baseline_solve(gram2, reg_coeff=reg)), want2, rtol=1e-7, atol=1e-9 * np.abs(want2).max())
    # non-finite normal equations: five attempts, still non-finite -- the oracle's fit ends the same way
    A3 = A2.copy(); A3[1, 1] = np.nan
    gram3 = torch.tensor(np.concatenate([A3.reshape(-1), b]), dtype=torch.float64, device=eng.device)
    assert not np.isfinite(cpu(eng.baseline_solve(gram3, reg_coeff=reg))).all()


@pytest.mark.parametrize('env', ['swimmer', 'humanoid'])
def test_baseline_solve_rank_deficient_features_match_lstsq(env):
    """A constant observation column (a clipped, saturated state dim) makes the feature matrix rank deficient: obs_j = c * ones and obs_j^2 = c^2 * ones are
    collinear with the constant feature.  F^T F + reg I is then positive definite only through reg (condition ~ 1e12).  The reference solves it with the
    SVD-based np.linalg.lstsq (linear_feature_baseline.py [rllab]); the device's elimination in the natural order must return the same FIT -- the
    predictions F x, which is all the next GAE reads -- and bounded coefficients, not large finite numbers out of a broken pivot."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 2, (64, 64), (32, 32) if env != 'humanoid' else (100, 50, 25), seed=3)
    ns = dm.ns
    rng = np.random.RandomState(11)
    n = 4000
    obs = np.clip(rng.randn(n, ns), -10, 10)
    obs[:, 1] = 0.75                                                                # the saturated dim
    obs[:, ns - 1] = obs[:, 0]                                                      # and an exactly duplicated one
    al = (np.arange(n) % 100).reshape(-1, 1) / 100.0
    Fm = np.concatenate([obs, obs ** 2, al, al ** 2, al ** 3, np.ones((n, 1))], axis=1)      # LinearFeatureBaseline._features
    y = rng.randn(n) + Fm[:, 0]
    A, b, reg = Fm.T @ Fm, Fm.T @ y, 1e-5
    gram = torch.tensor(np.concatenate([A.reshape(-1), b]), dtype=torch.float64, device=eng.device)
    got = cpu(eng.baseline_solve(gram, reg_coeff=reg))
    want = np.linalg.lstsq(A + reg * np.eye(Fm.shape[1]), b, rcond=None)[0]
    assert np.isfinite(got).all()
    assert np.abs(got).max() <= 10.0 * max(1.0, np.abs(want).max())                 # bounded like the minimum-norm solution
    np.testing.assert_allclose(Fm @ got, Fm @ want, rtol=0, atol=1e-6 * np.abs(y).max())      # the same fit
