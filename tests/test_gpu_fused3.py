"""GPU parity of the fused three-hidden-layer update kernels (csrc/policy_fused3.hip: Humanoid's 100-50-25 policy, params-humanoid.json:5-9) against the
float64 oracle (algos/npo.py:68-111 graph; [rllab] DiagonalGaussian / PerlmutterHvp / krylov.cg) and against the GEMM path they replace.
Tolerances: tests/tolerances.py = DESIGN.md section 5."""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
import tolerances as TOL
from test_gpu_engine import _update_problem, rel_l2, cpu

pytestmark = pytest.mark.gpu
POL = (100, 50, 25)


def _problem(N, seed=21):
    eng, th, pdims, obs, act, adv, om, ols = _update_problem('humanoid', N, seed=seed, pol_hidden=POL)
    assert eng.set_update_path(True) is True and eng.update_path(N) == 'mfma', 'fused kernels must serve the 100-50-25 policy'
    return eng, th, pdims, obs, act, adv, om, ols


@pytest.mark.parametrize('N', [16, 1500, 20011, 131072 + 5])
def test_fused3_grad_fvp_losskl_vs_oracle_and_gemm_path(N):
    """Every tile count class: one tile, fewer tiles than waves, a ragged last tile, several tiles per wave; with a valid mask."""
    eng, th, pdims, obs, act, adv, om, ols = _problem(N)
    valid = np.ones(N, np.uint8); valid[::7] = 0
    if N == 16:
        valid[:] = 1; valid[3] = 0
    keep = valid.astype(bool)
    batch = eng.make_batch(obs, act, adv, om, ols, valid=valid)
    out = cpu(eng.loss_grad(batch))
    loss, g = O.surrogate_loss_grad(th, pdims, obs[keep], act[keep], adv[keep], om[keep], ols[keep])
    assert abs(out[0] - loss) <= TOL.LOSS_RTOL * max(1.0, abs(loss))
    assert rel_l2(out[1:], g) <= TOL.GRAD_REL_L2
    v = np.random.RandomState(1).randn(eng.P)
    hv = cpu(eng.fvp(batch, v))
    ref = O.fisher_vector_product(th, pdims, obs[keep], v, reg_coeff=0.0)
    assert rel_l2(hv, ref) <= TOL.FVP_REL_L2
    th2 = (th + np.random.RandomState(2).randn(th.size) * 0.01).astype(np.float32)
    batch0 = eng.make_batch(obs, act, adv, om, ols[0], valid=valid)
    l2, k2 = O.surrogate_loss_kl(th2.astype(np.float64), pdims, obs[keep], act[keep], adv[keep], om[keep], ols[keep])
    for bt in (batch, batch0):
        lk = cpu(eng.loss_kl(bt, th2))
        assert abs(lk[0] - l2) <= TOL.LOSS_RTOL * max(1.0, abs(l2)) and abs(lk[1] - k2) <= max(TOL.KL_ATOL, TOL.KL_RTOL * k2)
    # the GEMM path on the same inputs (different summation orders: float32 rounding apart)
    assert eng.set_update_path('gemm') == 'gemm'
    out_g = cpu(eng.loss_grad(batch)); hv_g = cpu(eng.fvp(batch, v))
    assert rel_l2(out[1:], out_g[1:]) <= 2 * TOL.GRAD_REL_L2 and rel_l2(hv, hv_g) <= 2 * TOL.FVP_REL_L2
    # bitwise reproducible (fixed reduction orders)
    assert eng.set_update_path(True) is True
    assert torch.equal(eng.loss_grad(batch).clone(), eng.loss_grad(batch).clone())
    assert torch.equal(eng.fvp(batch, v).clone(), eng.fvp(batch, v).clone())


def test_fused3_option_switches_back_to_the_gemm_path():
    eng, th, pdims, obs, act, adv, om, ols = _problem(4000)
    eng.set_option('NO_POL_FUSED3', 1)
    assert eng.update_path(4000) == 'gemm'
    eng.set_option('NO_POL_FUSED3', None)
    assert eng.update_path(4000) == 'mfma'


@pytest.mark.parametrize('seed', [21, 22])
def test_fused3_trpo_update_vs_oracle(seed):
    """Whole update (device-fused CG tails, device-decided line search -- which the GEMM path does not have) against rllab's algorithm."""
    eng, th, pdims, obs, act, adv, om, ols = _problem(6000, seed=seed)
    batch = eng.make_batch(obs, act, adv, om, ols)
    out = eng.trpo_update(batch, max_kl=0.01, want_vectors=True)
    ref = O.cg_optimize(th, pdims, obs, act, adv, om, ols, max_kl=0.01)
    assert rel_l2(cpu(out['g']), ref['g']) <= TOL.GRAD_REL_L2
    d, dref = cpu(out['d']), ref['d']
    cos = d.dot(dref) / (np.linalg.norm(d) * np.linalg.norm(dref))
    assert cos >= TOL.CG_COS and rel_l2(d, dref) <= TOL.CG_REL_L2
    assert abs(out['beta'] - ref['beta']) <= TOL.STEP_SCALE_RTOL * ref['beta']
    assert out['accepted'] == ref['accepted'] and out['n_backtrack'] == ref['n_backtrack']
    assert out['cg_iters_run'] == 10
    assert abs(out['kl'] - ref['kl']) <= TOL.POST_UPDATE_RTOL * ref['kl'] and out['kl'] <= 0.01 and out['loss'] < out['loss_before']
    step_ref = ref['theta_new'] - th
    assert rel_l2(cpu(eng.get_policy()) - th.astype(np.float32).astype(np.float64), step_ref) <= TOL.THETA_STEP_REL_L2


@pytest.mark.parametrize('B,T,H,draws', [(100, 7, 4, False), (37, 6, 3, True), (500, 5, 5, False), (16, 4, 2, False)])
def test_humanoid_split_prestep_is_bitwise_the_wave_per_tile_prestep(B, T, H, draws):
    """rollout_gemm.hip k_big_pre_mfma3_split (a 16-env tile per workgroup, the layers' column blocks over its four waves: the params-file batches) against
    k_big_pre_mfma3 (a wave per tile; option NO_PRE_SPLIT): same fragment image, same k order per unit, same draws -- every trajectory tensor bit for bit.
    Merged rounds (T > H: tiles spanning two rounds), ragged last tile, supplied and production draws."""
    import helpers as Hh
    K = 3
    eng, dm, theta, pdims, pool = Hh.make_engine('humanoid', K, (128, 128), POL, seed=23)
    assert eng.set_rollout_variant(1) == 3
    kw = {}
    if draws:
        dr = Hh.draws(np.random.RandomState(5), K, B, T, dm.ns, dm.na, len(pool))
        kw = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    a = eng.rollout(B, T, H, 'step_rand', pool, seed=3, **kw)
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    a = [x.clone() for x in (a.obs, a.act, a.mean, a.rew, a.done, a.last_obs)]
    eng.set_option('NO_PRE_SPLIT', 1)
    b = eng.rollout(B, T, H, 'step_rand', pool, seed=3, **kw)
    for x, y in zip(a, (b.obs, b.act, b.mean, b.rew, b.done, b.last_obs)):
        assert torch.equal(x, y)


def test_fused3_host_driven_solve_agrees_with_the_device_fused_one():
    """An all-reduce callback between every reduction and its consumer (the multi-rank route without a peer-mapped transport: stand-alone k_cg_step kernels, host-side
    line search) on the fused kernels: same CG trajectory as the single-launch-sequence update up to the float64 rounding of the two reduction routes."""
    eng, th, pdims, obs, act, adv, om, ols = _problem(9000, seed=25)
    batch = eng.make_batch(obs, act, adv, om, ols)
    theta0 = eng.get_policy().clone()
    dev = eng.trpo_update(batch, max_kl=0.01, want_vectors=True)
    td = eng.get_policy().clone()
    eng.set_policy(theta0)
    host = eng.trpo_update(batch, max_kl=0.01, want_vectors=True, allreduce=lambda t: t)      # single rank: the sum is the identity
    assert torch.equal(dev['g'], host['g'])
    assert rel_l2(cpu(dev['d']), cpu(host['d'])) <= 1e-9 and abs(dev['beta'] - host['beta']) <= 1e-9 * host['beta']
    assert dev['n_backtrack'] == host['n_backtrack'] and dev['accepted'] == host['accepted'] and dev['cg_iters_run'] == host['cg_iters_run'] == 10
    np.testing.assert_allclose(cpu(td), cpu(eng.get_policy()), rtol=0, atol=1e-7)
