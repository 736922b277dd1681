"""The persistent CG solve (csrc/policy_mfma.hip MODE_CGP: all Fisher-vector products of [rllab] krylov.cg and the vector steps between them in ONE
launch, grid barriers in between; opt-in through option CG_PERSIST: it measured slower than the launch-per-product solve, profiles/r05_update_levers.txt)
against that solve -- bit for bit: same tile deal, same partial rows, the column sums in
k_finalize's order, the CG step on the same (virtual) 1024 threads -- and against the oracle."""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
import tolerances as TOL
from test_gpu_engine import _update_problem, rel_l2, cpu

pytestmark = pytest.mark.gpu


def _run(eng, batch, theta0, persist, **kw):
    eng.set_policy(theta0)
    eng.set_option('CG_PERSIST', 1 if persist else None)
    n0 = eng.cg_persist_launches()
    out = eng.trpo_update(batch, max_kl=0.01, want_vectors=True, **kw)
    took = eng.cg_persist_launches() - n0
    eng.set_option('CG_PERSIST', None)
    return out, eng.get_policy().clone(), took


@pytest.mark.parametrize('env,N', [('swimmer', 6000), ('swimmer', 16), ('swimmer', 200003), ('half_cheetah', 30011), ('ant', 50000), ('hopper', 4097), ('snake', 9999)])
def test_persistent_cg_solve_is_bitwise_the_per_launch_solve(env, N):
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(env, N)
    assert eng.set_update_path(True) is True
    valid = np.ones(N, np.uint8); valid[::5] = 0 if N > 16 else 1
    batch = eng.make_batch(obs, act, adv, om, ols, valid=valid)
    theta0 = eng.get_policy().clone()
    a, ta, took_a = _run(eng, batch, theta0, True)
    b, tb, took_b = _run(eng, batch, theta0, False)
    assert took_a == 1 and took_b == 0, 'option CG_PERSIST must select the one-launch solve on an exclusive device'
    assert torch.equal(a['g'], b['g']) and torch.equal(a['d'], b['d']) and torch.equal(ta, tb)
    assert a['beta'] == b['beta'] and a['cg_iters_run'] == b['cg_iters_run'] == 10
    assert a['n_backtrack'] == b['n_backtrack'] and a['accepted'] == b['accepted'] and a['kl'] == b['kl'] and a['loss'] == b['loss']


def test_persistent_cg_solve_vs_oracle_and_explicit_final_hvp():
    eng, th, pdims, obs, act, adv, om, ols = _update_problem('swimmer', 20000, seed=23)
    batch = eng.make_batch(obs, act, adv, om, ols)
    theta0 = eng.get_policy().clone()
    a, ta, took = _run(eng, batch, theta0, True)
    assert took == 1
    ref = O.cg_optimize(th, pdims, obs, act, adv, om, ols, max_kl=0.01)
    d, dref = cpu(a['d']), ref['d']
    assert rel_l2(d, dref) <= TOL.CG_REL_L2 and abs(a['beta'] - ref['beta']) <= TOL.STEP_SCALE_RTOL * ref['beta']
    assert a['accepted'] == ref['accepted'] and a['n_backtrack'] == ref['n_backtrack']
    # rllab's literal route (one more product on the descent direction) behind the persistent solve
    e, te, took = _run(eng, batch, theta0, True, explicit_final_hvp=True)
    f, tf, _ = _run(eng, batch, theta0, False, explicit_final_hvp=True)
    assert took == 1 and torch.equal(e['d'], f['d']) and e['beta'] == f['beta'] and torch.equal(te, tf)


def test_persistent_cg_solve_early_exit_on_residual_tol():
    """krylov.cg leaves when r.r < residual_tol: the remaining products of the launch change nothing (cg_step_body returns at S_DONE)."""
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=6000, seed=27)
    rr = O.cg_optimize(th, pdims, obs, act, adv, om, ols, max_kl=0.01)['rdotr']
    cand = [(min(rr[:i]) / rr[i], i) for i in range(1, 9) if rr[i] < 0.5 * min(rr[:i])]
    i_exit = max(cand)[1]
    tol = float(np.sqrt(rr[i_exit] * min(rr[:i_exit])))
    batch = eng.make_batch(obs, act, adv, om, ols)
    theta0 = eng.get_policy().clone()
    a, ta, took = _run(eng, batch, theta0, True, residual_tol=tol)
    b, tb, _ = _run(eng, batch, theta0, False, residual_tol=tol)
    assert took == 1 and a['cg_iters_run'] == b['cg_iters_run'] == i_exit + 1 < 10
    assert torch.equal(a['d'], b['d']) and a['beta'] == b['beta'] and torch.equal(ta, tb)
