"""Pin the TF-GRAPH half of the oracle against fixtures produced by the reference's OWN graph-building code
(tests/golden/make_golden_tf.py: training.py policy_model / dynamics_model closures, model_based_rl.build_dynamics_graph /
build_policy_graph / get_*_optimizer, RunningMeanStd, envs' cost_tf / is_done_tf executed unmodified on an eager tf stand-in,
float64).  CPU only.  Rows of SURVEY 8a covered: a1, a2, a3, a18; 8f: rank 1 (loss, gradient, optimiser step), rank 2
(normaliser), rank 3 (BPTT gradient, clipped Adam step)."""
import numpy as np
import pytest
from conftest import load_golden
from oracle import metrpo_oracle as O
from oracle import dynamics_oracle as DO
from oracle import bptt_oracle as BO

CASES = ['swimmer_2x64', 'half_cheetah_2x64', 'ant_2x64', 'swimmer_2x512', 'humanoid_3x128', 'hopper_2x32', 'snake_2x32']


def dm_from_tfgraph(d, prefix='', dtype=np.float64):
    env = str(d['env'])
    ns, na, n_drop = O.ENV_SPECS[env]
    assert int(d['n_drop']) == n_drop                     # the reference's params file for that env says the same
    L = len(d['dyn_hidden']) + 1
    Ws = [d['%sdynW%d' % (prefix, l)].astype(dtype) for l in range(L)]
    bs = [d['%sdynb%d' % (prefix, l)].astype(dtype) for l in range(L)]
    return O.DynamicsEnsemble(Ws, bs, ['relu'] * (L - 1), d['in_mean'].astype(dtype), d['in_std'].astype(dtype),
                              d['diff_mean'].astype(dtype), d['diff_std'].astype(dtype), n_drop, ns, na)


def pdims(d):
    ns, na, _ = O.ENV_SPECS[str(d['env'])]
    return O.policy_dims(ns, [int(h) for h in d['pol_hidden']], na)


@pytest.mark.parametrize('case', CASES)
def test_normaliser(case):
    """a2: RunningMeanStd.mean/std from the sums the reference's update() accumulated; the 0.1 floor is active."""
    d = load_golden('tfgraph_' + case)
    m, s = O.rms_mean_std(d['rms_in_sum'], d['rms_in_sumsq'], d['rms_in_count'])
    np.testing.assert_allclose(m, d['in_mean'], rtol=1e-14); np.testing.assert_allclose(s, d['in_std'], rtol=1e-14)
    m, s = O.rms_mean_std(d['rms_diff_sum'], d['rms_diff_sumsq'], d['rms_diff_count'])
    np.testing.assert_allclose(m, d['diff_mean'], rtol=1e-14); np.testing.assert_allclose(s, d['diff_std'], rtol=1e-14)
    assert d['in_std'][1] == pytest.approx(0.1, abs=1e-15) and d['diff_std'][3] == pytest.approx(0.1, abs=1e-15)
    assert (d['in_std'] > 0.1).sum() >= len(d['in_std']) - 1


@pytest.mark.parametrize('case', CASES)
def test_dynamics_heads(case):
    """a1: all K heads (the `_dynamics_outs` of build_dynamics_graph at test time)."""
    d = load_golden('tfgraph_' + case)
    dm = dm_from_tfgraph(d)
    ns = dm.ns
    got = O.dynamics_forward_all(dm, d['xu'][:, :ns], d['xu'][:, ns:])
    np.testing.assert_allclose(got, d['dyn_out'], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize('case', CASES)
def test_dynamics_losses(case):
    """f1: per-model prediction loss on the get_ith_tensor slices, regulariser, totals."""
    d = load_golden('tfgraph_' + case)
    dm = dm_from_tfgraph(d)
    K, bs, reg = dm.K, int(d['train_bs']), float(d['reg_constant'])
    xs, ys = DO.split_batch(d['train_x'], d['train_y'], bs, K)
    pl = DO.prediction_losses(dm, xs, ys)
    rl = np.array([DO.regularizer_loss(dm, k, reg) for k in range(K)])
    np.testing.assert_allclose(pl + rl, d['dynamics_losses'], rtol=1e-11)
    np.testing.assert_allclose(pl.sum(), d['prediction_loss'], rtol=1e-11)
    np.testing.assert_allclose(rl.sum(), d['regularizer_loss'], rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(pl.sum() + rl.sum(), d['dynamics_loss'], rtol=1e-11)
    if reg > 0:
        assert d['regularizer_loss'] > 0


@pytest.mark.parametrize('case', [c for c in CASES if c != 'swimmer_2x512'])
def test_dynamics_gradient(case):
    """f1: d(sum_i prediction_loss_i)/d(W, b) -- autograd over the reference's loss graph vs the oracle's back-prop."""
    d = load_golden('tfgraph_' + case)
    dm = dm_from_tfgraph(d)
    K, bs = dm.K, int(d['train_bs'])
    xs, ys = DO.split_batch(d['train_x'], d['train_y'], bs, K)
    for k in range(K):
        gW, gb = DO.model_gradients(dm, k, xs[k], ys[k])
        for l in range(len(gW)):
            np.testing.assert_allclose(gW[l], d['gradW%d' % l][k], rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(gb[l], d['gradb%d' % l][k], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize('case', [c for c in CASES if c != 'swimmer_2x512'])
def test_dynamics_optimiser_steps(case):
    """f1: three sess.run([opt_op, loss]) of get_dynamics_optimizer (Adam on the prediction loss + SGD on the regulariser)."""
    d = load_golden('tfgraph_' + case)
    dm = dm_from_tfgraph(d)
    adam = DO.AdamState(dm)
    bs, reg, lr = int(d['train_bs']), float(d['reg_constant']), float(d['train_lr'])
    losses = [DO.train_step(dm, adam, d['step%d_x' % it], d['step%d_y' % it], bs, lr, reg) for it in range(3)]
    # the regulariser's SGD step reads W before (oracle) or after (eager stand-in) Adam's update: an O(lr^2 * reg) = 1e-10
    # difference per step that is not a property of the reference (TF does not order the two ops); exact when reg = 0
    tol = 1e-7 if reg > 0 else 1e-11
    np.testing.assert_allclose(losses, d['step_losses'], rtol=tol)
    for l in range(len(dm.Ws)):
        np.testing.assert_allclose(dm.Ws[l], d['step2_dynW%d' % l], rtol=0, atol=tol)
        np.testing.assert_allclose(dm.bs[l], d['step2_dynb%d' % l], rtol=0, atol=tol)


@pytest.mark.parametrize('case', CASES)
def test_policy_mean(case):
    """a3: training.py:96-117 policy_model with stochastic = 0."""
    d = load_golden('tfgraph_' + case)
    got = O.policy_mean(d['theta'], pdims(d), d['obs'])
    np.testing.assert_allclose(got, d['policy_mean'], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('case', CASES)
def test_validation_costs(case):
    """a18: build_policy_graph per-model costs (Ant: running dones mask, updated after the cost of the step)."""
    d = load_golden('tfgraph_' + case)
    dm = dm_from_tfgraph(d)
    env = str(d['env'])
    got = O.validation_costs(dm, d['theta'], pdims(d), env, d['x0'], int(d['T']), float(d['gamma']))
    np.testing.assert_allclose(got, d['policy_costs'], rtol=1e-10, atol=1e-12)
    if env == 'ant':                                       # the mask must have mattered
        unmasked = np.array([_unmasked_cost(dm, d, i) for i in range(dm.K)])
        assert np.abs(unmasked - d['policy_costs']).max() > 1e-3


def _unmasked_cost(dm, d, i):
    x, c = d['x0'].copy(), 0.0
    for t in range(int(d['T'])):
        u = np.clip(O.policy_mean(d['theta'], pdims(d), x), -1, 1)
        xn = O.dynamics_forward(dm, i, x, u)
        c += float(d['gamma']) ** t * np.mean(O.cost_np_vec('ant', x, u, xn)); x = xn
    return c


@pytest.mark.parametrize('case', [c for c in CASES if c != 'humanoid_3x128'])
def test_bptt_gradient_and_steps(case):
    """f3: gradient of reduce_mean(policy_costs) (policy_grads_and_vars of get_policy_optimizer) and three Adam steps on the
    per-variable clip_by_norm'ed gradient.  (Humanoid's cost_tf applies np.square to a tensor -- forward only.)"""
    d = load_golden('tfgraph_' + case)
    dm = dm_from_tfgraph(d)
    env, dims, T, gamma = str(d['env']), pdims(d), int(d['T']), float(d['gamma'])
    costs, g = BO.policy_costs_and_grad(dm, d['theta'], dims, env, d['x0'], T, gamma)
    np.testing.assert_allclose(costs, d['policy_costs'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.mean(costs), d['training_policy_cost'], rtol=1e-10)
    ref = d['bptt_grad']
    assert np.linalg.norm(g - ref) <= 1e-9 * np.linalg.norm(ref)
    adam = BO.PolicyAdam(len(g))
    theta = d['theta'].copy()
    for it in range(3):
        _, g = BO.policy_costs_and_grad(dm, theta, dims, env, d['x0'], T, gamma)
        theta = adam.step(theta, g, dims, float(d['bptt_lr']), float(d['clip']))
        np.testing.assert_allclose(theta, d['bptt_thetas'][it], rtol=0, atol=1e-10)
    if case == 'ant_2x64':                                 # clip 0.05 was active for at least one variable
        assert np.linalg.norm(ref) > 0.05
