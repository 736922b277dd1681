"""GPU parity against fixtures made by the reference's OWN TF-graph code (tests/golden/make_golden_tf.py: the reference's
policy_model / dynamics_model closures, build_dynamics_graph, build_policy_graph, get_*_optimizer run unmodified on an eager
tf stand-in, float64).  The HIP path is compared with the REFERENCE's outputs directly -- the oracle is not involved.
Every call goes through the C ABI.  Tolerances: SURVEY 8d (fp32 device vs the float64 fixture)."""
import numpy as np
import pytest
import torch
from conftest import load_golden
import tolerances as TOL

pytestmark = pytest.mark.gpu

CASES = ['swimmer_2x64', 'half_cheetah_2x64', 'ant_2x64', 'swimmer_2x512', 'humanoid_3x128', 'hopper_2x32', 'snake_2x32']


def cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


def engine_from(d, dyn_prefix=''):
    import metrpo_amd
    env = str(d['env'])
    eng = metrpo_amd.Engine(env, int(d['K']), [int(h) for h in d['dyn_hidden']], [int(h) for h in d['pol_hidden']])
    L = len(d['dyn_hidden']) + 1
    eng.set_dynamics_layers([d['%sdynW%d' % (dyn_prefix, l)] for l in range(L)], [d['%sdynb%d' % (dyn_prefix, l)] for l in range(L)],
                            d['in_mean'], d['in_std'], d['diff_mean'], d['diff_std'])
    eng.set_policy(d['theta'])
    return eng


@pytest.mark.parametrize('case', CASES)
def test_all_heads_vs_reference_graph(case):
    """a1 (+a2 through the normalisers the reference's RunningMeanStd produced): K-head next states of metrpo_step."""
    d = load_golden('tfgraph_' + case)
    eng = engine_from(d)
    ns = eng.ns
    xu = d['xu']
    _, _, _, nall = eng.step(xu[:, :ns], xu[:, ns:], 'one_model', None, None, want_all=True)
    np.testing.assert_allclose(cpu(nall), d['dyn_out'], **TOL.wide_or_step(d['dyn_hidden']))


@pytest.mark.parametrize('case', CASES)
def test_policy_mean_vs_reference_graph(case):
    """a3: mean net of training.py:96-117 (metrpo_policy_actions with eps = NULL returns the mean)."""
    d = load_golden('tfgraph_' + case)
    eng = engine_from(d)
    a, m = eng.policy_actions(d['obs'], None)
    np.testing.assert_allclose(cpu(m), d['policy_mean'], **TOL.STEP)


@pytest.mark.parametrize('use_mfma', [True, False])
@pytest.mark.parametrize('case', CASES)
def test_validation_cost_vs_reference_graph(case, use_mfma):
    """a18: per-model costs of build_policy_graph (Ant with the running dones mask), both kernel families."""
    d = load_golden('tfgraph_' + case)
    eng = engine_from(d)
    eng.set_det_path(use_mfma)
    got = cpu(eng.validation_cost(d['x0'], int(d['T']), float(d['gamma'])))
    np.testing.assert_allclose(got, d['policy_costs'], **TOL.VALIDATION_COST)


@pytest.mark.parametrize('case', CASES)
def test_dynamics_losses_vs_reference_graph(case):
    """f1: per-model dynamics_losses of build_dynamics_graph on the get_ith_tensor slices = metrpo_dyn_train_step's reported
    losses (before the update); then three optimiser steps against get_dynamics_optimizer's."""
    d = load_golden('tfgraph_' + case)
    eng = engine_from(d)
    K, bs, reg, lr = eng.K, int(d['train_bs']), float(d['reg_constant']), float(d['train_lr'])
    eng.train_reset()
    got = cpu(eng.train_step(d['train_x'], d['train_y'], bs, 0.0, reg))          # lr = 0: losses only, weights unchanged
    np.testing.assert_allclose(got, d['dynamics_losses'], **TOL.DYN_LOSS)
    if 'step2_dynW0' not in d.files:
        return
    eng = engine_from(d)
    eng.train_reset()
    losses = [float(cpu(eng.train_step(d['step%d_x' % it], d['step%d_y' % it], bs, lr, reg)).sum()) for it in range(3)]
    np.testing.assert_allclose(losses, d['step_losses'], rtol=1e-4)
    flat = cpu(eng.get_dynamics())
    o = 0
    for l in range(len(d['dyn_hidden']) + 1):
        W, b = d['step2_dynW%d' % l], d['step2_dynb%d' % l]
        n = W.shape[1] * W.shape[2]
        gW = flat[:, o:o + n].reshape(W.shape); o += n
        gb = flat[:, o:o + b.shape[1]]; o += b.shape[1]
        # Adam moves a weight by ~lr per step whatever the gradient's size: entries whose fp32 gradient is rounding noise may
        # step the other way (see test_gpu_training); bound = 3 steps x lr, and 99.9 % far closer
        for got_, ref_ in ((gW, W), (gb, b)):
            err = np.abs(got_ - ref_)
            assert (err > 5e-5).mean() < 2e-3 and err.max() <= 3.3 * lr


@pytest.mark.parametrize('case', [c for c in CASES if c != 'humanoid_3x128'])
def test_bptt_vs_reference_graph(case):
    """f3: policy_grads_and_vars of get_policy_optimizer (autograd through the reference's unrolled graph) and its Adam steps
    on the per-variable clip_by_norm'ed gradient."""
    d = load_golden('tfgraph_' + case)
    eng = engine_from(d)
    T, gamma = int(d['T']), float(d['gamma'])
    costs, grad = eng.bptt_grad(d['x0'], T, gamma)
    np.testing.assert_allclose(cpu(costs), d['policy_costs'], **TOL.BPTT_COST)
    g, ref = cpu(grad), d['bptt_grad']
    assert np.linalg.norm(g - ref) < TOL.BPTT_GRAD_REL_L2 * np.linalg.norm(ref)
    assert float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref))) > 1 - 1e-7
    eng.policy_adam_reset()
    lr, clip = float(d['bptt_lr']), float(d['clip'])
    for it in range(3):
        _, grad = eng.bptt_grad(d['x0'], T, gamma)
        eng.policy_adam_step(grad, lr, clip)
        th = cpu(eng.get_policy())
        err = np.abs(th - d['bptt_thetas'][it])
        assert (err > 2e-5).mean() < 5e-3 and err.max() <= (it + 1) * 1.1 * lr
