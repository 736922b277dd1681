"""BPTT policy update on the GPU (csrc/bptt.hip) vs the float64 restatement (oracle/bptt_oracle.py, itself checked against
torch autograd in tests/test_oracle_bptt.py) -- SURVEY.md 8f rank 3.  All calls go through the C ABI."""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
from oracle import bptt_oracle as Bp
import helpers as Hh
import tolerances as TOL

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


def rel_l2(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize('env,K,dh,ph,B,T,gamma', [
    ('swimmer', 5, (64, 64), (32, 32), 300, 20, 1.0),          # MFMA gradient kernels for the parameter reduction
    ('swimmer', 3, (24, 16), (8, 8), 100, 15, 0.97),           # generic gradient kernels
    ('half_cheetah', 2, (64, 64), (32, 32), 70, 12, 0.99),
    ('hopper', 2, (32, 32), (32, 32), 90, 10, 1.0),
    ('ant', 3, (64, 64), (32, 32), 130, 14, 0.95),             # running `dones` mask
    ('snake', 2, (64, 64), (32, 32), 64, 9, 1.0),
    ('humanoid', 2, (48, 48, 32), (20, 10, 5), 40, 6, 1.0),    # three hidden layers on both nets
    ('swimmer', 4, (48, 20), (32, 32), 150, 12, 0.98),         # round 6: narrow nets on the MFMA sweeps over the zero-padded weights
    ('ant', 3, (40, 40), (32, 32), 100, 10, 0.95),
])
def test_bptt_gradient_matches_oracle(env, K, dh, ph, B, T, gamma):
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, dh, ph, seed=91)
    rng = np.random.RandomState(5)
    theta = theta + 0.25 * rng.randn(theta.size)                # some actions saturate the clip
    theta[-dm.na:] = 0.0
    eng.set_policy(theta)
    x0 = pool[:B].copy() * (3.0 if env in ('hopper', 'half_cheetah') else 1.0)
    if env == 'ant':
        x0[:B // 4, 2] = 0.15                                   # done from the first step on
    if env == 'hopper':
        x0[:10, 1] = 0.5; x0[10:20, 0] = 0.2; x0[20:24, 4] = 150.0
    x0 = x0.astype(np.float32)
    costs, grad = eng.bptt_grad(x0, T, gamma)
    th32 = cpu(eng.get_policy()).astype(np.float64)
    dm32 = dm.astype(np.float32).astype(np.float64)
    oc, og = Bp.policy_costs_and_grad(dm32, th32, pdims, env, x0.astype(np.float64), T, gamma)
    # forward costs: the same quantity metrpo_validation_cost returns (fp32 rollout vs fp64: T steps of chaotic growth are short here)
    np.testing.assert_allclose(cpu(costs), oc, **TOL.BPTT_COST)
    np.testing.assert_allclose(cpu(costs), cpu(eng.validation_cost(x0, T, gamma)), rtol=1e-6, atol=1e-7)
    g = cpu(grad)
    assert np.all(g[-dm.na:] == 0.0)
    # tolerance: fp32 forward/backward through T chained Jacobians; SURVEY 8d asks rel-L2 <= 1e-5 for one-step gradients, the
    # T-step chain amplifies rounding like the 10-iteration CG does (1e-3 there)
    assert rel_l2(g, og) < TOL.BPTT_GRAD_REL_L2, rel_l2(g, og)
    cosine = float(g @ og / (np.linalg.norm(g) * np.linalg.norm(og)))
    assert cosine > 1.0 - 1e-7


@pytest.mark.parametrize('env,dh', [('swimmer', (64, 64)), ('half_cheetah', (64, 64)), ('hopper', (64, 64)), ('snake', (64, 64)), ('ant', (64, 64)),
                                    ('swimmer', (32, 50)), ('ant', (16, 16))])
def test_mfma_and_generic_sweeps_agree(env, dh):
    """every env of the MFMA table: MFMA sweeps (default) vs the generic sweep kernels vs the oracle, incl. ragged batch sizes; round 6: also narrow nets,
    whose MFMA sweeps read the zero-padded copy of the weights."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 3, dh, (32, 32), seed=95)
    rng = np.random.RandomState(6)
    theta = theta + 0.2 * rng.randn(theta.size); theta[-dm.na:] = 0.0
    eng.set_policy(theta)
    B, T, gamma = 77, 11, 0.98                                   # 77 = one full 64-env workgroup + a ragged 13-env tile
    x0 = (pool[:B] * (3.0 if env in ('hopper', 'half_cheetah') else 1.0)).astype(np.float32)
    if env == 'ant':
        x0[:20, 2] = 0.15
    if env == 'hopper':
        x0[:10, 1] = 0.5; x0[10:20, 0] = 0.2; x0[20:24, 4] = 150.0
    assert eng.set_det_path(True) == 1
    cm, gm = eng.bptt_grad(x0, T, gamma); vm = eng.validation_cost(x0, T, gamma)
    cm, gm, vm = cpu(cm), cpu(gm), cpu(vm)
    assert eng.set_det_path(False) == 0
    cg, gg = eng.bptt_grad(x0, T, gamma); vg = eng.validation_cost(x0, T, gamma)
    eng.set_det_path(True)
    th32 = cpu(eng.get_policy()).astype(np.float64)
    oc, og = Bp.policy_costs_and_grad(dm.astype(np.float32).astype(np.float64), th32, pdims, env, x0.astype(np.float64), T, gamma)
    np.testing.assert_allclose(cm, oc, **TOL.BPTT_COST); np.testing.assert_allclose(vm, cm, rtol=1e-12)
    np.testing.assert_allclose(cpu(cg), oc, **TOL.BPTT_COST); np.testing.assert_allclose(cpu(vg), cpu(cg), rtol=1e-6, atol=1e-7)
    assert rel_l2(gm, og) < TOL.BPTT_GRAD_REL_L2 and rel_l2(cpu(gg), og) < TOL.BPTT_GRAD_REL_L2 and rel_l2(gm, cpu(gg)) < TOL.BPTT_GRAD_REL_L2


@pytest.mark.parametrize('env,K,dh,ph,B,T', [('swimmer', 3, (128, 128), (32, 32), 150, 9), ('half_cheetah', 2, (256, 128), (32, 32), 70, 7),
                                             ('ant', 2, (128, 160, 128), (24, 16), 90, 8), ('hopper', 2, (128, 128), (32, 32), 64, 8),
                                             ('humanoid', 2, (128, 128), (20, 10, 5), 40, 5)])
def test_gemm_path_sweeps_match_oracle_and_generic(env, K, dh, ph, B, T):
    """large dynamics nets (hidden >= 128): GEMM-path sweeps (det_gemm.hip) vs the generic sweep kernels vs the oracle."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, dh, ph, seed=96)
    rng = np.random.RandomState(7)
    theta = theta + 0.2 * rng.randn(theta.size); theta[-dm.na:] = 0.0
    eng.set_policy(theta)
    gamma = 0.97
    x0 = (pool[:B] * (3.0 if env in ('hopper', 'half_cheetah') else 1.0)).astype(np.float32)
    if env == 'ant':
        x0[:20, 2] = 0.15
    if env == 'hopper':
        x0[:10, 1] = 0.5; x0[10:20, 0] = 0.2; x0[20:24, 4] = 150.0
    assert eng.set_det_path(True) == 2
    cm, gm = eng.bptt_grad(x0, T, gamma); vm = eng.validation_cost(x0, T, gamma)
    cm, gm, vm = cpu(cm), cpu(gm), cpu(vm)
    assert eng.set_det_path(False) == 0
    cg, gg = eng.bptt_grad(x0, T, gamma); vg = eng.validation_cost(x0, T, gamma)
    eng.set_det_path(True)
    th32 = cpu(eng.get_policy()).astype(np.float64)
    oc, og = Bp.policy_costs_and_grad(dm.astype(np.float32).astype(np.float64), th32, pdims, env, x0.astype(np.float64), T, gamma)
    np.testing.assert_allclose(cm, oc, **TOL.BPTT_COST); np.testing.assert_allclose(vm, cm, rtol=1e-12)
    np.testing.assert_allclose(cpu(cg), oc, **TOL.BPTT_COST); np.testing.assert_allclose(cpu(vg), cpu(cg), rtol=1e-6, atol=1e-7)
    assert rel_l2(gm, og) < TOL.BPTT_GRAD_REL_L2 and rel_l2(cpu(gg), og) < TOL.BPTT_GRAD_REL_L2 and rel_l2(gm, cpu(gg)) < TOL.BPTT_GRAD_REL_L2


def test_bptt_gradient_is_bitwise_reproducible_and_linear_in_weights():
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 5, (64, 64), (32, 32), seed=92)
    x0 = pool[:256].astype(np.float32)
    c1, g1 = eng.bptt_grad(x0, 25, 0.99)
    c2, g2 = eng.bptt_grad(x0, 25, 0.99)
    assert torch.equal(g1, g2) and torch.equal(c1, c2)
    # size-independent property: the batch mean is linear -- the gradient on a batch equals the mean of the gradients of its halves
    _, ga = eng.bptt_grad(x0[:128], 25, 0.99)
    _, gb = eng.bptt_grad(x0[128:], 25, 0.99)
    assert rel_l2(cpu(g1), 0.5 * (cpu(ga) + cpu(gb))) < 1e-6


def test_policy_adam_step_matches_oracle():
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 5, (64, 64), (32, 32), seed=93)
    P = theta.size
    rng = np.random.RandomState(2)
    adam = Bp.PolicyAdam(P)
    th = cpu(eng.get_policy()).astype(np.float64)
    eng.policy_adam_reset()
    for it in range(5):
        g = rng.randn(P) * (5.0 if it % 2 == 0 else 1e-3)
        g[-dm.na:] = 0.0
        th = adam.step(th, g, pdims, lr=3e-3, clip_val=1.5)
        eng.policy_adam_step(g, 3e-3, clip_val=1.5)
        np.testing.assert_allclose(cpu(eng.get_policy()), th, rtol=2e-6, atol=2e-7)
    # no clipping and a reset restart the moments
    eng.policy_adam_reset(); adam = Bp.PolicyAdam(P)
    g = rng.randn(P); g[-dm.na:] = 0.0
    th = adam.step(cpu(eng.get_policy()).astype(np.float64), g, pdims, lr=1e-2, clip_val=None)
    eng.policy_adam_step(g, 1e-2)
    np.testing.assert_allclose(cpu(eng.get_policy()), th, rtol=2e-6, atol=2e-7)


def test_bptt_training_loop_lowers_the_cost():
    """the 'bptt' branch run for a few iterations (model_based_rl.py:1181-1187): cost on fixed validation states goes down,
    and the host mirror follows the oracle trajectory."""
    import metrpo_amd
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 3, (32, 32), (16, 16), seed=94)
    T, gamma, lr, clip = 15, 1.0, 2e-2, 1.0
    opt = metrpo_amd.BPTT(eng, T=T, gamma=gamma, learning_rate=lr, grad_norm_clipping=clip, batch_size=128)
    val = pool[128:256].astype(np.float32)
    c0 = float(cpu(eng.validation_cost(val, T, gamma)).mean())
    th = cpu(eng.get_policy()).astype(np.float64)
    adam = Bp.PolicyAdam(th.size)
    dm32 = dm.astype(np.float32).astype(np.float64)
    for it in range(6):
        xb = pool[(it * 16) % 64:(it * 16) % 64 + 128].astype(np.float32)
        cost = float(opt.step(xb))
        oc, og = Bp.policy_costs_and_grad(dm32, th, pdims, 'swimmer', xb.astype(np.float64), T, gamma)
        assert abs(cost - oc.mean()) < 2e-4 * max(1.0, abs(oc.mean()))
        th = adam.step(th, og, pdims, lr=lr, clip_val=clip)
    np.testing.assert_allclose(cpu(eng.get_policy()), th, rtol=2e-3, atol=2e-4)      # Adam normalises: early steps are sign-like
    c1 = float(cpu(eng.validation_cost(val, T, gamma)).mean())
    assert c1 < c0


def test_bptt_error_behaviour():
    import metrpo_amd
    eng = metrpo_amd.Engine('swimmer', 2, (16, 16), (8, 8))
    with pytest.raises(RuntimeError):
        eng.bptt_grad(np.zeros((4, 10), np.float32), 5, 1.0)     # no dynamics / policy yet
