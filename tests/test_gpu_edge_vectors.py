"""The reference-held edge-case vectors (tests/golden/rewards_*.npz: HalfCheetah's +-10 clip, Hopper's three penalty terms;
ant_done.npz: NaN / +-Inf rows and the 0.2 / 1.0 boundaries -- produced by the reference's own cost_np_vec / is_done) pushed
through the HIP kernels.  With all-zero dynamics weights, zero diff_mean and a zero policy the imagined step is the identity
(next = s) and the action equals the supplied noise, so the kernels' reward / termination code sees exactly the fixture rows."""
import numpy as np
import pytest
import torch
from conftest import load_golden
from oracle import metrpo_oracle as O

pytestmark = pytest.mark.gpu
ENVS = list(O.ENV_SPECS)


def cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


def identity_engine(env, K, hidden, pol_hidden=(32, 32)):
    import metrpo_amd
    ns, na, _ = O.ENV_SPECS[env]
    eng = metrpo_amd.Engine(env, K, hidden, pol_hidden)
    z = torch.zeros(K, eng.dyn_param_count)
    eng.set_dynamics(z, np.zeros(ns + na), np.ones(ns + na), np.zeros(ns), np.ones(ns))
    eng.set_policy(np.zeros(eng.P))                                   # mean = 0, log_std = 0  ->  action = eps
    return eng, ns, na


@pytest.mark.parametrize('env', ENVS)
def test_reward_vectors_through_step(env):
    d = load_golden('rewards_' + env)
    eng, ns, na = identity_engine(env, 3, (64, 64))
    xn, u = d['x_next'], d['u']
    s_next, rew, done = eng.step(xn, u, 'one_model', None, None)
    np.testing.assert_array_equal(cpu(s_next), xn.astype(np.float32).astype(np.float64))      # identity dynamics
    np.testing.assert_allclose(cpu(rew), -d['cost'], rtol=2e-6, atol=2e-6)
    if env == 'half_cheetah':
        assert (np.abs(d['cost']) == 10.0).sum() >= 2                 # the clip rows are in the fixture
    if env == 'hopper':
        assert (np.abs(xn[:, 2:]) > 100).any() and (xn[:, 0] < 0.45).any() and (np.abs(xn[:, 1]) > 0.2).any()


def test_ant_termination_vectors_through_step():
    d = load_golden('ant_done')
    eng, ns, na = identity_engine('ant', 3, (64, 64))
    xn = d['x_next']
    _, _, done = eng.step(xn, np.zeros((len(xn), na)), 'one_model', None, None)
    assert np.array_equal(cpu(done).astype(bool), d['done'])
    assert not np.isfinite(xn).all() and d['done'][:8].tolist() == [False, False, True, True, True, True, True, True]


@pytest.mark.parametrize('variant', ['generic', 'head_per_wave', 'coop', 'gemm'])
@pytest.mark.parametrize('env', ENVS)
def test_edge_vectors_through_fused_rollouts(env, variant):
    """Same rows through every rollout kernel family (their reward / done epilogues are separate code)."""
    K = 5
    hidden = (128, 128) if variant == 'gemm' else (64, 64)
    eng, ns, na = identity_engine(env, K, hidden)
    running = eng.set_rollout_variant(1 if variant == 'head_per_wave' else 0)
    want = dict(generic=None, head_per_wave=1, coop=2, gemm=3)[variant]
    if want is not None and running != want:
        pytest.skip('no %s kernel for %s (running variant %d)' % (variant, env, running))
    rows = [load_golden('rewards_' + env)]
    xn, u, ref_rew = rows[0]['x_next'], rows[0]['u'], -rows[0]['cost']
    ref_done = np.zeros(len(xn), bool)
    if env == 'ant':
        a = load_golden('ant_done')
        n = len(a['x_next'])
        xn = np.concatenate([xn, a['x_next']]); u = np.concatenate([u, np.zeros((n, na))])
        ref_rew = np.concatenate([ref_rew, -O.cost_np_vec('ant', a['x_next'], np.zeros((n, na)), a['x_next'])])
        ref_done = np.concatenate([O.is_done('ant', rows[0]['x_next'], rows[0]['x_next']), a['done']])
    B = len(xn)
    zeros_i = np.zeros((2, B), np.int32)
    traj = eng.rollout(B, 1, 1000, 'eps_rand', xn.astype(np.float32), eps=u[None].astype(np.float32),
                       reset_idx=np.stack([np.arange(B), np.zeros(B)]).astype(np.int32), reset_model=zeros_i,
                       force_generic=(variant == 'generic'))
    fin = np.isfinite(xn).all(axis=1)      # non-finite states make the policy output NaN: np.clip keeps NaN, fminf/fmaxf do not -- the reward of such a row is meaningless in both
    np.testing.assert_allclose(cpu(traj.rew[0])[fin], ref_rew[fin], rtol=2e-6, atol=2e-6)
    assert np.array_equal(cpu(traj.done[0]).astype(bool), ref_done)
    np.testing.assert_array_equal(cpu(traj.act[0])[fin], u.astype(np.float32).astype(np.float64)[fin])
