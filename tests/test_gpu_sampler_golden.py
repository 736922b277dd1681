"""The reference's OWN sampler runs (tests/golden/sampler_*.npz: samplers/vectorized_sampler.py:45-116 +
samplers/base.py:48-104 executed in the build container) replayed through the HIP VectorizedSampler / process_samples with the
reference's exact random draws: path count, lengths, order, per-sample values, advantages, returns, refitted baseline.
Includes the early-terminating Ant case (step-granular stop rule, dropped open paths) with the rollout issued in chunks."""
import numpy as np
import pytest
import torch
from conftest import load_golden, dm_from_golden
from oracle import metrpo_oracle as O
import helpers as Hh
import tolerances as TOL

pytestmark = pytest.mark.gpu

SAMPLERS = ['sampler_0_swimmer_step_rand', 'sampler_1_swimmer_step_rand', 'sampler_2_swimmer_eps_rand',
            'sampler_3_ant_step_rand', 'sampler_4_half_cheetah_model_mean_std']


def cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


def build(d, chunk=None):
    import metrpo_amd
    env, sam_mode = str(d['env']), str(d['sam_mode'])
    dm = dm_from_golden(d, env)
    hidden = [w.shape[2] for w in dm.Ws[:-1]]
    pdims = [int(x) for x in d['pdims']]
    eng = metrpo_amd.Engine(env, dm.K, hidden, pdims[1:-1])
    eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=0)
    eng.set_policy(d['theta'])
    nne = metrpo_amd.NeuralNetEnv(env=metrpo_amd.InitStatePool(d['pool'], dm.na), inner_env=None, cost_np=env, dynamics_in=None,
                                  dynamics_outs=eng, sam_mode=sam_mode)
    base = metrpo_amd.LinearFeatureBaseline()
    if bool(d['has_coeffs']):
        base.set_param_values(d['coeffs_before'].copy())
    algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=base, batch_size=int(d['batch_size']), max_path_length=int(d['H']),
                           discount=float(d['gamma']), gae_lambda=float(d['lam']), step_size=0.01,
                           sampler_args=dict(n_envs=int(d['B'])))
    if chunk:
        algo.sampler_chunk = chunk
    return algo, eng, dm


@pytest.mark.parametrize('chunk', [None, 3])
@pytest.mark.parametrize('name', SAMPLERS)
def test_reference_sampler_run_through_hip(name, chunk):
    d = load_golden(name)
    env = str(d['env'])
    if chunk and env != 'ant':
        pytest.skip('chunking only applies to early-terminating envs')
    algo, eng, dm = build(d, chunk)
    H, B = int(d['H']), int(d['B'])
    dr, T_ref = Hh.reference_sampler_draws(d, T_pad=H + 4)           # padding: the chunked Ant loop may roll past the stop step
    dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v.astype(np.int32)) for k, v in dr.items()}
    dev = {k: torch.as_tensor(v, device=eng.device) for k, v in dr32.items()}
    if env != 'ant':                                                  # whole rounds: exactly the reference's step count
        dev = {k: (v[:T_ref + 1] if k.startswith('reset') else v[:T_ref]) for k, v in dev.items()}
    algo.start_worker()
    paths = algo.obtain_samples(0, determ=bool(d['determ']), draws=dev)
    assert paths.traj.T == T_ref                                      # same number of env steps as the reference took
    plist = paths.to_paths()
    assert len(plist) == int(d['n_paths'])
    assert [len(p['rewards']) for p in plist] == list(d['lengths'])
    cat = lambda k: np.concatenate([p[k] for p in plist])
    tol = TOL.LONG_RUN                                                # row 4: free-running fp32 rollout vs the float64 reference run
    np.testing.assert_allclose(cat('observations'), d['observations'], **tol)
    np.testing.assert_allclose(cat('actions'), d['actions'], **tol)
    np.testing.assert_allclose(cat('rewards'), d['rewards'], **tol)
    np.testing.assert_allclose(np.concatenate([p['agent_infos']['mean'] for p in plist]), d['mean'], **tol)
    np.testing.assert_allclose(np.concatenate([p['agent_infos']['log_std'] for p in plist]), d['log_std'], atol=1e-6)
    if env == 'ant':
        assert len(set(d['lengths'])) > 1 and sum(d['lengths']) < T_ref * B      # ragged paths, open paths dropped

    samples = algo.process_samples(0, paths)
    tr = paths.traj
    done = cpu(tr.done).astype(bool)
    order, start = [], np.zeros(B, int)                               # sample order of the reference: paths in completion order
    for t, b in zip(*np.nonzero(done)):
        order += [tt * B + b for tt in range(start[b], t + 1)]; start[b] = t + 1
    order = np.array(order)
    v = cpu(samples['valids']).astype(bool)
    assert v.sum() == len(order) == len(d['s_advantages']) == samples['n_valid_global'] and v[order].all()
    np.testing.assert_allclose(cpu(samples['returns'])[order], d['s_returns'], **TOL.RETURNS)
    np.testing.assert_allclose(cpu(samples['advantages'])[order], d['s_advantages'], **TOL.ADVANTAGE_CENTRED)
    np.testing.assert_allclose(cpu(samples['observations'])[order], d['s_observations'], **tol)
    # refitted baseline (base.py:164-167) predicts like the reference's
    F = np.concatenate([O.LinearFeatureBaselineOracle.features(dict(observations=p['observations'], rewards=p['rewards'])) for p in plist])
    ref_pred = F @ d['coeffs_after']
    np.testing.assert_allclose(F @ algo.baseline.coeffs, ref_pred, rtol=0, atol=TOL.BASELINE_FIT * max(1.0, np.abs(ref_pred).max()))
