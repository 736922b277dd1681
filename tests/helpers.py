"""Shared helpers for the GPU parity tests: time-major replay of the oracle with explicit draws."""
import numpy as np
from oracle import metrpo_oracle as O


def make_engine(env, K, dyn_hidden, pol_hidden, seed=0, n_pool=256, dyn_act='relu'):
    import metrpo_amd
    dm, theta, pdims, pool = O.make_problem(env, K=K, dyn_hidden=dyn_hidden, pol_hidden=pol_hidden, seed=seed,
                                            n_pool=n_pool, dyn_act=dyn_act)
    rng = np.random.RandomState(seed + 1000)
    # non-trivial normalisers and a perturbed policy so that nothing is tested at an identity/zero point
    ns, na = dm.ns, dm.na
    dm.in_mean = rng.randn(ns + na) * 0.05
    dm.in_std = np.maximum(np.abs(1.0 + rng.randn(ns + na) * 0.2), 0.1)
    dm.diff_mean = rng.randn(ns) * 0.01
    dm.diff_std = np.abs(0.1 + rng.randn(ns) * 0.02)
    theta = theta + rng.randn(theta.size) * 0.05
    theta[-na:] = rng.randn(na) * 0.2 - 0.3
    eng = metrpo_amd.Engine(env, K, dyn_hidden, pol_hidden, dyn_act=dyn_act)
    eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    eng.set_policy(theta)
    return eng, dm, theta, pdims, pool


def draws(rng, K, B, T, ns, na, n_pool):
    return dict(eps=rng.randn(T, B, na), model_idx=rng.randint(K, size=(T, B)), sel_noise=rng.randn(T, B, ns),
                reset_idx=rng.randint(n_pool, size=(T + 1, B)), reset_model=rng.randint(K, size=(T + 1, B)))


def oracle_rollout(dm, theta, pdims, env, pool, dr, B, T, H, sam_mode, determ=False, teacher_obs=None):
    """Time-major oracle rollout with the draws `dr`.  teacher_obs [T,B,ns] (optional) replaces the
    oracle's own state at every step (teacher forcing) so per-step parity is not polluted by the
    chaotic growth of fp32-vs-fp64 differences."""
    K, ns, na = dm.K, dm.ns, dm.na
    out = dict(obs=np.zeros((T, B, ns)), act=np.zeros((T, B, na)), rew=np.zeros((T, B)), mean=np.zeros((T, B, na)),
               done=np.zeros((T, B), bool), tpath=np.zeros((T, B), np.int64), next=np.zeros((T, B, ns)))
    s = pool[dr['reset_idx'][0]].copy()
    cur_model = dr['reset_model'][0].copy()
    ts = np.zeros(B)
    for t in range(T):
        if teacher_obs is not None:
            s = teacher_obs[t].astype(np.float64)
        a, info = O.policy_get_actions(theta, pdims, s, dr['eps'][t])
        if determ:
            a = info['mean']
        ac = np.clip(a, -1, 1)
        idx = dr['model_idx'][t] if sam_mode == 'step_rand' else cur_model
        nxt = O.select_next(O.dynamics_forward_all(dm, s, ac), sam_mode, idx, dr['sel_noise'][t])
        out['obs'][t], out['act'][t], out['mean'][t] = s, a, info['mean']
        out['rew'][t] = -O.cost_np_vec(env, s, ac, nxt)
        out['next'][t] = nxt
        ts += 1
        dn = O.is_done(env, nxt, nxt) | (ts >= H)
        out['done'][t], out['tpath'][t] = dn, ts - 1
        s = np.where(dn[:, None], pool[dr['reset_idx'][t + 1]], nxt)
        cur_model = np.where(dn, dr['reset_model'][t + 1], cur_model)
        ts[dn] = 0
    out['last_obs'] = s
    return out


def paths_from_timemajor(tr):
    """Split time-major arrays into the reference's list of path dicts, in the sampler's completion
    order (time step, then env index); trailing unfinished paths are dropped."""
    T, B = tr['rew'].shape
    start = np.zeros(B, int)
    paths = []
    for t in range(T):
        for b in range(B):
            if tr['done'][t, b]:
                sl = slice(start[b], t + 1)
                paths.append(dict(observations=tr['obs'][sl, b], actions=tr['act'][sl, b], rewards=tr['rew'][sl, b],
                                  agent_infos=dict(mean=tr['mean'][sl, b], log_std=np.zeros_like(tr['mean'][sl, b])),
                                  _tb=[(tt, b) for tt in range(start[b], t + 1)]))
                start[b] = t + 1
    return paths


def reference_sampler_draws(d, T_pad=0):
    """Turn the np.random stream the REFERENCE's VectorizedSampler consumed (recorded in a sampler_*.npz golden) into the
    explicit per-step draw tensors of metrpo_rollout.  The stream's call pattern is fixed by the reference's code
    (env_helpers.py:583 initial cur_model_idx; :590-593 one randint per reset env in index order; policy noise per step;
    :619 step_rand indices or :626 mean_std noise), but WHICH envs reset depends on the simulated dones, so the pinned oracle
    replays the run once to supply them.  The pool is handed out sequentially (make_golden.PoolEnv).  Returns (draws, n_steps)."""
    from conftest import ReplayRNG, dm_from_golden, PoolReset
    env, sam_mode = str(d['env']), str(d['sam_mode'])
    dm = dm_from_golden(d, env)
    K, B, H = int(d['K']), int(d['B']), int(d['H'])
    theta, pdims = d['theta'], [int(x) for x in d['pdims']]
    rng = ReplayRNG(d['rng_kinds'], d['rng_offs'], d['rng_flat'])
    log = []                                                       # every value the reference drew, in order

    class Rec(object):
        def randint(self, high, size=None):
            v = rng.randint(high, size=size); log.append(np.array(v)); return v

        def normal(self, size=None):
            v = rng.normal(size=size); log.append(np.array(v)); return v
    rec = Rec()
    ve = O.VecEnvOracle(env, lambda s, a: O.dynamics_forward_all(dm, s, a), K, B, dm.ns, H, sam_mode, PoolReset(d['pool']), rng=rec)
    dones = []
    orig_step = ve.step
    ve.step = lambda a: (lambda out: (dones.append(np.array(out[2])), out)[1])(orig_step(a))
    O.obtain_samples(ve, lambda obs: O.policy_get_actions(theta, pdims, np.asarray(obs), rec.normal(size=(len(obs), pdims[-1]))),
                     int(d['batch_size']), determ=bool(d['determ']))
    assert rng.exhausted()
    T = len(dones)
    Tp = T + T_pad
    n_pool = len(d['pool'])
    dr = dict(eps=np.zeros((Tp, B, dm.na)), model_idx=np.zeros((Tp, B), np.int64), sel_noise=np.zeros((Tp, B, dm.ns)),
              reset_idx=np.zeros((Tp + 1, B), np.int64), reset_model=np.zeros((Tp + 1, B), np.int64))
    it = iter(log)
    next(it)                                                       # env_helpers.py:583 (overwritten by the first reset)
    pool_i = 0
    for b in range(B):
        dr['reset_idx'][0, b] = pool_i % n_pool; pool_i += 1
        dr['reset_model'][0, b] = int(next(it))
    for t in range(T):
        dr['eps'][t] = next(it)
        if sam_mode == 'step_rand':
            dr['model_idx'][t] = next(it)
        elif sam_mode == 'model_mean_std':
            dr['sel_noise'][t] = next(it)
        for b in np.nonzero(dones[t])[0]:
            dr['reset_idx'][t + 1, b] = pool_i % n_pool; pool_i += 1
            dr['reset_model'][t + 1, b] = int(next(it))
    assert next(it, None) is None
    return dr, T
