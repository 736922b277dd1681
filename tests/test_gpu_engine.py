"""GPU parity: every C-ABI kernel entry point vs the CPU oracle (float64) on identical inputs and
identical supplied random draws.  Tolerances: tests/tolerances.py = the table of DESIGN.md section 5 (rows cited at each use)."""
import numpy as np
import pytest
import torch
from conftest import load_golden, dm_from_golden
from oracle import metrpo_oracle as O
import helpers as Hh
import tolerances as TOL

pytestmark = pytest.mark.gpu

STEP_TOL = TOL.STEP                # row 1


def cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize('env,sam_mode', [('swimmer', m) for m in O.SAM_MODES] +
                         [('ant', 'step_rand'), ('half_cheetah', 'model_med'), ('hopper', 'eps_rand'),
                          ('snake', 'model_mean_std'), ('humanoid', 'step_rand')])
def test_step_parity(env, sam_mode):
    K = 5 if env != 'humanoid' else 4
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (64, 64), (32, 32), seed=3)
    rng = np.random.RandomState(0)
    B = 333
    s = pool[rng.randint(len(pool), size=B)] + rng.randn(B, dm.ns) * 0.05
    if env == 'ant':
        s[:, 2] = rng.uniform(0.1, 1.1, size=B)
    a = rng.randn(B, dm.na) * 0.9
    idx = rng.randint(K, size=B)
    noise = rng.randn(B, dm.ns)
    s32 = s.astype(np.float32).astype(np.float64); a32 = a.astype(np.float32).astype(np.float64)
    s_next, rew, done, nall = eng.step(s, a, sam_mode, idx, noise.astype(np.float32), want_all=True)
    ac = np.clip(a32, -1, 1)
    ref_all = O.dynamics_forward_all(dm, s32, ac)
    np.testing.assert_allclose(cpu(nall), ref_all, **STEP_TOL)
    ref_next = O.select_next(ref_all, sam_mode, idx, noise.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(cpu(s_next), ref_next, **STEP_TOL)
    np.testing.assert_allclose(cpu(rew), -O.cost_np_vec(env, s32, ac, ref_next), **TOL.REWARD)
    ref_done = O.is_done(env, ref_next, ref_next)
    near = np.zeros(B, bool)
    if env == 'ant':   # exclude samples within fp32 rounding of the 0.2/1.0 thresholds
        near = (np.abs(ref_next[:, 2] - 0.2) < 1e-5) | (np.abs(ref_next[:, 2] - 1.0) < 1e-5)
        assert ref_done.any() and not ref_done.all()
    assert np.array_equal(cpu(done).astype(bool)[~near], ref_done[~near])


@pytest.mark.parametrize('name', ['vecenv_swimmer_step_rand', 'vecenv_ant_step_rand', 'vecenv_ant_eps_rand', 'vecenv_swimmer_model_med',
                                  'vecenv_swimmer_model_mean_std', 'vecenv_swimmer_eps_rand', 'vecenv_swimmer_one_model',
                                  'vecenv_swimmer_model_mean'])
def test_step_against_reference_golden(name):
    """Teacher-forced replay of the reference's own VecSimpleEnv traces (tests/golden)."""
    import metrpo_amd
    d = load_golden(name)
    env, sam_mode = str(d['env']), str(d['sam_mode'])
    dm = dm_from_golden(d, env)
    hidden = [w.shape[2] for w in dm.Ws[:-1]]
    eng = metrpo_amd.Engine(env, dm.K, hidden, (8, 8))
    eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    T, B = d['actions'].shape[:2]
    state = d['first_obs'].copy()
    # recover the per-step model indices / noise the reference drew from its recorded K-head outputs
    for t in range(T):
        nall_ref = d['next_all'][t]
        s_next, rew, done, nall = eng.step(state, d['actions'][t], 'one_model', None, None, want_all=True)
        np.testing.assert_allclose(cpu(nall), nall_ref, **TOL.STEP)
        # the reference's post-step state for non-reset envs must be one of / a function of the heads we computed
        dn = d['dones'][t]
        got = cpu(nall)
        if sam_mode in ('model_mean', 'model_med', 'one_model'):
            s2, r2, d2 = eng.step(state, d['actions'][t], sam_mode, None, None)
            keep = ~dn
            np.testing.assert_allclose(cpu(s2)[keep], d['states'][t][keep], **TOL.STEP)
            np.testing.assert_allclose(cpu(r2), d['rewards'][t], **TOL.REWARD)
        elif sam_mode in ('step_rand', 'eps_rand'):
            # find which head the reference selected, feed that index, compare reward
            ref_sel = np.array([np.argmin([np.abs(nall_ref[k, b] - (d['states'][t][b] if not dn[b] else nall_ref[k, b])).max()
                                           for k in range(dm.K)]) for b in range(B)])
            s2, r2, d2 = eng.step(state, d['actions'][t], sam_mode, ref_sel, None)
            keep = ~dn
            np.testing.assert_allclose(cpu(s2)[keep], d['states'][t][keep], **TOL.STEP)
            np.testing.assert_allclose(cpu(r2)[keep], d['rewards'][t][keep], **TOL.REWARD)
        state = d['states'][t]          # teacher forcing with the reference's own next state (incl. resets)
        del got


def test_policy_actions_parity():
    eng, dm, theta, pdims, pool = Hh.make_engine('half_cheetah', 3, (64, 64), (32, 32), seed=5)
    rng = np.random.RandomState(1)
    obs = rng.randn(777, dm.ns).astype(np.float32); eps = rng.randn(777, dm.na).astype(np.float32)
    a, m = eng.policy_actions(obs, eps)
    ra, info = O.policy_get_actions(theta.astype(np.float32).astype(np.float64), pdims, obs.astype(np.float64), eps.astype(np.float64))
    np.testing.assert_allclose(cpu(m), info['mean'], **TOL.STEP)
    np.testing.assert_allclose(cpu(a), ra, **TOL.ACTION)
    a2, m2 = eng.policy_actions(obs, None)            # determ=True path: actions = mean
    assert torch.equal(a2, m2) and torch.equal(m2, m)


@pytest.mark.parametrize('variant', ['generic', 'head_per_wave', 'coop', 'coop_two_per_cu'])
@pytest.mark.parametrize('env,sam_mode,determ', [('swimmer', 'step_rand', False), ('swimmer', 'eps_rand', True),
                                                 ('ant', 'step_rand', False), ('swimmer', 'model_mean_std', False),
                                                 ('half_cheetah', 'model_med', False), ('snake', 'model_mean', False),
                                                 ('hopper', 'one_model', False)])
def test_rollout_parity_teacher_forced(env, sam_mode, determ, variant):
    """Fused rollout vs oracle with supplied draws.  Free-running for the discrete structure (dones,
    tpath, resets), teacher-forced (oracle fed the device's own observations) for the per-step values."""
    _rollout_parity_teacher_forced(env, sam_mode, determ, variant, 5)


@pytest.mark.parametrize('variant', ['coop', 'coop_two_per_cu'])
@pytest.mark.parametrize('env,sam_mode,K', [('swimmer', 'step_rand', 1), ('swimmer', 'model_med', 3), ('half_cheetah', 'step_rand', 2), ('hopper', 'model_mean_std', 4),
                                            ('ant', 'step_rand', 3), ('snake', 'eps_rand', 2), ('ant', 'model_mean', 4)])
def test_cooperative_rollout_with_one_to_four_heads(env, sam_mode, K, variant):
    """rollout_coop.hip is instantiated for K = 1 ... 5 heads (every params file has 5; fewer heads used to fall to the head-per-wave kernel): the same
    teacher-forced comparison against the oracle, both launch forms, selection modes that read all heads included."""
    _rollout_parity_teacher_forced(env, sam_mode, False, variant, K)


@pytest.mark.parametrize('env,sam_mode,K', [('swimmer', 'step_rand', 10), ('swimmer', 'model_med', 7), ('half_cheetah', 'step_rand', 8), ('hopper', 'model_mean_std', 6),
                                            ('ant', 'step_rand', 8), ('snake', 'eps_rand', 9), ('ant', 'model_mean', 6), ('half_cheetah', 'model_mean_std', 9), ('hopper', 'step_rand', 10)])
def test_cooperative_rollout_with_six_to_ten_heads(env, sam_mode, K):
    """Round 6 (verdict r5 item 6a): K = 6 ... 10 heads at 2 x 64 on the cooperative kernel's one-workgroup-per-CU instantiation (rollout_coop_k<K>.hip; until then the
    head-per-wave kernel for K <= 8 and step-wise tile GEMMs for 9 / 10).  The same teacher-forced comparison against the oracle, selection modes that read all heads included."""
    _rollout_parity_teacher_forced(env, sam_mode, False, 'coop', K)


@pytest.mark.parametrize('env,sam_mode,K,w', [('swimmer', 'step_rand', 5, 48), ('hopper', 'model_mean_std', 3, 32), ('ant', 'step_rand', 5, 20), ('half_cheetah', 'model_med', 4, 50),
                                              ('snake', 'eps_rand', 7, 63), ('swimmer', 'one_model', 5, 1), ('hopper', 'step_rand', 5, (24, 64)), ('swimmer', 'model_mean', 2, (64, 17))])
def test_narrow_dynamics_nets_run_zero_padded_on_the_cooperative_kernel(env, sam_mode, K, w):
    """Round 6: two hidden layers of at most 64 units (not both 64; unequal widths included) run on the fused cooperative kernel over a zero-padded copy of the weights in its 64 x 64 layout (the padded
    units add exact zeros; until then: step-wise tile GEMMs, 2.7 ms where 64 x 64 takes 0.47 -- tools/width_table.py).  Same teacher-forced comparison against the oracle."""
    _rollout_parity_teacher_forced(env, sam_mode, False, 'coop', K, hidden=w if isinstance(w, tuple) else (w, w))


def test_padded_weights_follow_every_writer_of_the_dynamics():
    """The padded copy is rebuilt in front of every rollout launch: set_dynamics_model (one head replaced) must show in the next rollout -- compared with a fresh context
    that was given the final weights at once (bitwise)."""
    env, K, w, B, T, H = 'swimmer', 4, 40, 96, 5, 5
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (w, w), (32, 32), seed=21)
    pool_t = torch.tensor(pool, dtype=torch.float32, device=eng.device)
    eng.rollout(B, T, H, 'step_rand', pool_t, seed=2)
    assert eng.last_rollout_kernel() == 'mfma-cooperative' and eng.rollout_note() == ''
    other = Hh.make_engine(env, K, (w, w), (32, 32), seed=22)[1]
    eng.set_dynamics_model(2, np.concatenate([np.concatenate([other.Ws[l][2].ravel(), other.bs[l][2].ravel()]) for l in range(3)]).astype(np.float32))
    a = eng.rollout(B, T, H, 'step_rand', pool_t, seed=3)
    Ws = [W.copy() for W in dm.Ws]; bs = [b.copy() for b in dm.bs]
    for l in range(3):
        Ws[l][2] = other.Ws[l][2]; bs[l][2] = other.bs[l][2]
    fresh = Hh.make_engine(env, K, (w, w), (32, 32), seed=21)[0]
    fresh.set_dynamics_layers(Ws, bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    b = fresh.rollout(B, T, H, 'step_rand', pool_t, seed=3)
    assert torch.equal(a.obs, b.obs) and torch.equal(a.rew, b.rew)


def test_cooperative_launch_rule_does_not_pick_the_slow_form():
    """The launch rule of rollout_coop.hip chooses between one workgroup per CU (tiles migrate) and two co-resident workgroups from constants MEASURED at K = 5
    (1.50 / 1.58 / 1.65 per pair-step).  If the kernel changes and the constants do not, the rule goes silently wrong -- so its pick is timed against the forced
    two-per-CU form here (K = 5 and K = 3, below and above two tiles per CU): never more than 15 % behind it (profiles/r05_e_coop_heads.txt: within 4 %)."""
    import metrpo_amd
    from metrpo_amd import synthetic
    for K, B in ((5, 5000), (5, 8192), (3, 5000)):
        ms = {}
        for var in (0, 2):
            eng = metrpo_amd.Engine('swimmer', K, (64, 64), (32, 32))
            Ws, bs, norm = synthetic.make_dynamics('swimmer', K, (64, 64), seed=0)
            eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
            eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
            eng.set_rollout_variant(var)
            pool = torch.as_tensor(synthetic.make_pool('swimmer'), device='cuda')
            out = eng.alloc_trajectory(B, 100, 100)
            for i in range(3):
                eng.rollout(B, 100, 100, 'step_rand', pool, seed=i, out=out)
            best = 1e9
            for rep in range(3):                                   # best of three 10-launch means: a timing assertion must not trip over a noisy neighbour
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(10):
                    eng.rollout(B, 100, 100, 'step_rand', pool, seed=10 + i, out=out)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            ms[var] = best
        assert ms[0] <= 1.15 * ms[2], (K, B, ms)


def test_heads_beyond_the_cooperative_kernels_lds_say_so():
    """Ant holds 8 heads in a CU's LDS, half-cheetah 9: one more lands on the step-wise tile GEMMs and metrpo_rollout_note names the table."""
    for env, K in (('ant', 9), ('half_cheetah', 10)):
        eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (64, 64), (32, 32), seed=9)
        eng.rollout(64, 3, 3, 'step_rand', pool, seed=1)
        assert eng.last_rollout_kernel() == 'gemm-stepwise' and 'beyond the fused kernels' in eng.rollout_note(), (eng.last_rollout_kernel(), eng.rollout_note())


def test_cooperative_rollout_more_than_five_heads_needs_a_cu_per_workgroup():
    """K > 5 exists only as one workgroup per CU.  More tiles than CUs: the migrating schedule on an exclusive device (bitwise the trajectories of a launch that has a
    CU per tile ... the same draws, tile by tile); on a shared device the dispatcher leaves the kernel (head-per-wave for K <= 8) and says so."""
    env, K, T, H = 'swimmer', 7, 6, 4
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (64, 64), (32, 32), seed=9)
    n_cu = torch.cuda.get_device_properties(eng.device).multi_processor_count
    B = 16 * (n_cu + 9)
    pool_t = torch.tensor(pool, dtype=torch.float32, device=eng.device)
    a = eng.rollout(B, T, H, 'step_rand', pool_t, seed=4)
    assert eng.last_rollout_kernel() == 'mfma-cooperative' and eng.rollout_note() == ''
    keep = [x.clone() for x in (a.obs, a.act, a.rew, a.done)]
    # the first n_cu tiles alone (a CU per tile, no migration): Philox streams are per env index -> the same rows
    b = eng.rollout(16 * n_cu, T, H, 'step_rand', pool_t, seed=4)
    for x, y in zip(keep, (b.obs, b.act, b.rew, b.done)):
        assert torch.equal(x[:, :16 * n_cu], y)
    eng.set_exclusive(False)
    c = eng.rollout(B, T, H, 'step_rand', pool_t, seed=4)
    assert eng.last_rollout_kernel() == 'mfma-head-per-wave' and 'CU to itself' in eng.rollout_note()
    np.testing.assert_allclose(cpu(c.obs), cpu(keep[0]), **TOL.CROSS_KERNEL)


def _rollout_parity_teacher_forced(env, sam_mode, determ, variant, K, hidden=(64, 64)):
    B, T, H = 200, 12, 5
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, hidden, (32, 32), seed=7)
    force_generic = variant == 'generic'
    running = eng.set_rollout_variant({'head_per_wave': 1, 'coop_two_per_cu': 2}.get(variant, 0))
    if not force_generic:
        assert running == (1 if variant == 'head_per_wave' else 2)
        eng.rollout(16, 2, 2, sam_mode, pool, seed=1)
        assert eng.last_rollout_kernel() == ('mfma-head-per-wave' if variant == 'head_per_wave' else 'mfma-cooperative')
    if env == 'ant':
        pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    th = theta.astype(np.float32).astype(np.float64)
    dr = Hh.draws(np.random.RandomState(2), K, B, T, dm.ns, dm.na, len(pool))
    dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    traj = eng.rollout(B, T, H, sam_mode, pool, determ=determ, force_generic=force_generic, **dr32)
    torch.cuda.synchronize()
    drf = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in dr32.items()}
    pool32 = pool.astype(np.float32).astype(np.float64)
    ref = Hh.oracle_rollout(dm, th, pdims, env, pool32, drf, B, T, H, sam_mode, determ, teacher_obs=cpu(traj.obs))
    np.testing.assert_allclose(cpu(traj.mean), ref['mean'], **TOL.STEP)
    np.testing.assert_allclose(cpu(traj.act), ref['act'], **TOL.ACTION)
    np.testing.assert_allclose(cpu(traj.rew), ref['rew'], **TOL.REWARD)
    # next state: obs[t+1] == oracle next(t) wherever no reset happened; pool rows where it did
    dn = cpu(traj.done).astype(bool)
    for t in range(T - 1):
        keep = ~dn[t]
        np.testing.assert_allclose(cpu(traj.obs[t + 1])[keep], ref['next'][t][keep], **TOL.STEP)
        np.testing.assert_array_equal(cpu(traj.obs[t + 1])[~keep], pool32[dr['reset_idx'][t + 1]][~keep])
    np.testing.assert_array_equal(cpu(traj.obs[0]), pool32[dr['reset_idx'][0]])
    # discrete structure: horizon resets at exactly H unless terminated early; tpath counts from 0
    free = Hh.oracle_rollout(dm, th, pdims, env, pool32, drf, B, T, H, sam_mode, determ)
    if env != 'ant':
        assert np.array_equal(dn, free['done'])
        assert np.array_equal(cpu(traj.tpath), free['tpath'])
        assert dn[H - 1].all() and not dn[:H - 1].any()
    else:
        agree = (dn == free['done']).mean()
        assert agree > 0.98 and dn.any() and (cpu(traj.tpath) <= H - 1).all()
        assert dn[cpu(traj.tpath) == H - 1].all()


def test_rollout_free_running_short():
    """t <= 10 free-running agreement on the default (fastest available) path."""
    K, B, T, H = 5, 256, 10, 50
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', K, (64, 64), (32, 32), seed=9)
    dr = Hh.draws(np.random.RandomState(3), K, B, T, dm.ns, dm.na, len(pool))
    dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    traj = eng.rollout(B, T, H, 'step_rand', pool, **dr32)
    drf = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in dr32.items()}
    ref = Hh.oracle_rollout(dm, theta.astype(np.float32).astype(np.float64), pdims, 'swimmer',
                            pool.astype(np.float32).astype(np.float64), drf, B, T, H, 'step_rand')
    np.testing.assert_allclose(cpu(traj.obs), ref['obs'], **TOL.FREE_RUN)
    np.testing.assert_allclose(cpu(traj.rew), ref['rew'], **TOL.FREE_RUN)
    np.testing.assert_allclose(cpu(traj.last_obs), ref['last_obs'], **TOL.FREE_RUN)


def test_rollout_philox_statistics():
    """Production RNG: draws are N(0,1) / uniform over K and pool; different seeds differ; same seed repeats."""
    K, B, T, H = 5, 4096, 8, 4
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', K, (64, 64), (32, 32), seed=11)
    t1 = eng.rollout(B, T, H, 'step_rand', pool, seed=123)
    a1 = cpu(t1.act).copy(); o1 = cpu(t1.obs).copy()
    t2 = eng.rollout(B, T, H, 'step_rand', pool, seed=123)
    assert np.array_equal(a1, cpu(t2.act)) and np.array_equal(o1, cpu(t2.obs))
    t3 = eng.rollout(B, T, H, 'step_rand', pool, seed=124)
    assert not np.array_equal(a1, cpu(t3.act))
    z = (a1 - cpu(t1.mean)) / np.exp(theta[-dm.na:].astype(np.float32))
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(np.mean(z ** 3)) < 0.05 and abs(np.mean(z ** 4) - 3.0) < 0.15
    # initial states are pool rows, spread over the pool
    rows = {r.tobytes() for r in pool.astype(np.float32)}
    assert all(r.tobytes() in rows for r in o1[0].astype(np.float32)[:64])
    assert len({tuple(r) for r in o1[0]}) > 0.9 * min(B, len(pool)) * (1 - np.exp(-B / len(pool)))
    # stream_offset shifts the env counter: rank 1's envs == the upper half of a double-sized run
    big = eng.rollout(2 * B, T, H, 'step_rand', pool, seed=7)
    hi = eng.rollout(B, T, H, 'step_rand', pool, seed=7, stream_offset=B)
    assert np.array_equal(cpu(big.act)[:, B:], cpu(hi.act))


@pytest.mark.parametrize('env', ['swimmer', 'half_cheetah', 'hopper'])
def test_rollout_variants_share_rng_stream(env):
    """Production (Philox) mode: the three rollout kernels consume the same counter-based stream, so the same seed
    gives the same trajectory up to fp32 rounding, whichever kernel runs."""
    K, B, T, H = 5, 300, 6, 3
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (64, 64), (32, 32), seed=17)
    ref = eng.rollout(B, T, H, 'step_rand', pool, seed=99, force_generic=True)
    for variant in (1, 0, 2):
        eng.set_rollout_variant(variant)
        got = eng.rollout(B, T, H, 'step_rand', pool, seed=99)
        np.testing.assert_allclose(cpu(got.act), cpu(ref.act), **TOL.CROSS_KERNEL)
        np.testing.assert_allclose(cpu(got.obs), cpu(ref.obs), **TOL.CROSS_KERNEL)
        assert torch.equal(got.done, ref.done) and torch.equal(got.tpath, ref.tpath)
    # the reset rows / models drawn at t = H differ per env and are valid pool rows
    rows = {r.tobytes() for r in pool.astype(np.float32)}
    assert all(r.tobytes() in rows for r in cpu(ref.obs)[H].astype(np.float32)[:32])


def test_gae_statistics_are_summed_in_a_fixed_order():
    """sum(adv), sum(adv^2), count come from 79 workgroups (B = 5000): added in workgroup order by the last one to finish, so thirty launches
    give thirty identical triples whatever order the workgroups finish in, the triple is ACCUMULATED into the caller's buffer (metrpo.h), and a
    launch with another grid right after uses the same ticket."""
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 5, (64, 64), (32, 32), seed=3)
    traj = eng.rollout(5000, 40, 20, 'step_rand', pool, seed=11)
    coeffs = np.random.RandomState(2).randn(2 * dm.ns + 4) * 0.05
    first = None
    for rep in range(30):
        adv, ret, valid, stats = eng.gae(traj, coeffs, 0.99, 0.95)
        got = (cpu(stats).tobytes(), cpu(adv).tobytes())
        first = first or got
        assert got == first
    s1 = np.frombuffer(first[0], dtype=np.float64)
    assert s1[2] == 5000 * 40 and abs(s1[0] - cpu(adv).astype(np.float64).sum()) <= 1e-6 * max(1.0, abs(s1[0]))
    acc = torch.full((3,), 2.0, dtype=torch.float64, device='cuda')
    eng.gae(traj, coeffs, 0.99, 0.95, stats=acc)
    np.testing.assert_array_equal(cpu(acc), s1 + 2.0)
    small = eng.rollout(70, 12, 6, 'step_rand', pool, seed=12)                   # 2 workgroups after 79
    a2, r2, v2, st2 = eng.gae(small, coeffs, 0.99, 0.95)
    assert cpu(st2)[2] == 70 * 12 and abs(cpu(st2)[0] - cpu(a2).astype(np.float64).sum()) <= 1e-9 * max(1.0, abs(cpu(st2)[0]))
    adv, ret, valid, stats = eng.gae(traj, coeffs, 0.99, 0.95)
    assert cpu(stats).tobytes() == first[0]


@pytest.mark.parametrize('gamma,lam,use_coeffs', [(1.0, 1.0, False), (0.99, 0.95, True), (0.99, 1.0, True)])
def test_gae_center_gram_parity(gamma, lam, use_coeffs):
    env, K, B, T, H = 'ant', 4, 96, 23, 6
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (64, 64), (32, 32), seed=13)
    pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
    eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    dr = Hh.draws(np.random.RandomState(4), K, B, T, dm.ns, dm.na, len(pool))
    traj = eng.rollout(B, T, H, 'step_rand', pool, **{k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()})
    coeffs = np.random.RandomState(5).randn(2 * dm.ns + 4) * 0.05 if use_coeffs else None
    adv, ret, valid, stats = eng.gae(traj, coeffs, gamma, lam)
    tr = dict(obs=cpu(traj.obs), act=cpu(traj.act), rew=cpu(traj.rew), mean=cpu(traj.mean),
              done=cpu(traj.done).astype(bool), tpath=cpu(traj.tpath))
    paths = Hh.paths_from_timemajor(tr)
    assert len({len(p['rewards']) for p in paths}) > 1            # ragged
    base = O.LinearFeatureBaselineOracle()
    base._coeffs = coeffs
    samples = O.process_samples(paths, base, gamma, lam, center_adv=False)
    tb = [x for p in paths for x in p['_tb']]
    tt, bb = np.array(tb).T
    v = cpu(valid).astype(bool)
    assert v.sum() == len(tb) and v[tt, bb].all()                 # trailing unfinished paths are masked out
    np.testing.assert_allclose(cpu(ret)[tt, bb], samples['returns'], **TOL.RETURNS)
    np.testing.assert_allclose(cpu(adv)[tt, bb], samples['advantages'], **TOL.ADVANTAGE)
    a = samples['advantages']
    np.testing.assert_allclose(cpu(stats), [a.sum(), (a * a).sum(), len(a)], rtol=1e-5, atol=1e-4)
    eng.center_advantages(adv, valid, stats)
    np.testing.assert_allclose(cpu(adv)[tt, bb], O.center_advantages(a), **TOL.ADVANTAGE_CENTRED)
    assert (cpu(adv)[~v] == 0).all()
    # baseline normal equations
    AtA, Aty = eng.baseline_gram(traj.obs, ret, traj.tpath, valid)
    F = np.concatenate([O.LinearFeatureBaselineOracle.features(p) for p in paths])
    # products are formed in f32 (16-sample MFMA tiles) and summed in f64: compare at f32-product accuracy,
    # scaled by the diagonal (entries with cancellation are small relative to sqrt(G_ii G_jj))
    G = F.T @ F
    scale = np.sqrt(np.outer(np.diag(G), np.diag(G)))
    assert (np.abs(cpu(AtA) - G) <= TOL.NORMAL_EQ * scale + 1e-9).all()
    y = samples['returns']
    assert (np.abs(cpu(Aty) - F.T @ y) <= TOL.NORMAL_EQ * np.sqrt(np.diag(G) * (y @ y)) + 1e-6).all()
    # and the quantity that matters: the fitted coefficients agree with the float64 normal equations
    import metrpo_amd
    bl = metrpo_amd.LinearFeatureBaseline()
    got_c = bl.solve(cpu(AtA), cpu(Aty))
    ref_c = O.LinearFeatureBaselineOracle(); ref_c.fit(paths)
    pred_got, pred_ref = F @ got_c, F @ ref_c._coeffs
    assert np.abs(pred_got - pred_ref).max() <= TOL.BASELINE_FIT * max(1.0, np.abs(pred_ref).max())


@pytest.mark.parametrize('env', ['swimmer', 'hopper', 'snake', 'half_cheetah', 'ant'])
def test_gram_kernel_on_ragged_sample_counts(env):
    """k_gram_mfma (process.hip): a wave stages 64 consecutive samples per trip through LDS (coalesced loads) and runs their four 16-sample MFMA tiles.  Sample
    counts around the 16 / 64 boundaries, a random valid mask, observations beyond the +-10 clip: against the float64 normal equations of the reference's feature
    map ([rllab] LinearFeatureBaseline._features: [o, o^2, t/100, (t/100)^2, (t/100)^3, 1], o = clip(obs, -10, 10)) at f32-product accuracy; accumulating calls add."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 2, (64, 64), (32, 32), seed=5)
    ns, dev = dm.ns, eng.device
    F = 2 * ns + 4
    rng = np.random.RandomState(77)
    for N in (1, 15, 16, 63, 64, 65, 1000, 4099, 70001):
        obs = (rng.randn(N, ns) * 4.0).astype(np.float32)
        obs[rng.rand(N, ns) < 0.02] *= 5.0                          # some entries beyond the clip
        ret = rng.randn(N).astype(np.float32) * 3.0
        tpath = rng.randint(0, 200, size=N).astype(np.int32)
        valid = (rng.rand(N) < 0.8).astype(np.uint8)
        o = np.clip(obs.astype(np.float64), -10.0, 10.0)
        al = (tpath.astype(np.float32) / np.float32(100.0)).astype(np.float64)[:, None]
        Fm = np.concatenate([o, o * o, al, al ** 2, al ** 3, np.ones((N, 1))], axis=1) * valid[:, None]
        G, b = Fm.T @ Fm, Fm.T @ (ret.astype(np.float64) * valid)
        out = torch.zeros(F * F + F, dtype=torch.float64, device=dev)
        args = [torch.as_tensor(x, device=dev) for x in (obs, ret, tpath, valid)]
        eng.baseline_gram(*args, out=out)
        AtA, Aty = cpu(out[:F * F]).reshape(F, F), cpu(out[F * F:])
        scale = np.sqrt(np.outer(np.diag(G), np.diag(G)))
        assert (np.abs(AtA - G) <= TOL.NORMAL_EQ * scale + 1e-9).all(), (env, N)
        assert (np.abs(Aty - b) <= TOL.NORMAL_EQ * np.sqrt(np.diag(G) * float((ret.astype(np.float64) * valid) @ (ret.astype(np.float64) * valid))) + 1e-6).all(), (env, N)
        eng.baseline_gram(*args, out=out)                           # ACCUMULATES (metrpo.h)
        np.testing.assert_allclose(cpu(out[:F * F]).reshape(F, F), 2.0 * AtA, rtol=1e-12, atol=1e-12)
        again = torch.zeros_like(out); eng.baseline_gram(*args, out=again)
        assert torch.equal(again[:F * F], torch.as_tensor(AtA.reshape(-1), device=dev))      # bitwise repeatable


def _update_problem(env='swimmer', N=5000, seed=21, pol_hidden=(32, 32)):
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 2, (64, 64), pol_hidden, seed=seed)
    rng = np.random.RandomState(seed)
    th = theta.astype(np.float32).astype(np.float64)
    obs = (rng.randn(N, dm.ns) * 0.5).astype(np.float32).astype(np.float64)
    old_mean = O.policy_mean(th, pdims, obs).astype(np.float32).astype(np.float64)
    old_ls = np.broadcast_to(O.policy_log_std(th, pdims), old_mean.shape).copy()
    act = (old_mean + np.exp(old_ls) * rng.randn(*old_mean.shape)).astype(np.float32).astype(np.float64)
    adv = O.center_advantages(rng.randn(N)).astype(np.float32).astype(np.float64)
    return eng, th, pdims, obs, act, adv, old_mean, old_ls


_REL_L2_SEEN = {}


def rel_l2(a, b):
    v = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    import os
    if os.environ.get('METRPO_TOL_REPORT'):                     # worst value per call site, next to the assert_allclose report (conftest.py)
        import traceback, json
        fr = traceback.extract_stack()[-2]
        key = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
        _REL_L2_SEEN[key] = max(_REL_L2_SEEN.get(key, 0.0), float(v))
        json.dump(_REL_L2_SEEN, open(os.environ['METRPO_TOL_REPORT'] + '.rel_l2', 'w'), indent=1, sort_keys=True)
    return v


@pytest.mark.parametrize('use_mfma', [True, False, 'gemm'])
@pytest.mark.parametrize('env,pol_hidden,N', [('swimmer', (32, 32), 5000), ('half_cheetah', (32, 32), 3001),
                                              ('ant', (32, 32), 2000), ('hopper', (32, 32), 777), ('snake', (32, 32), 1024),
                                              ('humanoid', (100, 50, 25), 1500), ('humanoid', (100, 50, 25), 20011), ('swimmer', (17,), 3000)])
def test_loss_grad_fvp_losskl_parity(env, pol_hidden, N, use_mfma):
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(env, N, pol_hidden=pol_hidden)
    active = eng.set_update_path(use_mfma)
    if use_mfma is True and not active:
        pytest.skip('no fused MFMA update kernels for this policy shape (generic / GEMM paths cover it)')
    if use_mfma == 'gemm':
        assert active == 'gemm'
    valid = np.ones(N, np.uint8); valid[::7] = 0
    keep = valid.astype(bool)
    batch = eng.make_batch(obs, act, adv, om, ols, valid=valid)
    out = cpu(eng.loss_grad(batch))
    loss, g = O.surrogate_loss_grad(th, pdims, obs[keep], act[keep], adv[keep], om[keep], ols[keep])
    assert abs(out[0] - loss) <= TOL.LOSS_RTOL * max(1.0, abs(loss))
    assert rel_l2(out[1:], g) <= TOL.GRAD_REL_L2                   # row 10
    v = np.random.RandomState(1).randn(eng.P)
    hv = cpu(eng.fvp(batch, v))
    ref = O.fisher_vector_product(th, pdims, obs[keep], v, reg_coeff=0.0)
    assert rel_l2(hv, ref) <= TOL.FVP_REL_L2
    # trial theta away from theta_old (lr != 1, kl > 0) -- also the broadcast log_std form (stride 0)
    th2 = (th + np.random.RandomState(2).randn(th.size) * 0.02).astype(np.float32)
    batch0 = eng.make_batch(obs, act, adv, om, ols[0], valid=valid)
    for bt in (batch, batch0):
        lk = cpu(eng.loss_kl(bt, th2))
        l2, k2 = O.surrogate_loss_kl(th2.astype(np.float64), pdims, obs[keep], act[keep], adv[keep], om[keep], ols[keep])
        assert abs(lk[0] - l2) <= TOL.LOSS_RTOL * max(1.0, abs(l2)) and abs(lk[1] - k2) <= max(TOL.KL_ATOL, TOL.KL_RTOL * k2)
    lk0 = cpu(eng.loss_kl(batch))                                 # at theta_old: lr = 1, kl = 0
    assert abs(lk0[1]) < TOL.KL_ATOL and abs(lk0[0] - loss) < 1e-6


def test_policy_kernels_many_tiles_per_wave_and_both_log_std_forms():
    """N large enough that every wave of the MFMA kernels walks through several rounds of its 7 : 6 tile deal (policy_mfma.hip POL_SPLIT_R) and
    ends on a partial tile; gradient / FVP / loss-KL against the oracle, and the broadcast (stride 0, hoisted exponentials) and per-sample
    forms of the old log_std bit for bit."""
    N = 300007
    eng, th, pdims, obs, act, adv, om, ols = _update_problem('swimmer', N)
    assert eng.set_update_path(True)
    rows, bcast = eng.make_batch(obs, act, adv, om, ols), eng.make_batch(obs, act, adv, om, ols[0])
    out = eng.loss_grad(rows).clone()
    assert torch.equal(out, eng.loss_grad(bcast))
    loss, g = O.surrogate_loss_grad(th, pdims, obs, act, adv, om, ols)
    out = cpu(out)
    assert abs(out[0] - loss) <= TOL.LOSS_RTOL * max(1.0, abs(loss)) and rel_l2(out[1:], g) <= TOL.GRAD_REL_L2
    v = np.random.RandomState(1).randn(eng.P)
    assert rel_l2(cpu(eng.fvp(rows, v)), O.fisher_vector_product(th, pdims, obs, v, reg_coeff=0.0)) <= TOL.FVP_REL_L2
    th2 = (th + np.random.RandomState(2).randn(th.size) * 0.02).astype(np.float32)
    lk = eng.loss_kl(rows, th2).clone()
    assert torch.equal(lk, eng.loss_kl(bcast, th2))
    l2, k2 = O.surrogate_loss_kl(th2.astype(np.float64), pdims, obs, act, adv, om, ols)
    lk = cpu(lk)
    assert abs(lk[0] - l2) <= TOL.LOSS_RTOL * max(1.0, abs(l2)) and abs(lk[1] - k2) <= max(TOL.KL_ATOL, TOL.KL_RTOL * k2)


def test_update_is_bitwise_reproducible():
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=4000)
    batch = eng.make_batch(obs, act, adv, om, ols)
    a = eng.loss_grad(batch).clone(); b = eng.loss_grad(batch).clone()
    assert torch.equal(a, b)
    v = np.random.RandomState(1).randn(eng.P)
    assert torch.equal(eng.fvp(batch, v).clone(), eng.fvp(batch, v).clone())


@pytest.mark.parametrize('use_mfma', [True, False, 'gemm'])
@pytest.mark.parametrize('seed', [21, 22])
def test_trpo_update_parity(seed, use_mfma):
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=6000, seed=seed)
    assert eng.set_update_path(use_mfma) == use_mfma
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
    assert abs(out['loss_before'] - ref['loss_before']) < 1e-6
    # post-update KL/loss are evaluated at slightly different theta_new (d matches to rel-L2 1e-3): relative 5e-3
    assert abs(out['kl'] - ref['kl']) <= TOL.POST_UPDATE_RTOL * ref['kl'] and out['kl'] <= 0.01 and out['loss'] < out['loss_before']
    assert abs(out['loss'] - ref['loss']) <= TOL.POST_UPDATE_RTOL * abs(ref['loss'])
    # theta_new = theta - ratio * beta * d inherits d's tolerance (rel-L2 <= 1e-3 after 10 CG iterations)
    step_ref = ref['theta_new'] - th
    np.testing.assert_allclose(cpu(eng.get_policy()), ref['theta_new'], rtol=0, atol=TOL.CG_REL_L2 * np.linalg.norm(step_ref) + 1e-6)
    assert rel_l2(cpu(eng.get_policy()) - th.astype(np.float32).astype(np.float64), step_ref) <= TOL.THETA_STEP_REL_L2


@pytest.mark.parametrize('use_mfma', [True, False, 'gemm'])
def test_step_scale_from_cg_recurrence_equals_explicit_hvp(use_mfma):
    """[rllab] optimize() evaluates f_Hx(descent_direction) once more for the step scale; the default here takes d.(H d) from
    krylov.cg's own recurrence (H d = g - r).  Both routes must give the same beta / step / accepted theta (float32 FVP rounding)."""
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=20000, seed=23)
    assert eng.set_update_path(use_mfma) == use_mfma
    batch = eng.make_batch(obs, act, adv, om, ols)
    theta0 = eng.get_policy().clone()
    a = eng.trpo_update(batch, max_kl=0.01, want_vectors=True, explicit_final_hvp=False)
    ta = eng.get_policy().clone()
    eng.set_policy(theta0)
    b = eng.trpo_update(batch, max_kl=0.01, want_vectors=True, explicit_final_hvp=True)
    tb = eng.get_policy().clone()
    assert torch.equal(a['d'], b['d']) and torch.equal(a['g'], b['g'])          # identical CG trajectory
    assert abs(a['beta'] - b['beta']) <= 2e-6 * b['beta']
    assert a['n_backtrack'] == b['n_backtrack'] and a['accepted'] and b['accepted']
    np.testing.assert_allclose(cpu(ta), cpu(tb), rtol=0, atol=1e-7)
    ref = O.cg_optimize(th, pdims, obs, act, adv, om, ols, max_kl=0.01)          # the oracle takes rllab's explicit route
    assert abs(a['beta'] - ref['beta']) <= TOL.STEP_SCALE_RTOL * ref['beta']


def test_trpo_update_rejects_and_restores():
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=2000)
    before = eng.get_policy().clone()
    out = eng.trpo_update(eng.make_batch(obs, act, adv * 0.0, om, ols))       # zero advantage -> nan step -> rejected
    assert not out['accepted'] and torch.equal(before, eng.get_policy())


@pytest.mark.parametrize('env', ['swimmer', 'ant', 'half_cheetah'])
def test_validation_cost_parity(env):
    K, Bv, T, gamma = 4, 500, 15, 0.97
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (64, 64), (32, 32), seed=31, n_pool=512)
    if env == 'ant':
        pool[::3, 2] = 0.25; dm.diff_mean[2] = -0.01
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    s0 = pool[:Bv].astype(np.float32)
    got = cpu(eng.validation_cost(s0, T, gamma))
    ref = O.validation_costs(dm, theta.astype(np.float32).astype(np.float64), pdims, env, s0.astype(np.float64), T, gamma)
    np.testing.assert_allclose(got, ref, **TOL.VALIDATION_COST)


def test_error_behaviour():
    import metrpo_amd
    from metrpo_amd._lib import MetrpoError
    eng = metrpo_amd.Engine('swimmer', 3, (16, 16), (8, 8))
    with pytest.raises(MetrpoError, match='not been called'):
        eng.step(np.zeros((4, 10), np.float32), np.zeros((4, 2), np.float32), 'one_model')
    with pytest.raises(KeyError):
        eng.step(np.zeros((4, 10), np.float32), np.zeros((4, 2), np.float32), 'no_such_mode')
    with pytest.raises(Exception):
        metrpo_amd.Engine('swimmer', 0, (16,), (8,))
    # empty batch is a no-op, not an error
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 3, (16, 16), (8, 8))
    s, r, d = eng.step(np.zeros((0, 10), np.float32), np.zeros((0, 2), np.float32), 'one_model')
    assert s.shape == (0, 10)
    with pytest.raises(MetrpoError, match='model_idx'):
        eng.step(np.zeros((4, 10), np.float32), np.zeros((4, 2), np.float32), 'step_rand')


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs as parity cases (shapes of C0..C4 at batch sizes the oracle finishes in seconds)
CONFIG_SHAPES = [
    ('C0-baseline-json', 'swimmer', 5, (64, 64), (32, 32), 100, 50, 50),        # the reference's CPU-runnable case, full size
    ('C0-params-file', 'swimmer', 5, (512, 512), (32, 32), 100, 8, 8),          # params-swimmer.json:20-23 network
    ('C2-2x1024', 'half_cheetah', 5, (1024, 1024), (32, 32), 48, 6, 6),         # params-half-cheetah.json:20-21 network
    ('C3', 'ant', 10, (512, 512), (32, 32), 64, 6, 4),                          # K=10 ensemble, early termination
    ('C4', 'humanoid', 20, (1024, 1024, 1024), (100, 50, 25), 24, 4, 4),        # K=20, 3x1024, policy 100-50-25
]


@pytest.mark.parametrize('name,env,K,dh,ph,B,T,H', CONFIG_SHAPES)
def test_baseline_config_shapes_end_to_end(name, env, K, dh, ph, B, T, H):
    """rollout (whatever kernel the shape selects) -> GAE -> TRPO update, each stage against the oracle."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, dh, ph, seed=51)
    if env == 'ant':
        pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    th = theta.astype(np.float32).astype(np.float64)
    pool32 = pool.astype(np.float32).astype(np.float64)
    dr = Hh.draws(np.random.RandomState(6), K, B, T, dm.ns, dm.na, len(pool))
    dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    traj = eng.rollout(B, T, H, 'step_rand', pool, **dr32)
    expected_path = 2 if (max(dh) <= 64 and K == 5) else (3 if min(dh) >= 128 else None)
    if expected_path is not None:
        assert eng.set_rollout_variant(0) == expected_path
    drf = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in dr32.items()}
    ref = Hh.oracle_rollout(dm, th, pdims, env, pool32, drf, B, T, H, 'step_rand', teacher_obs=cpu(traj.obs))
    tol = TOL.wide_or_step(dh)                                  # rows 1 / 2
    if expected_path == 3:      # the step-wise GEMM path and the generic kernel agree (same draws, short horizon)
        gen = eng.rollout(B, T, H, 'step_rand', pool, force_generic=True, **dr32)
        np.testing.assert_allclose(cpu(traj.obs), cpu(gen.obs), **TOL.CROSS_KERNEL)
        assert torch.equal(traj.tpath, gen.tpath)
    np.testing.assert_allclose(cpu(traj.mean), ref['mean'], **tol)
    np.testing.assert_allclose(cpu(traj.rew), ref['rew'], **tol)
    dn = cpu(traj.done).astype(bool)
    for t in range(T - 1):
        np.testing.assert_allclose(cpu(traj.obs[t + 1])[~dn[t]], ref['next'][t][~dn[t]], **tol)
    adv, ret, valid, stats = eng.gae(traj, None, 1.0, 1.0)
    eng.center_advantages(adv, valid, stats)
    v = cpu(valid).astype(bool)
    assert v.any()
    batch = eng.make_batch(traj.obs, traj.act, adv, traj.mean, eng.get_policy()[-dm.na:], valid=valid)
    keep = v.reshape(-1)
    obs, act = cpu(traj.obs).reshape(-1, dm.ns)[keep], cpu(traj.act).reshape(-1, dm.na)[keep]
    a, om = cpu(adv).reshape(-1)[keep], cpu(traj.mean).reshape(-1, dm.na)[keep]
    ols = np.broadcast_to(O.policy_log_std(th, pdims), om.shape).copy()
    out = cpu(eng.loss_grad(batch))
    loss, g = O.surrogate_loss_grad(th, pdims, obs, act, a, om, ols)
    assert rel_l2(out[1:], g) <= TOL.GRAD_REL_L2 and abs(out[0] - loss) <= TOL.LOSS_RTOL * max(1.0, abs(loss))
    vv = np.random.RandomState(2).randn(eng.P)
    assert rel_l2(cpu(eng.fvp(batch, vv)), O.fisher_vector_product(th, pdims, obs, vv, reg_coeff=0.0)) <= TOL.FVP_REL_L2
    res = eng.trpo_update(batch)
    assert np.isfinite(res['loss_before']) and (not res['accepted'] or (res['kl'] <= 0.01 and res['loss'] < res['loss_before']))


@pytest.mark.parametrize('sam_mode', list(O.SAM_MODES))
def test_gemm_rollout_all_sam_modes(sam_mode):
    """Large-net (step-wise GEMM) path: every sam_mode, supplied draws, vs the oracle teacher-forced."""
    env, K, B, T, H = 'half_cheetah', 4, 70, 7, 3
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (128, 256), (32, 32), seed=61)
    assert eng.set_rollout_variant(0) == 3
    th = theta.astype(np.float32).astype(np.float64)
    pool32 = pool.astype(np.float32).astype(np.float64)
    dr = Hh.draws(np.random.RandomState(8), K, B, T, dm.ns, dm.na, len(pool))
    dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    traj = eng.rollout(B, T, H, sam_mode, pool, **dr32)
    drf = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in dr32.items()}
    ref = Hh.oracle_rollout(dm, th, pdims, env, pool32, drf, B, T, H, sam_mode, teacher_obs=cpu(traj.obs))
    np.testing.assert_allclose(cpu(traj.act), ref['act'], **TOL.WIDE)
    np.testing.assert_allclose(cpu(traj.rew), ref['rew'], **TOL.WIDE)
    dn = cpu(traj.done).astype(bool)
    assert dn[H - 1].all() and not dn[:H - 1].any()
    for t in range(T - 1):
        np.testing.assert_allclose(cpu(traj.obs[t + 1])[~dn[t]], ref['next'][t][~dn[t]], **TOL.WIDE)
        np.testing.assert_array_equal(cpu(traj.obs[t + 1])[dn[t]], pool32[dr['reset_idx'][t + 1]][dn[t]])
    np.testing.assert_allclose(cpu(traj.last_obs), np.where(dn[T - 1][:, None], pool32[dr['reset_idx'][T]], ref['next'][T - 1]), **TOL.WIDE)


@pytest.mark.parametrize('draws', [True, False])
def test_gemm_rollout_wide_policy_gemm_prestep(draws, monkeypatch):
    """Humanoid-shaped policy (100-50-25: no MFMA pre-kernel) on the step-wise GEMM rollout: the pre-step that runs the policy layers as
    GEMMs over the batch (large B in production, forced here) vs the oracle, and vs the 64-env-block pre-kernel it replaces there."""
    env, K, B, T, H = 'humanoid', 3, 90, 6, 4
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (128, 128), (100, 50, 25), seed=23)
    assert eng.set_rollout_variant(0) == 3
    kw = {}
    if draws:
        dr = Hh.draws(np.random.RandomState(5), K, B, T, dm.ns, dm.na, len(pool))
        kw = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    eng.set_option('METRPO_PRE_GEMM', '0')
    blocks = eng.rollout(B, T, H, 'step_rand', pool, seed=3, **kw)
    ref = {k: cpu(getattr(blocks, k)).copy() for k in ('obs', 'act', 'mean', 'rew')}
    ref_done = blocks.done.clone()
    eng.set_option('METRPO_PRE_GEMM', '1')
    gemm = eng.rollout(B, T, H, 'step_rand', pool, seed=3, **kw)
    for k in ref:
        np.testing.assert_allclose(cpu(getattr(gemm, k)), ref[k], **TOL.CROSS_KERNEL, err_msg=k)
    assert torch.equal(gemm.done, ref_done)
    if draws:
        th = theta.astype(np.float32).astype(np.float64)
        pool32 = pool.astype(np.float32).astype(np.float64)
        drf = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in kw.items()}
        orc = Hh.oracle_rollout(dm, th, pdims, env, pool32, drf, B, T, H, 'step_rand', teacher_obs=cpu(gemm.obs))
        np.testing.assert_allclose(cpu(gemm.mean), orc['mean'], **TOL.WIDE)
        np.testing.assert_allclose(cpu(gemm.act), orc['act'], **TOL.WIDE)
        np.testing.assert_allclose(cpu(gemm.rew), orc['rew'], **TOL.WIDE)


@pytest.mark.parametrize('env,K,B,hidden', [('humanoid', 16, 4096, (512, 512)),      # 128x128 tiles, 45 output columns (3 column tiles)
                                            ('ant', 4, 300, (128, 256))])            # 64x64 tiles, 29 output columns
def test_gemm_rollout_fused_output_layer_equals_separate_layers(env, K, B, hidden, monkeypatch):
    """Step-wise GEMM rollout: the last hidden layer's launch contracts its relu tile with the output weights (EPI_RELU_OUT) instead of
    writing the activations and running the output layer as its own GEMM.  Same trajectories up to fp32 summation order."""
    T, H = 4, 3
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, hidden, (32, 32) if env != 'humanoid' else (100, 50, 25), seed=29)
    assert eng.set_rollout_variant(0) == 3
    eng.set_option('METRPO_NO_FUSED_OUT', '1')
    sep = eng.rollout(B, T, H, 'model_mean', pool, seed=9)
    ref = {k: cpu(getattr(sep, k)).copy() for k in ('obs', 'act', 'rew')}
    ref_done = sep.done.clone()
    eng.set_option('METRPO_NO_FUSED_OUT', None)
    fused = eng.rollout(B, T, H, 'model_mean', pool, seed=9)
    for k in ref:
        got = cpu(getattr(fused, k))
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, ref[k], **TOL.CROSS_KERNEL, err_msg=k)
    assert torch.equal(fused.done, ref_done)
    assert float(np.abs(ref['obs'][1] - ref['obs'][0]).max()) > 1e-3                  # the dynamics did move the state


def test_wide_feature_baseline_gram_humanoid():
    """114 features + the return column = 8 column blocks: the block-cooperative MFMA Gram kernel (process.hip k_gram_mfma_wide) vs float64 NumPy
    on the oracle's feature matrix, at a sample count that is not a multiple of the tile or the grid."""
    env, K, B, T, H = 'humanoid', 3, 333, 9, 4
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (64, 64), (32, 32), seed=19)
    traj = eng.rollout(B, T, H, 'step_rand', pool, seed=4)
    adv, ret, valid, stats = eng.gae(traj, None, 0.99, 0.95)
    AtA, Aty = eng.baseline_gram(traj.obs, ret, traj.tpath, valid)
    tr = dict(obs=cpu(traj.obs), act=cpu(traj.act), rew=cpu(traj.rew), mean=cpu(traj.mean), done=cpu(traj.done).astype(bool), tpath=cpu(traj.tpath))
    paths = Hh.paths_from_timemajor(tr)
    F = np.concatenate([O.LinearFeatureBaselineOracle.features(p) for p in paths])
    tb = np.array([x for p in paths for x in p['_tb']])
    y = cpu(ret)[tb[:, 0], tb[:, 1]]
    nf = 2 * dm.ns + 4
    assert F.shape[1] == nf == 114 and len(F) == int(cpu(valid).sum())
    G = F.T @ F
    scale = np.sqrt(np.outer(np.diag(G), np.diag(G)))
    assert (np.abs(cpu(AtA) - G) <= TOL.NORMAL_EQ * scale + 1e-9).all()
    np.testing.assert_allclose(cpu(AtA), cpu(AtA).T, rtol=0, atol=0)               # mirrored blocks
    assert (np.abs(cpu(Aty) - F.T @ y) <= TOL.NORMAL_EQ * np.sqrt(np.diag(G) * (y @ y)) + 1e-6).all()
    # accumulates (+=) into the caller's buffers like the narrow kernels: a second call doubles the sums
    out = torch.cat([AtA.reshape(-1), Aty]).clone()
    eng.baseline_gram(traj.obs, ret, traj.tpath, valid, out=out)
    np.testing.assert_allclose(cpu(out[:nf * nf]).reshape(nf, nf), 2 * cpu(AtA), rtol=1e-12)


def test_gemm_rollout_random_shapes_vs_generic_kernel():
    """Random (env, K, hidden widths, B, sam_mode) on the step-wise GEMM rollout -- widths that are and are not multiples of 64 (fused vs separate
    output layer), 2 and 3 hidden layers, MFMA / block / GEMM pre-step -- against the thread-per-env generic kernel with the same supplied draws."""
    rs = np.random.RandomState(1234)
    envs = ['swimmer', 'half_cheetah', 'ant', 'hopper', 'snake', 'humanoid']
    for case in range(14):
        env = envs[case % len(envs)]
        K = int(rs.randint(1, 6))
        nl = 2 if case % 3 else 3
        hidden = tuple(int(rs.choice([128, 160, 192, 256, 320])) for _ in range(nl))
        if case >= 8:                                                   # widths below 128 (round 5: no longer the thread-per-env kernel's), not multiples of 4 included
            hidden = tuple(int(rs.choice([16, 48, 65, 72, 96, 100, 127])) for _ in range(nl))
        B = int(rs.randint(20, 400)); H = int(rs.randint(2, 6)); T = int(rs.randint(2, 2 * H + 1))
        mode = ['step_rand', 'eps_rand', 'model_mean', 'model_med', 'one_model'][int(rs.randint(5))]
        pol = (100, 50, 25) if env == 'humanoid' else (32, 32)
        eng, dm, theta, pdims, pool = Hh.make_engine(env, K, hidden, pol, seed=300 + case)
        assert eng.set_rollout_variant(0) == 3
        dr = Hh.draws(np.random.RandomState(case), K, B, T, dm.ns, dm.na, len(pool))
        dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
        dr32.pop('sel_noise', None)
        got = eng.rollout(B, T, H, mode, pool, **dr32)
        assert eng.last_rollout_kernel() in ('gemm-stepwise', 'gemm-streamk', 'resident'), (case, hidden, eng.last_rollout_kernel())
        ref = eng.rollout(B, T, H, mode, pool, force_generic=True, **dr32)
        msg = str((case, env, K, hidden, B, T, H, mode))
        assert torch.equal(got.tpath, ref.tpath) and torch.equal(got.done, ref.done), msg
        np.testing.assert_allclose(cpu(got.mean), cpu(ref.mean), **TOL.CROSS_KERNEL, err_msg=msg)
        np.testing.assert_allclose(cpu(got.obs), cpu(ref.obs), **TOL.CROSS_KERNEL, err_msg=msg)
        np.testing.assert_allclose(cpu(got.rew), cpu(ref.rew), **TOL.CROSS_KERNEL, err_msg=msg)


@pytest.mark.parametrize('merged', [True, False])
@pytest.mark.parametrize('env,K,hidden,B,H,R,mode,pol', [('swimmer', 5, (512, 512), 100, 6, 3, 'step_rand', (32, 32)), ('half_cheetah', 3, (256, 192), 77, 4, 5, 'eps_rand', (32, 32)),
                                                          ('hopper', 2, (128, 128, 128), 50, 5, 2, 'model_mean_std', (32, 32)), ('swimmer', 5, (512, 512), 100, 3, 8, 'model_med', (32, 32)),
                                                          ('humanoid', 3, (256, 256), 37, 4, 5, 'step_rand', (100, 50, 25)),     # wide policy: GEMM pre-step, tiles spanning two rounds
                                                          ('snake', 2, (128, 128), 20, 3, 4, 'eps_rand', (24, 24))])           # thread-per-env pre-step
def test_gemm_rollout_concurrent_rounds_equal_sequential_rounds(env, K, hidden, B, H, R, mode, pol, merged, monkeypatch):
    """Small-batch rollouts of horizon-terminated envs run their T / H rounds side by side (reset states of the later rounds computed from the
    draws of the step before them, rollout_gemm.hip): as ONE merged batch of R B envs stepping H times (R B <= 1024), or one stream per round
    (METRPO_NO_MERGED_ROUNDS=1: what larger batches take) -- bit for bit the sequential step loop, production draws."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, hidden, pol, seed=77)
    assert eng.set_rollout_variant(1) == 3                          # 1: stay on the step-wise path where the resident kernel would take over (test_gpu_resident.py)
    if not merged:
        eng.set_option('METRPO_NO_MERGED_ROUNDS', '1')
    T = R * H
    par = eng.rollout(B, T, H, mode, pool, seed=5)
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    par = [x.clone() for x in (par.obs, par.act, par.mean, par.rew, par.done, par.tpath, par.last_obs)]
    eng.set_option('METRPO_SEQ_ROUNDS', '1')
    seq = eng.rollout(B, T, H, mode, pool, seed=5)
    for a, b in zip(par, (seq.obs, seq.act, seq.mean, seq.rew, seq.done, seq.tpath, seq.last_obs)):
        assert torch.equal(a, b)
    assert int(seq.done.sum()) == R * B and bool(seq.done[H - 1::H].all())


@pytest.mark.parametrize('env,B,T', [('swimmer', 70, 13), ('half_cheetah', 130, 9), ('hopper', 64, 17), ('snake', 33, 8), ('humanoid', 65, 11), ('ant', 200, 5)])
def test_gae_fused_baseline_predict_every_state_width(env, B, T):
    """baseline.predict (samplers/base.py:55) is evaluated inside the GAE kernel from coalesced, LDS-transposed observation blocks
    (k_gae<ns>, one instantiation per env): every state width, partial 64-env blocks, horizons that are not a multiple of the 8 time
    chunks -- advantages / returns against the oracle's per-path scan with the same coefficients."""
    from metrpo_amd.engine import Trajectory
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 2, (64, 64), (32, 32) if env != 'humanoid' else (100, 50, 25), seed=3)
    rs = np.random.RandomState(B + T)
    H = 4
    obs = rs.randn(T, B, dm.ns).astype(np.float32) * 3.0           # some entries beyond the +-10 clip of the features
    obs[rs.rand(T, B, dm.ns) < 0.02] *= 5.0
    rew = rs.randn(T, B).astype(np.float32)
    tpath = np.zeros((T, B), np.int32); done = np.zeros((T, B), np.uint8)
    ts = np.zeros(B, np.int32)
    for t in range(T):
        tpath[t] = ts
        ts += 1
        dn = (ts >= H) | (rs.rand(B) < 0.1)
        done[t] = dn
        ts[dn] = 0
    dev = eng.device
    traj = Trajectory(torch.as_tensor(obs, device=dev), torch.zeros(T, B, dm.na, device=dev), torch.as_tensor(rew, device=dev),
                      torch.zeros(T, B, dm.na, device=dev), torch.as_tensor(done, device=dev), torch.as_tensor(tpath, device=dev),
                      torch.zeros(B, dm.ns, device=dev), B, T, H)
    coeffs = rs.randn(2 * dm.ns + 4) * 0.05
    adv, ret, valid, stats = eng.gae(traj, coeffs, 0.99, 0.95)
    tr = dict(obs=obs, act=np.zeros((T, B, dm.na), np.float32), rew=rew, mean=np.zeros((T, B, dm.na), np.float32), done=done.astype(bool), tpath=tpath)
    paths = Hh.paths_from_timemajor(tr)
    base = O.LinearFeatureBaselineOracle(); base._coeffs = coeffs
    samples = O.process_samples(paths, base, 0.99, 0.95, center_adv=False)
    tt, bb = np.array([x for p in paths for x in p['_tb']]).T
    np.testing.assert_allclose(cpu(ret)[tt, bb], samples['returns'], **TOL.RETURNS)
    np.testing.assert_allclose(cpu(adv)[tt, bb], samples['advantages'], **TOL.ADVANTAGE)
    adv2 = eng.gae(traj, coeffs, 0.99, 0.95)[0]
    assert torch.equal(adv, adv2)                                  # bitwise repeatable


def test_phase_timers_and_trace_ranges():
    """the reference accounts policy / env / process time in VectorizedSampler.obtain_samples (vectorized_sampler.py:54-56,106)
    and policy_opt_time in the outer loop (model_based_rl.py:694).  Here: HIP-event timers on the algo (no synchronisation in the loop) and
    roctx ranges inside libmetrpo.so (exercised by every entry point; must be harmless without a profiler attached)."""
    import metrpo_amd
    from metrpo_amd import synthetic
    env, K, B, H = 'swimmer', 5, 256, 20
    eng = metrpo_amd.Engine(env, K, (64, 64), (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, K, (64, 64), seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    policy = metrpo_amd.GaussianMLPPolicy(eng, init_std=1.0, seed=0)
    nne = metrpo_amd.NeuralNetEnv(env=metrpo_amd.InitStatePool(synthetic.make_pool(env), 2), inner_env=None, cost_np=env, dynamics_in=None,
                                  dynamics_outs=eng, sam_mode='step_rand')
    algo = metrpo_amd.TRPO(env=nne, policy=policy, baseline=metrpo_amd.LinearFeatureBaseline(), batch_size=B * H, max_path_length=H,
                           discount=0.99, step_size=0.01, sampler_args=dict(n_envs=B))
    assert algo.timers.summary() == {'rollout_ms': None, 'process_ms': None, 'policy_opt_ms': None, 'n': 0}     # off by default
    algo.timers.enable()
    for j in range(4):
        algo.start_worker()
        paths = algo.obtain_samples(j)
        algo.optimize_policy(j, algo.process_samples(j, paths))
    s = algo.timers.summary()
    assert s['n'] == 4 and all(0.0 < s[k] < 1000.0 for k in ('rollout_ms', 'process_ms', 'policy_opt_ms')), s
    assert len(algo.timers.times_ms('rollout')) == 4


@pytest.mark.parametrize('use_mfma', [True, False, 'gemm'])
def test_trpo_update_in_two_halves_equals_one_call(use_mfma):
    """metrpo_trpo_update_begin / _end (line-search accept test on the device for the first S trials, no host synchronisation in the first
    half) against metrpo_trpo_update: theta, loss, KL, trial index and acceptance bit for bit, for S below, at and above the accepted trial
    ('late' = the policy changed inside the second half); the GEMM path falls back to the synchronous update inside the first half."""
    import metrpo_amd
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=6000, seed=29)
    eng.set_update_path(use_mfma)
    for scale in (1.0, 40.0):                                   # different step sizes -> different numbers of backtracks
        batch = eng.make_batch(obs, act, adv * scale, om, ols)
        eng.set_policy(th)
        ref = eng.trpo_update(batch, want_vectors=True)
        th_ref = eng.get_policy().clone()
        for S in (1, 2, 4):
            eng.set_policy(th)
            assert eng.trpo_update(batch, want_vectors=True, spec_trials=S) is None
            with pytest.raises(metrpo_amd._lib.MetrpoError, match='still open'):
                eng.trpo_update(batch)
            out = eng.trpo_update_end()
            assert torch.equal(eng.get_policy(), th_ref), (scale, S)
            for key in ('loss_before', 'loss', 'kl', 'beta', 'n_backtrack', 'accepted', 'cg_iters_run'):
                assert out[key] == ref[key], (key, scale, S)
            assert torch.equal(out['g'], ref['g']) and torch.equal(out['d'], ref['d'])
            assert out['late'] == (use_mfma != 'gemm' and ref['accepted'] and ref['n_backtrack'] >= S)
    with pytest.raises(metrpo_amd._lib.MetrpoError, match='no update is open'):
        eng._upd_open = (batch, None, None, None, 1); eng.trpo_update_end()


@pytest.mark.parametrize('explicit', [False, True])
def test_trial_thetas_built_in_the_tails_equal_the_launches(explicit):
    """Device-decided line search: theta of trial 0 comes out of the tail that finishes the step size (CG recurrence or, explicit_final_hvp, the extra
    product), theta of trial n + 1 out of trial n's accept test (cg_device.h CgTail::nx_*) -- against the synchronous update, whose trial points are
    k_try_theta launches: same policy, same trial index, bit for bit, for searches that stop at trial 0, in the middle and behind the speculated trials."""
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=6000, seed=29)
    seen = set()
    for scale in (1.0, 40.0, 400.0):
        batch = eng.make_batch(obs, act, adv * scale, om, ols)
        eng.set_policy(th)
        ref = eng.trpo_update(batch, explicit_final_hvp=explicit)
        th_ref = eng.get_policy().clone()
        seen.add(ref['n_backtrack'])
        for S in (1, 3, 6):
            eng.set_policy(th)
            assert eng.trpo_update(batch, explicit_final_hvp=explicit, spec_trials=S) is None
            out = eng.trpo_update_end()
            assert torch.equal(eng.get_policy(), th_ref), (scale, S)
            for key in ('loss_before', 'loss', 'kl', 'beta', 'n_backtrack', 'accepted', 'cg_iters_run'):
                assert out[key] == ref[key], (key, scale, S)
    assert max(seen) >= 1, seen                                  # at least one search goes past trial 0 (theta of trial n + 1 built by trial n's accept test)


def test_process_begin_is_log_std_clamp_plus_zero_fill():
    """metrpo_process_begin: [rllab] GaussianMLPPolicy's clamped log_std of the ctx policy (what policy.log_std() = clamp(get_policy()[-na:]) returns) and
    the zeroed accumulators, one launch; a log_std below log 1e-6 and a NaN behave as torch.clamp does."""
    import math
    eng, dm, theta, pdims, pool = Hh.make_engine('half_cheetah', 2, (64, 64), (32, 32), seed=3)
    na = dm.na
    th = np.asarray(theta, dtype=np.float32).copy()
    th[-na] = -15.0; th[-na + 1] = -0.3; th[-na + 2] = float('nan')
    eng.set_policy(th)
    want = torch.clamp(eng.get_policy()[-na:], min=float(np.log(1e-6)))
    ls, acc = eng.process_begin(3 + 40 * 40 + 40)
    assert ls.dtype == torch.float32 and acc.dtype == torch.float64 and acc.numel() == 1643
    assert torch.equal(torch.nan_to_num(ls, nan=7.0), torch.nan_to_num(want, nan=7.0)) and math.isnan(float(ls[2]))
    assert float(ls[0]) == float(np.float32(np.log(1e-6)))
    assert torch.count_nonzero(acc).item() == 0
    acc.fill_(3.0)
    ls2, acc2 = eng.process_begin(5)
    assert acc2.numel() == 5 and torch.count_nonzero(acc2).item() == 0


def test_deferred_optimizer_closes_after_the_next_rollout():
    """algos.async_line_search: optimize_policy only enqueues the update; the next obtain_samples enqueues its rollout and THEN closes it.
    Three iterations give the policy and the trajectories of the synchronous order (same seeds), whatever the trial the search stops at."""
    from test_gpu_api import build_algo
    outs = []
    for mode in (False, True):
        algo, eng, dm, theta, pdims, pool = build_algo('swimmer', B=128, H=20, gamma=1.0, lam=1.0)
        algo.async_line_search = mode
        algo.optimizer.spec_trials = 1                           # forces the late path (and the repeated rollout) whenever trial 0 is rejected
        acts = []
        for j in range(3):
            algo.start_worker()
            paths = algo.obtain_samples(j)
            acts.append(paths.traj.act.clone())
            samples = algo.process_samples(j, paths)
            algo.optimize_policy(j, samples)
        assert algo.optimizer.pending == mode
        if mode:                                                 # an open update owns theta: the C entry point refuses to replace it ...
            import ctypes
            from metrpo_amd import _lib as L
            tdev = torch.as_tensor(theta, dtype=torch.float32, device='cuda')
            rc = L.lib.metrpo_set_policy(eng._ctx, ctypes.c_void_p(tdev.data_ptr()), eng._stream())
            assert rc != 0 and 'still open' in L.lib.metrpo_last_error(eng._ctx).decode()
            th_now = eng.get_policy()                            # ... and the host class closes the update before it reads (or replaces) the policy
            assert getattr(eng, '_upd_open', None) is None
        d = algo.optimizer.last_diag
        assert not algo.optimizer.pending and d['accepted']
        if mode:
            assert torch.equal(th_now, eng.get_policy())
        outs.append((eng.get_policy().clone(), acts))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('env', ['swimmer', 'half_cheetah', 'humanoid'])
def test_baseline_solve_on_device_matches_lstsq(env):
    """metrpo_baseline_solve (LinearFeatureBaseline.fit's solve as a kernel: float64 elimination with partial pivoting, rllab's x10
    regularisation on a NaN) against the host route the reference takes (np.linalg.lstsq of the same normal equations): predicted values of
    the fitted samples agree; the rank-deficient case every real run starts from (all paths of one length: the time features repeat) included."""
    from metrpo_amd.baseline import LinearFeatureBaseline
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 2, (64, 64), (32, 32) if env != 'humanoid' else (100, 50, 25), seed=5)
    rs = np.random.RandomState(11)
    T, B, H = 24, 40, 8
    F = 2 * dm.ns + 4
    obs = (rs.randn(T, B, dm.ns) * 2.0).astype(np.float32)
    ret = rs.randn(T, B).astype(np.float32) * 3.0 + obs[..., 0]
    tpath = np.tile((np.arange(T) % H)[:, None], (1, B)).astype(np.int32)
    valid = np.ones((T, B), np.uint8)
    dev = eng.device
    out = torch.zeros(F * F + F, dtype=torch.float64, device=dev)
    eng.baseline_gram(torch.as_tensor(obs, device=dev), torch.as_tensor(ret, device=dev), torch.as_tensor(tpath, device=dev), torch.as_tensor(valid, device=dev), out=out)
    g = cpu(out)
    AtA, Aty = g[:F * F].reshape(F, F), g[F * F:]
    host = LinearFeatureBaseline(); host.solve(AtA, Aty)
    devb = LinearFeatureBaseline(); devb.solve_device(eng, out)
    assert devb.coeffs_for_kernel.is_cuda and devb.coeffs.shape == (F,)
    o = np.clip(obs.reshape(-1, dm.ns).astype(np.float64), -10, 10); al = tpath.reshape(-1, 1) / 100.0
    feat = np.concatenate([o, o ** 2, al, al ** 2, al ** 3, np.ones_like(al)], axis=1)
    ph, pd_ = feat @ host.coeffs, feat @ devb.coeffs
    np.testing.assert_allclose(pd_, ph, rtol=0, atol=1e-6 * max(1.0, np.abs(ph).max()))
    # residual of the regularised normal equations themselves
    A = AtA + 1e-5 * np.eye(F)
    assert np.linalg.norm(A @ devb.coeffs - Aty) <= 1e-8 * max(1.0, np.linalg.norm(Aty)) + 1e-6 * np.linalg.norm(A @ host.coeffs - Aty)


def test_device_baseline_fit_pipeline_equals_host_fit():
    """algo.device_baseline_fit: three iterations of obtain / process / optimize with the fit's solve on the device against the host solve:
    same advantages (to the conditioning of the 24 x 24 system) and the same policy trajectory."""
    from test_gpu_api import build_algo
    res = []
    for dev_fit in (False, True):
        algo, eng, dm, theta, pdims, pool = build_algo('swimmer', B=128, H=20, gamma=0.99, lam=0.95)
        algo.device_baseline_fit = dev_fit
        advs = []
        for j in range(3):
            algo.start_worker()
            paths = algo.obtain_samples(j)
            samples = algo.process_samples(j, paths)
            advs.append(cpu(samples['advantages']))
            algo.optimize_policy(j, samples)
        res.append((advs, cpu(eng.get_policy()), np.asarray(algo.baseline.coeffs, dtype=np.float64)))
    for a, b in zip(res[0][0], res[1][0]):
        np.testing.assert_allclose(b, a, rtol=0, atol=TOL.ADVANTAGE_CENTRED['atol'] * max(1.0, np.abs(a).max()))
    np.testing.assert_allclose(res[1][1], res[0][1], rtol=0, atol=TOL.MULTI_RANK_THETA * max(1e-3, np.abs(res[0][1] - cpu(torch.as_tensor(theta, dtype=torch.float32))).max()))
