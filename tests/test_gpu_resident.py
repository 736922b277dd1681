"""Resident rollout kernel (rollout_resident.hip): the whole time loop of a small-batch rollout of a 2 x 512 dynamics ensemble in ONE
launch -- the shape of the reference's own params-swimmer.json (K = 5, B = 100, three rounds of 200 steps).  Reference path:
samplers/vectorized_sampler.py:45-116, env_helpers.py:597-635, training.py:218-269.  Checked against the CPU oracle (supplied draws,
teacher-forced), against the step-wise GEMM path (production draws) and against itself across its launch shapes."""
import numpy as np
import pytest
import torch

import helpers as Hh
import tolerances as TOL

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


def _fields(tr):
    return [x.clone() for x in (tr.obs, tr.act, tr.mean, tr.rew, tr.done, tr.tpath, tr.last_obs)]


@pytest.mark.parametrize('hid', [512, 1024])
@pytest.mark.parametrize('env,K,B,T,H,mode', [('swimmer', 5, 100, 12, 12, 'step_rand'),       # params-swimmer.json network and batch
                                               ('swimmer', 5, 128, 9, 4, 'eps_rand'),         # 8 full env tiles, resets inside the call
                                               ('swimmer', 3, 5, 7, 7, 'one_model'),          # one partial tile
                                               ('hopper', 5, 37, 8, 8, 'step_rand'),
                                               ('snake', 4, 50, 8, 5, 'eps_rand'),
                                               ('half_cheetah', 5, 77, 6, 6, 'step_rand'),    # ns = 18: two output-dim tiles, 6 input k-steps
                                               ('swimmer', 5, 100, 12, 4, 'step_rand'),       # T = 3 H: the three rounds side by side, every reset from the supplied draws
                                               ('hopper', 3, 1, 3, 1, 'eps_rand'),            # one env, every step ends an episode
                                               ('ant', 5, 100, 10, 6, 'step_rand')])          # params-ant.json network: episodes also end on the state reached
def test_resident_rollout_against_oracle(env, K, B, T, H, mode, hid):
    """hid = 1024: the 4-wave workgroups of resident_compute_wide (params-half-cheetah / -hopper / -snake.json networks)."""
    if hid == 1024 and (env, B) in (('swimmer', 128), ('hopper', 1)):
        pytest.skip('covered at 512')
    if hid == 512 and env == 'ant':
        pytest.skip('no 2 x 512 Ant in the kernel table (no params file has one)')
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (hid, hid), (32, 32), seed=61)
    if env == 'ant':                                                 # some envs start near the lower z bound: state-dependent dones inside the call
        pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    th = theta.astype(np.float32).astype(np.float64)
    pool32 = pool.astype(np.float32).astype(np.float64)
    dr = Hh.draws(np.random.RandomState(8), K, B, T, dm.ns, dm.na, len(pool))
    dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    traj = eng.rollout(B, T, H, mode, pool, **dr32)
    assert eng.last_rollout_kernel() == 'resident'
    drf = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in dr32.items()}
    ref = Hh.oracle_rollout(dm, th, pdims, env, pool32, drf, B, T, H, mode, teacher_obs=cpu(traj.obs))
    tol = TOL.WIDE                                                   # row 2: 512- / 1024-wide fp32 sums in a fixed, different order
    np.testing.assert_allclose(cpu(traj.mean), ref['mean'], **tol)
    np.testing.assert_allclose(cpu(traj.act), ref['act'], **tol)
    np.testing.assert_allclose(cpu(traj.rew), ref['rew'], **tol)
    dn = cpu(traj.done).astype(bool)
    assert np.array_equal(dn, ref['done'].astype(bool)) and np.array_equal(cpu(traj.tpath), ref['tpath'])
    for t in range(T - 1):
        np.testing.assert_allclose(cpu(traj.obs[t + 1])[~dn[t]], ref['next'][t][~dn[t]], **tol)
        if dn[t].any():                                              # reset rows come straight from the pool
            np.testing.assert_array_equal(cpu(traj.obs[t + 1])[dn[t]], pool.astype(np.float32)[dr['reset_idx'][t + 1][dn[t]]])
    # and the step-wise GEMM path on the same draws (its own summation order)
    eng.set_rollout_variant(1)
    stepwise = eng.rollout(B, T, H, mode, pool, **dr32)
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    generic = eng.rollout(B, T, H, mode, pool, force_generic=True, **dr32)        # thread-per-env kernel: a third, independent summation order
    np.testing.assert_allclose(cpu(traj.obs), cpu(generic.obs), **TOL.CROSS_KERNEL, err_msg='resident kernel vs generic kernel')
    dd = np.abs(cpu(stepwise.obs) - cpu(generic.obs)); badi = np.argwhere(dd > 2e-3)
    info = ''
    if len(badi):
        t0 = badi[:, 0].min(); e_bad = sorted(set(badi[:, 1]))
        info = ' first bad step %d, envs %s, act diff at t0-1 %.3g, mean diff %.3g, model_idx at t0-1 %s, done at t0-1 %s' % (
            t0, e_bad, np.abs(cpu(stepwise.act)[t0 - 1] - cpu(generic.act)[t0 - 1]).max(), np.abs(cpu(stepwise.mean)[t0 - 1] - cpu(generic.mean)[t0 - 1]).max(),
            dr['model_idx'][t0 - 1][e_bad].tolist(), cpu(stepwise.done)[t0 - 1][e_bad].tolist())
    np.testing.assert_allclose(cpu(stepwise.obs), cpu(generic.obs), **TOL.CROSS_KERNEL, err_msg='step-wise GEMM path vs generic kernel' + info)
    assert torch.equal(traj.done, stepwise.done) and torch.equal(traj.tpath, stepwise.tpath)


def test_stepwise_workspace_after_freed_resident_regions():
    """Engines come and go (every one allocates and frees the resident kernel's packet region), each followed by a step-wise rollout on a
    freshly allocated workspace: all three kernels agree every time.  (With the packet region in hipDeviceMallocUncached memory a later
    engine's workspace could land on a freed region and read two stale cache lines of it: envs 32-36 of 37 off by 0.1 from step 1 on.)"""
    import gc
    for rep in range(6):
        eng, dm, theta, pdims, pool = Hh.make_engine('hopper', 5, (512, 512), (32, 32), seed=61 + rep)
        B, T, H = 37 + 3 * rep, 8, 8
        dr = Hh.draws(np.random.RandomState(8 + rep), 5, B, T, dm.ns, dm.na, len(pool))
        dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
        res = eng.rollout(B, T, H, 'step_rand', pool, **dr32)
        assert eng.last_rollout_kernel() == 'resident'
        eng.set_rollout_variant(1)
        stepwise = eng.rollout(B, T, H, 'step_rand', pool, **dr32)
        assert eng.last_rollout_kernel() == 'gemm-stepwise'
        generic = eng.rollout(B, T, H, 'step_rand', pool, force_generic=True, **dr32)
        np.testing.assert_allclose(cpu(res.obs), cpu(generic.obs), **TOL.CROSS_KERNEL)
        np.testing.assert_allclose(cpu(stepwise.obs), cpu(generic.obs), **TOL.CROSS_KERNEL)
        del eng, res, stepwise, generic
        gc.collect()


@pytest.mark.parametrize('env,B,T,H,mode', [('ant', 100, 23, 9, 'step_rand'),            # 7 tiles on 3 columns: 3 / 2 / 2 -> every column 3, 2, 2, 3, ... (Ant's chunks)
                                             ('half_cheetah', 100, 14, 14, 'eps_rand'),    # one round of a horizon-terminated env
                                             ('hopper', 70, 11, 4, 'one_model'),           # 5 tiles on 3 columns, heads nobody selects
                                             ('snake', 16 * 4, 9, 9, 'step_rand')])        # 4 tiles on 3 columns
def test_rotating_tile_deal_and_sentinel_wait_are_bitwise_the_fixed_deal(env, B, T, H, mode, monkeypatch):
    """Launches of few env tiles that do not divide by the workgroup columns (4-wave form): tile g is served by column (g + step) mod columns, and the
    post waves wait on one packet per slice before they read a tile's partial sums (rollout_resident.hip: ResidentK::rot, ::sentinel).  Neither
    changes who adds what in which order: every trajectory tensor bit for bit the fixed deal's, with and without the sentinel wait."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 5, (1024, 1024), (32, 32), seed=67)
    if env == 'ant':
        pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    rot = _fields(eng.rollout(B, T, H, mode, pool, seed=11))
    assert eng.last_rollout_kernel() == 'resident'
    for plan in ('nosentinel', 'nosentinel,norotate'):      # first without the sentinel wait, then also on the fixed deal
        eng.set_option('RESIDENT_PLAN', plan)
        other = eng.rollout(B, T, H, mode, pool, seed=11)
        assert eng.last_rollout_kernel() == 'resident'
        for a, b in zip(rot, _fields(other)):
            assert torch.equal(a, b), var
    eng.comm_check()


@pytest.mark.parametrize('K,B,H,R,ws', [(5, 100, 7, 3, 32),        # params-file layout: 240 compute workgroups, rounds side by side
                                        (5, 100, 3, 8, 32),        # more rounds than fit at once: round groups (3 + 3 + 2), one launch each
                                        (2, 40, 5, 2, 16),         # narrow slices with the rounds side by side
                                        (5, 100, 4, 5, 64)])       # 2 x 1024 nets (4-wave workgroups), params-half-cheetah.json layout: five rounds as 3 + 2
def test_resident_rounds_side_by_side_equal_sequential_rounds(K, B, H, R, ws, monkeypatch):
    """Production draws: the R rounds of a horizon-terminated rollout run side by side (every round from the reset its predecessor's last
    step draws) -- bit for bit the single-round-at-a-time loop at the same slice width (a lone round would otherwise take the narrower
    slices: other partial sums), and within float32 summation order of the step-wise GEMM path."""
    hid = 1024 if ws == 64 else 512
    eng, dm, theta, pdims, pool = Hh.make_engine('half_cheetah' if ws == 64 else 'swimmer', K, (hid, hid), (32, 32), seed=77)
    T = R * H
    eng.set_option('METRPO_RESIDENT_WS', str(ws))
    par = eng.rollout(B, T, H, 'step_rand', pool, seed=5)
    assert eng.last_rollout_kernel() == 'resident'
    par = _fields(par)
    eng.set_option('METRPO_SEQ_ROUNDS', '1')
    seq = eng.rollout(B, T, H, 'step_rand', pool, seed=5)
    assert eng.last_rollout_kernel() == 'resident'
    for a, b in zip(par, _fields(seq)):
        assert torch.equal(a, b)
    assert int(seq.done.sum()) == R * B and bool(seq.done[H - 1::H].all())
    eng.set_rollout_variant(1)
    gm = eng.rollout(B, T, H, 'step_rand', pool, seed=5)
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    assert torch.equal(par[4], gm.done) and torch.equal(par[5], gm.tpath)
    np.testing.assert_array_equal(cpu(par[0][::H]), cpu(gm.obs[::H]))                     # reset states: same pool rows
    np.testing.assert_allclose(cpu(par[0]), cpu(gm.obs), **TOL.CROSS_KERNEL)
    np.testing.assert_allclose(cpu(par[3]), cpu(gm.rew), **TOL.CROSS_KERNEL)


@pytest.mark.parametrize('hid', [512, 1024])
def test_resident_random_shapes_vs_generic_kernel(hid):
    """Random (env, K, B, T, H, mode) with production draws: the resident kernel (8-wave form at 2 x 512, 4-wave form at 2 x 1024, whatever deal
    of rounds / tile columns the launcher picks) against the thread-per-env generic kernel on the same Philox streams -- same dones and
    path-time indices, states within float32 summation order over the few steps before the trajectories' chaotic drift matters."""
    rng = np.random.RandomState(1234 + hid)
    envs = ['swimmer', 'hopper', 'snake', 'half_cheetah'] + (['ant'] if hid == 1024 else [])
    for case in range(10):
        env = envs[rng.randint(len(envs))]
        K = int(rng.randint(1, 6))
        B = int(rng.choice([1, 15, 16, 17, 33, 64, 100, 127, 128]))
        H = int(rng.randint(1, 7))
        R = int(rng.randint(1, 7)) if env != 'ant' else 1
        T = R * H if rng.rand() < 0.7 else R * H + int(rng.randint(1, 3))      # sometimes a ragged tail: resets inside one launch, no round split
        mode = ['step_rand', 'eps_rand', 'one_model'][rng.randint(3)]
        eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (hid, hid), (32, 32), seed=500 + case)
        res = eng.rollout(B, T, H, mode, pool, seed=case)
        assert eng.last_rollout_kernel() == 'resident', (env, K, B, T, H, mode)
        gen = eng.rollout(B, T, H, mode, pool, seed=case, force_generic=True)
        msg = str((env, K, B, T, H, mode))
        np.testing.assert_array_equal(cpu(res.obs[0]), cpu(gen.obs[0]), err_msg=msg)
        if env != 'ant':                                                    # Ant's dones depend on the state: compare them where the states still agree
            assert torch.equal(res.done, gen.done) and torch.equal(res.tpath, gen.tpath), msg
        tcmp = min(T, 4)
        same = np.ones(B, bool)
        for t in range(1, tcmp):
            same &= (cpu(res.done[t - 1]) == cpu(gen.done[t - 1]))
            np.testing.assert_allclose(cpu(res.obs[t])[same], cpu(gen.obs[t])[same], **TOL.CROSS_KERNEL, err_msg=msg + ' step %d' % t)
        np.testing.assert_allclose(cpu(res.act[:1]), cpu(gen.act[:1]), **TOL.WIDE, err_msg=msg)
        del eng


def test_resident_deterministic_policy_and_repeatability():
    """determ=True follows the policy mean (no noise draws); two launches with the same seed are bit-identical at the params-file size (600 steps:
    the packets of the second launch carry later stamps on the same slots), a different seed is a different trajectory."""
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 5, (512, 512), (32, 32), seed=21)
    B, H, R = 100, 200, 3
    a = _fields(eng.rollout(B, R * H, H, 'step_rand', pool, seed=4))
    assert eng.last_rollout_kernel() == 'resident'
    b = _fields(eng.rollout(B, R * H, H, 'step_rand', pool, seed=4))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    c = _fields(eng.rollout(B, R * H, H, 'step_rand', pool, seed=5))
    assert not torch.equal(a[1], c[1])
    assert bool(torch.isfinite(a[0]).all()) and bool(torch.isfinite(a[3]).all())
    assert int(a[4].sum()) == R * B and bool(a[4][H - 1::H].all()) and int(a[5].max()) == H - 1
    d = eng.rollout(B, 8, 8, 'step_rand', pool, seed=4, determ=True)
    assert eng.last_rollout_kernel() == 'resident'
    assert torch.equal(d.act, d.mean)
    np.testing.assert_array_equal(cpu(d.obs[0]), cpu(a[0][0]))                               # same reset draws as the noisy rollout of that seed


@pytest.mark.parametrize('env,hid', [('swimmer', 512), ('ant', 1024)])
def test_resident_chunked_continuation_equals_one_call(env, hid):
    """A rollout cut into chunks (t0, resume, last_state: the sampler's step-granular stop rule) gives the trajectory of the single call.
    Ant (2 x 1024, the 4-wave workgroups): the env the chunks exist for -- its episodes end on the state, the sampler stops step-granular."""
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 5, (hid, hid), (32, 32), seed=13)
    if env == 'ant':
        pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    B, T, H = 70, 12, 5
    whole = eng.rollout(B, T, H, 'eps_rand', pool, seed=9)
    assert eng.last_rollout_kernel() == 'resident'
    whole = _fields(whole)
    dev = eng.device
    lts, lmd = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    parts, resume, t0 = [], None, 0
    stop = torch.zeros(1, dtype=torch.int32, device=dev)
    for Tc in (5, 4, 3):
        tr = eng.rollout(B, Tc, H, 'eps_rand', pool, seed=9, t0=t0, resume=resume, last_state=(lts, lmd), stop=stop)
        assert eng.last_rollout_kernel() == 'resident'
        parts.append(_fields(tr))
        resume = (tr.last_obs.clone(), lts.clone(), lmd.clone())
        t0 += Tc
    for i in range(6):
        assert torch.equal(torch.cat([p[i] for p in parts], 0), whole[i])
    assert torch.equal(parts[-1][6], whole[6])
    if env == 'ant':
        dn = whole[4].bool(); tp = whole[5]
        assert bool((dn & (tp < H - 1)).any())                        # some episodes ended before the horizon
    # a raised stop flag turns the launch into a no-op
    stop.fill_(1)
    sentinel = eng.alloc_trajectory(B, 3, H)
    sentinel.rew.fill_(-7.0)
    eng.rollout(B, 3, H, 'eps_rand', pool, seed=9, t0=t0, resume=resume, last_state=(lts, lmd), stop=stop, out=sentinel)
    assert bool((sentinel.rew == -7.0).all())


def test_resident_kernel_scope():
    """Shapes and modes outside the kernel's table stay on the step-wise GEMM path."""
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 5, (512, 512), (32, 32), seed=3)
    eng.rollout(129, 4, 4, 'step_rand', pool, seed=1)
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    eng.rollout(64, 4, 4, 'model_med', pool, seed=1)
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    eng.rollout(64, 4, 4, 'step_rand', pool, seed=1)
    assert eng.last_rollout_kernel() == 'resident'
    eng2 = Hh.make_engine('swimmer', 5, (256, 256), (32, 32), seed=3)[0]
    eng2.rollout(64, 4, 4, 'step_rand', pool, seed=1)
    assert eng2.last_rollout_kernel() == 'gemm-stepwise'
    eng3 = Hh.make_engine('swimmer', 5, (64, 64), (32, 32), seed=3)[0]
    eng3.rollout(64, 4, 4, 'step_rand', pool, seed=1)
    assert eng3.last_rollout_kernel() == 'mfma-cooperative'
    # 2 x 1024 (every other params file of the reference with a 2 x 32 policy): the 4-wave workgroups
    eng4, _, _, _, pool4 = Hh.make_engine('hopper', 5, (1024, 1024), (32, 32), seed=3)
    eng4.rollout(100, 10, 5, 'step_rand', pool4, seed=1)
    assert eng4.last_rollout_kernel() == 'resident'
    eng4.rollout(100, 4, 4, 'model_mean_std', pool4, seed=1)
    assert eng4.last_rollout_kernel() == 'gemm-stepwise'
    eng5, _, _, _, pool5 = Hh.make_engine('ant', 5, (1024, 1024), (32, 32), seed=3)
    eng5.rollout(100, 4, 4, 'step_rand', pool5, seed=1)
    assert eng5.last_rollout_kernel() == 'resident'                      # Ant too (one round at a time: its episodes end on the state)
    eng6, _, _, _, pool6 = Hh.make_engine('ant', 5, (512, 512), (32, 32), seed=3)
    eng6.rollout(100, 4, 4, 'step_rand', pool6, seed=1)
    assert eng6.last_rollout_kernel() == 'gemm-stepwise'


@pytest.mark.parametrize('env,hid', [('swimmer', 512), ('hopper', 1024)])
def test_resident_missing_workgroup_times_out_and_context_recovers(env, hid, monkeypatch):
    """A workgroup of the grid that never runs (here: told to leave at once; in the field: another process holding CUs) must not hang the GPU:
    the waiting waves give up after 2 s, the launch ends, the next check of the context reports it, and from then on the context rolls out on the
    step-wise path -- with results that match a context that never tried."""
    import time
    import metrpo_amd
    eng, dm, theta, pdims, pool = Hh.make_engine(env, 5, (hid, hid), (32, 32), seed=33)
    B, T, H = 64, 6, 6
    eng.set_option('METRPO_RESIDENT_TEST_SKIP', '3')
    t0 = time.time()
    eng.rollout(B, T, H, 'step_rand', pool, seed=2)
    assert eng.last_rollout_kernel() == 'resident'
    with pytest.raises(metrpo_amd._lib.MetrpoError, match="resident kernel's hand-over timed out"):
        eng.comm_check()
    assert 1.5 < time.time() - t0 < 20.0
    eng.set_option('METRPO_RESIDENT_TEST_SKIP', None)
    again = eng.rollout(B, T, H, 'step_rand', pool, seed=2)
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    eng.comm_check()                                                     # the error cell was cleared: the context is usable
    ref_eng = Hh.make_engine(env, 5, (hid, hid), (32, 32), seed=33)[0]
    ref_eng.set_rollout_variant(1)
    ref = ref_eng.rollout(B, T, H, 'step_rand', pool, seed=2)
    for a, b in zip(_fields(again), _fields(ref)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('env,hid,K,Bv,T,gamma', [('swimmer', 512, 5, 500, 12, 1.0),        # params-swimmer.json's validation batch: 160 tiles, two columns per model
                                                  ('half_cheetah', 1024, 5, 500, 8, 0.99),
                                                  ('ant', 1024, 3, 77, 15, 0.97),           # sticky dones mask, a partial last tile
                                                  ('hopper', 512, 2, 16, 9, 1.0),           # one tile per model
                                                  ('snake', 1024, 4, 1, 5, 0.9)])           # one env
def test_resident_validation_costs_against_oracle_and_the_stepwise_sweep(env, hid, K, Bv, T, gamma, monkeypatch):
    """metrpo_validation_cost (build_policy_graph's forward, model_based_rl.py:106-151) in the resident kernel's validation mode: model k's
    workgroups deal ITS env tiles to their columns, the post wave of (model, tile) follows the deterministic policy under that one model and
    sums gamma^t cost.  Against the oracle and against the step-wise GEMM sweep (det_gemm.hip) it replaces for these shapes."""
    from oracle import metrpo_oracle as O
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (hid, hid), (32, 32), seed=31, n_pool=512)
    if env == 'ant':
        pool[::3, 2] = 0.25; dm.diff_mean[2] = -0.01
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    s0 = pool[:Bv].astype(np.float32)
    got = cpu(eng.validation_cost(s0, T, gamma))
    eng.comm_check()                                                      # no hand-over timed out
    eng.set_option('METRPO_NO_RESIDENT_VALIDATION', '1')
    sweep = cpu(eng.validation_cost(s0, T, gamma))
    eng.set_option('METRPO_NO_RESIDENT_VALIDATION', None)
    ref = O.validation_costs(dm, theta.astype(np.float32).astype(np.float64), pdims, env, s0.astype(np.float64), T, gamma)
    np.testing.assert_allclose(got, ref, **TOL.VALIDATION_COST)
    np.testing.assert_allclose(got, sweep, **TOL.VALIDATION_COST)
    assert not np.array_equal(got, sweep)                                 # two different kernels did run (their float32 sums differ in the last bits)
    again = cpu(eng.validation_cost(s0, T, gamma))
    np.testing.assert_array_equal(got, again)                             # bitwise repeatable
    for ntw in (1, 2, 4):                                                 # tiles per post wave (the launcher picks by a cost model): same sums whatever the deal
        eng.set_option('VAL_PLAN', str(ntw))
        np.testing.assert_allclose(cpu(eng.validation_cost(s0, T, gamma)), got, rtol=1e-6, atol=1e-6)
    for nb in (2, 3, 5):                                                  # the batch in chunks (one launch each; the last one ragged)
        if Bv >= nb:
            eng.set_option('VAL_PLAN', '0,%d' % nb)
            np.testing.assert_allclose(cpu(eng.validation_cost(s0, T, gamma)), got, rtol=1e-6, atol=1e-6)
    eng.comm_check()


def test_rollout_under_a_cu_mask_takes_a_launch_that_fits(tmp_path):
    """The resident kernels need their whole grid on the chip at once.  With most CUs masked away (HSA_CU_MASK: the device property still says 256)
    the launch rule must size the grid by the CUs that really schedule the process's waves -- or take the step-wise path -- WITHOUT first burning
    the 2 s hand-over bound and reporting invalid trajectories (round-3 verdict).  Same draws, masked vs unmasked: same path structure, states
    within the fp32 rounding of a 6-step free run."""
    import os, subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'masked.py'
    script.write_text('''
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import helpers as Hh
eng, dm, theta, pdims, pool = Hh.make_engine("swimmer", 5, (512, 512), (32, 32), seed=5)
ncu = eng.schedulable_cus()
B, T, H = 100, 6, 6
dr = Hh.draws(np.random.RandomState(3), 5, B, T, dm.ns, dm.na, len(pool))
dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
eng.rollout(B, T, H, "step_rand", pool, **dr32); torch.cuda.synchronize()          # first call: allocations, census
t0 = time.time()
traj = eng.rollout(B, T, H, "step_rand", pool, **dr32); torch.cuda.synchronize()
dt = time.time() - t0
eng.comm_check()                                                                   # raises if a hand-over timed out
np.save(sys.argv[1], traj.obs.cpu().numpy())
print(json.dumps({"ncu": ncu, "kernel": eng.last_rollout_kernel(), "dt": dt}))
''' % (root, root))

    def run(mask, out):
        env = dict(os.environ)
        env.pop('HSA_CU_MASK', None)
        if mask:
            env['HSA_CU_MASK'] = mask
        res = subprocess.run([sys.executable, str(script), str(out)], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
        return json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])

    full = run(None, tmp_path / 'full.npy')
    masked = run('0:0-47', tmp_path / 'masked.npy')                                 # 48 CUs: fewer than one column of compute workgroups (5 x 16 = 80)
    if masked['ncu'] >= full['ncu']:
        pytest.skip('HSA_CU_MASK is not honoured on this box (census sees %d CUs either way)' % masked['ncu'])
    assert full['kernel'] == 'resident' and full['ncu'] >= 200
    assert masked['ncu'] <= 48 and masked['kernel'] == 'gemm-stepwise', masked
    assert masked['dt'] < 1.0, masked                                                # no 2 s hand-over bound burnt on the way
    np.testing.assert_allclose(np.load(tmp_path / 'masked.npy'), np.load(tmp_path / 'full.npy'), **TOL.CROSS_KERNEL)
    mid = run('0:0-119', tmp_path / 'mid.npy')                                       # 120 CUs: a resident launch sized for them
    assert mid['ncu'] <= 120 and mid['dt'] < 1.0, mid
    np.testing.assert_allclose(np.load(tmp_path / 'mid.npy'), np.load(tmp_path / 'full.npy'), **TOL.CROSS_KERNEL)
