"""Persistent stream-K rollout (csrc/mlp_persist.h): every step of a rollout chunk of a two-hidden-layer ensemble (hidden >= 256) in ONE launch -- whole
k_mlp_sk tiles, the row block's last-arriving workgroup closes the step and prepares the next one with the wave functions of the launch-per-step path
(big_prepost.h).  Reference: training.py:171-214,218-269 (K-head forward), env_helpers.py:597-635 (step), samplers/vectorized_sampler.py:45-116.
Same k-ordered sums, same closing arithmetic: every tensor must be BIT FOR BIT the launch-per-step stream-K path's (option NO_PERSIST), which
tests/test_gpu_streamk.py holds against the oracle; the oracle comparison is repeated here on the persistent launch itself."""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
import helpers as Hh
import tolerances as TOL
from test_gpu_streamk import _rollout_vs_oracle

pytestmark = pytest.mark.gpu
FIELDS = ('obs', 'act', 'mean', 'rew', 'done', 'tpath', 'last_obs')


def cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _both(eng, B, T, H, mode, pool, **kw):
    """(persistent, launch-per-step) trajectories of the same call."""
    eng.set_option('NO_PERSIST', None)
    a = eng.rollout(B, T, H, mode, pool, **kw)
    assert eng.last_rollout_kernel() == 'streamk-persistent', eng.last_rollout_kernel()
    a = {k: getattr(a, k).clone() for k in FIELDS}
    # two adjacent column blocks per tile sharing the layer-0 producer (selected by itself from ~1.2 such tiles per workgroup and step; forced here), and a
    # different number of closing workgroups: who computes and who closes changes, the sums do not
    for opt, val in (('PERSIST_WIDE', '1'), ('PERSIST_NCLOSE', '3')):
        eng.set_option(opt, val)
        n = eng.rollout(B, T, H, mode, pool, **kw)
        assert eng.last_rollout_kernel() == 'streamk-persistent'
        for k in FIELDS:
            assert torch.equal(a[k], getattr(n, k)), (opt, k)
        eng.set_option(opt, None)
    eng.set_option('NO_PERSIST', '1')
    eng.set_option('STREAMK_LATE', '0')                          # forced launches below one tile per CU: whole tiles (the side-by-side pieces of SkArgs::late add in another order)
    b = eng.rollout(B, T, H, mode, pool, **kw)
    assert eng.last_rollout_kernel() == 'gemm-streamk', eng.last_rollout_kernel()
    b = {k: getattr(b, k).clone() for k in FIELDS}
    eng.set_option('NO_PERSIST', None)
    return a, b


# every (env, input steps, output tiles) instantiation: swimmer (3, 1), hopper (4, 1), snake (5, 1), half-cheetah (6, 2), ant (9, 2); ragged batches, 1-2
# column blocks, several XCD lists shorter than the others (tiles per step not a multiple of 8), fewer tiles than workgroups
SHAPES = [('swimmer', 5, (512, 512), 300, 9, 4), ('hopper', 3, (256, 256), 77, 6, 3), ('snake', 4, (256, 512), 530, 7, 7), ('half_cheetah', 5, (1024, 1024), 148, 5, 2),
          ('ant', 10, (512, 512), 200, 8, 3), ('ant', 3, (512, 256), 333, 11, 5)]


@pytest.mark.parametrize('env,K,dh,B,T,H', SHAPES)
def test_persistent_rollout_is_bitwise_the_launch_per_step_path_and_matches_the_oracle(env, K, dh, B, T, H):
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, dh, (32, 32), seed=71)
    if env == 'ant':
        pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    eng.set_option('STREAMK', '1')                               # also below one tile per CU
    eng.set_rollout_variant(1)                                   # large nets: not the resident kernel
    traj, dr32 = _rollout_vs_oracle(eng, dm, theta, pdims, pool, env, K, B, T, H)
    assert eng.last_rollout_kernel() == 'streamk-persistent'
    a, b = _both(eng, B, T, H, 'step_rand', pool, **dr32)
    for k in FIELDS:
        assert torch.equal(a[k], b[k]), k
        assert torch.equal(a[k], getattr(traj, k)), k            # and repeatable
    if env == 'ant':
        assert bool(a['done'][:-1].any()) and int(a['tpath'].max()) < H


@pytest.mark.parametrize('sam_mode', list(O.SAM_MODES))
def test_persistent_rollout_all_sam_modes_production_draws(sam_mode):
    env, K, B, T, H = 'half_cheetah', 4, 270, 9, 4
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (256, 256), (32, 32), seed=61)
    eng.set_option('STREAMK', '1'); eng.set_rollout_variant(1)
    a, b = _both(eng, B, T, H, sam_mode, pool, seed=17)          # Philox draws on the device
    for k in FIELDS:
        assert torch.equal(a[k], b[k]), k
    assert bool(a['done'][H - 1].all()) and np.isfinite(cpu(a['obs'])).all()


def test_persistent_rollout_at_the_c3_share_many_steps():
    """BASELINE C3's per-GPU share (Ant, K = 10, 2 x 512, B = 2500: 400 tiles per step on 256 workgroups, 1.56 tiles per workgroup and step), 40 steps with
    early termination: the steady state of the launch -- arrivals, closing workgroups and ready flags interleaved over many steps -- bit for bit the
    launch-per-step path; twice, to catch anything that depends on timing."""
    env, K, B, T, H = 'ant', 10, 2500, 40, 500
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (512, 512), (32, 32), seed=81, n_pool=1024)
    pool[::5, 2] = 0.21; dm.diff_mean[2] = -0.004
    eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    a, b = _both(eng, B, T, H, 'step_rand', pool, seed=5)
    for k in FIELDS:
        assert torch.equal(a[k], b[k]), k
    a2, _ = _both(eng, B, T, H, 'step_rand', pool, seed=5)
    for k in FIELDS:
        assert torch.equal(a[k], a2[k]), k
    assert 0 < int(a['done'].sum()) and np.isfinite(cpu(a['rew'])).all()


def test_persistent_chunked_continuation_equals_one_call():
    """The sampler's chunks (early-terminating envs: metrpo_rollout continued with t0 / resume / last_state / stop) on the persistent launch = one long call;
    a raised stop flag turns the launch into a no-op."""
    env, K, B, H = 'ant', 5, 300, 6
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (256, 256), (32, 32), seed=13)
    pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
    eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    eng.set_option('STREAMK', '1'); eng.set_rollout_variant(1)
    whole = eng.rollout(B, 12, H, 'eps_rand', pool, seed=9)
    assert eng.last_rollout_kernel() == 'streamk-persistent'
    whole = {k: getattr(whole, k).clone() for k in FIELDS}
    dev = eng.device
    lts, lmd = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    stop = torch.zeros(1, dtype=torch.int32, device=dev)
    parts, resume, t0 = [], None, 0
    for Tc in (5, 4, 3):
        tr = eng.rollout(B, Tc, H, 'eps_rand', pool, seed=9, t0=t0, resume=resume, last_state=(lts, lmd), stop=stop)
        assert eng.last_rollout_kernel() == 'streamk-persistent'
        parts.append({k: getattr(tr, k).clone() for k in FIELDS})
        resume = (tr.last_obs.clone(), lts.clone(), lmd.clone())
        t0 += Tc
    for k in ('obs', 'act', 'mean', 'rew', 'done', 'tpath'):
        assert torch.equal(torch.cat([p_[k] for p_ in parts], 0), whole[k]), k
    assert torch.equal(parts[-1]['last_obs'], whole['last_obs'])
    assert bool((whole['done'].bool() & (whole['tpath'] < H - 1)).any())          # some episodes ended before the horizon
    stop.fill_(1)
    sentinel = eng.alloc_trajectory(B, 3, H)
    sentinel.rew.fill_(-7.0)
    eng.rollout(B, 3, H, 'eps_rand', pool, seed=9, t0=t0, resume=resume, last_state=(lts, lmd), stop=stop, out=sentinel)
    assert bool((sentinel.rew == -7.0).all())


def test_persistent_is_not_selected_on_a_shared_gpu_or_for_one_step():
    env, K, B, H = 'swimmer', 5, 300, 4
    eng, dm, theta, pdims, pool = Hh.make_engine(env, K, (512, 512), (32, 32), seed=3)
    eng.set_option('STREAMK', '1'); eng.set_rollout_variant(1)
    eng.rollout(B, 1, H, 'step_rand', pool, seed=1)
    assert eng.last_rollout_kernel() == 'gemm-streamk'           # a single step: nothing to chain
    eng.rollout(B, 3, H, 'step_rand', pool, seed=1)
    assert eng.last_rollout_kernel() == 'streamk-persistent'
    eng.set_exclusive(False)                                     # other tenants: no kernel that waits on workgroups of its own launch
    eng.rollout(B, 3, H, 'step_rand', pool, seed=1)
    assert eng.last_rollout_kernel() == 'gemm-streamk'


def test_stop_rule_inside_the_launch_gives_the_chunked_samplers_paths():
    """Early-terminating env through VectorizedSampler.obtain_samples: on the persistent path everything behind the first ceil(batch / B) steps is ONE call
    that applies `while n_samples < batch_size` (vectorized_sampler.py:60,104) inside the launch (metrpo_rollout_args::stop_batch) and stops stepping at the
    stop step; the launch-per-step path runs chunks and a flag.  Same stop step, same trajectory prefix bit for bit, same list of paths."""
    from test_gpu_api import build_algo
    B, H, batch = 300, 12, 2100
    outs = []
    for persist in (True, False):
        algo, eng, dm, theta, pdims, pool = build_algo('ant', K=4, B=B, H=H, batch=batch, dyn_hidden=(256, 256))
        pool[::3, 2] = 0.21
        algo.env.env.states[::3, 2] = 0.21; algo.env.env._dev = None
        eng.set_option('STREAMK', '1'); eng.set_option('STREAMK_LATE', '0'); eng.set_rollout_variant(1)
        if not persist:
            eng.set_option('NO_PERSIST', '1')
        algo.start_worker()
        paths = algo.obtain_samples(3)
        assert eng.last_rollout_kernel() == ('streamk-persistent' if persist else 'gemm-streamk')
        tr = paths.traj
        outs.append({k: getattr(tr, k).clone() for k in ('obs', 'act', 'mean', 'rew', 'done', 'tpath')})
        done, tp = cpu(tr.done), cpu(tr.tpath)
        per_step = (done * (tp + 1)).sum(axis=1).cumsum()
        assert per_step[-1] >= batch and per_step[-2] < batch        # the LAST step is the first at which completed samples >= batch
        assert tr.T > -(-batch // B)                                 # the stop step lies behind the first call
    for k in outs[0]:
        assert outs[0][k].shape == outs[1][k].shape and torch.equal(outs[0][k], outs[1][k]), k
