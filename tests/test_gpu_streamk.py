"""Step-wise rollout on the stream-K fused ensemble kernel (csrc/mlp_streamk.h): the whole dynamics ensemble of a time step in ONE evenly split
launch (two hidden layers: layer 0 as operand producer, output layer in the epilogue; three hidden layers: two launches).  Reference:
training.py:171-214,218-269 (the K-head forward), env_helpers.py:597-635 (the step around it).  Every case runs through metrpo_rollout
(C ABI) with supplied draws and is compared with the float64 oracle teacher-forced on the device's own states, and with the tile-GEMM
path of the same library on the same draws.

Tolerances: tests/tolerances.py / DESIGN.md section 5 table, row 2 "one step, wide nets" (sums of 512-1024 fp32 products in a fixed
order that differs from the oracle's float64 order)."""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
import helpers as Hh
import tolerances as TOL

pytestmark = pytest.mark.gpu
WIDE_TOL = TOL.WIDE


def cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


def make_engine(*a, **k):
    """This file tests the LAUNCH-PER-STEP stream-K path; the persistent form that production takes for the two-hidden-layer shapes (mlp_persist.h) has
    tests/test_gpu_persist.py."""
    out = Hh.make_engine(*a, **k)
    out[0].set_option('NO_PERSIST', '1')
    return out


def _rollout_vs_oracle(eng, dm, theta, pdims, pool, env, K, B, T, H, sam_mode='step_rand', seed=6):
    th = theta.astype(np.float32).astype(np.float64)
    pool32 = pool.astype(np.float32).astype(np.float64)
    dr = Hh.draws(np.random.RandomState(seed), K, B, T, dm.ns, dm.na, len(pool))
    dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    traj = eng.rollout(B, T, H, sam_mode, pool, **dr32)
    drf = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in dr32.items()}
    ref = Hh.oracle_rollout(dm, th, pdims, env, pool32, drf, B, T, H, sam_mode, teacher_obs=cpu(traj.obs))
    np.testing.assert_allclose(cpu(traj.mean), ref['mean'], **WIDE_TOL)
    np.testing.assert_allclose(cpu(traj.rew), ref['rew'], **WIDE_TOL)
    dn = cpu(traj.done).astype(bool)
    assert np.array_equal(dn, ref['done'])
    for t in range(T - 1):
        np.testing.assert_allclose(cpu(traj.obs[t + 1])[~dn[t]], ref['next'][t][~dn[t]], **WIDE_TOL)
    return traj, dr32


# (env, K, hidden, B): every (input steps, output tiles) instantiation of the fused kernel -- swimmer (3, 1), hopper (4, 1), snake (5, 1),
# half-cheetah (6, 2), ant (9, 2) -- and the three-hidden-layer form (humanoid: stored layer + (.., 4)); batches that are not multiples of 16 / 128.
# Below one tile per CU the forced launches cut every tile into pieces that run side by side and are added at their ends (SkArgs::late).
SHAPES = [('swimmer', 5, (512, 512), 100), ('hopper', 3, (256, 256), 77), ('snake', 4, (256, 512), 130), ('half_cheetah', 5, (1024, 1024), 48),
          ('ant', 10, (512, 512), 64), ('ant', 3, (512, 256), 333), ('humanoid', 4, (1024, 1024, 1024), 24), ('humanoid', 2, (256, 512, 256), 150),
          ('humanoid', 3, (512, 1024), 200), ('humanoid', 5, (1024, 1024), 500)]    # two hidden layers behind 76 inputs: layer 0 stored, the rest in one launch; the params file's batch (80 tiles: pieces side by side)


@pytest.mark.parametrize('env,K,dh,B', SHAPES)
def test_streamk_rollout_vs_oracle_and_tile_gemm(env, K, dh, B, monkeypatch):
    ph = (100, 50, 25) if env == 'humanoid' else (32, 32)
    T, H = 5, 3
    eng, dm, theta, pdims, pool = make_engine(env, K, dh, ph, seed=71)
    if env == 'ant':
        pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    eng.set_option('METRPO_STREAMK', '1')                  # also below one tile per CU
    eng.set_rollout_variant(1)                                 # large nets: the step-wise path even where the resident kernel applies
    traj, dr32 = _rollout_vs_oracle(eng, dm, theta, pdims, pool, env, K, B, T, H)
    assert eng.last_rollout_kernel() == 'gemm-streamk'
    again = eng.rollout(B, T, H, 'step_rand', pool, **dr32)    # bitwise repeatable
    assert torch.equal(again.obs, traj.obs) and torch.equal(again.rew, traj.rew)
    eng.set_option('METRPO_STREAMK', None)
    eng.set_option('METRPO_NO_STREAMK', '1')
    tile = eng.rollout(B, T, H, 'step_rand', pool, **dr32)     # free-running for 5 steps: same path structure, states within fp32 rounding of 5 steps
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    assert torch.equal(tile.tpath, traj.tpath) and torch.equal(tile.done, traj.done)
    np.testing.assert_allclose(cpu(tile.obs), cpu(traj.obs), **TOL.CROSS_KERNEL)


@pytest.mark.parametrize('sam_mode', list(O.SAM_MODES))
def test_streamk_rollout_all_sam_modes(sam_mode, monkeypatch):
    env, K, B, T, H = 'half_cheetah', 4, 70, 6, 3
    eng, dm, theta, pdims, pool = make_engine(env, K, (256, 256), (32, 32), seed=61)
    eng.set_option('METRPO_STREAMK', '1')
    eng.set_rollout_variant(1)
    _rollout_vs_oracle(eng, dm, theta, pdims, pool, env, K, B, T, H, sam_mode=sam_mode)
    assert eng.last_rollout_kernel() == 'gemm-streamk'


def test_streamk_is_selected_by_itself_at_the_humanoid_params_file_shape():
    """params-humanoid.json: K = 5, 2 x 1024, 100 envs, rounds of 100 steps side by side = 500 rows per head: 80 tiles of 34 units on the chip's CUs.  With one
    hand-over per tile (SkArgs::late) the stream-K launch beats the tile GEMM there and is the default (rollout_gemm.hip sk_select: the wide-input form from 8
    units per workgroup up), with the tile GEMM for layer 0 and the split pre-step + separate post launch of small batches.  Merged rounds (tiles spanning two
    rounds, resets inside the batch), production draws: bitwise repeatable, and within cross-kernel rounding of the tile-GEMM path (option STREAMK_LATE = 0)."""
    env, K, B, H, R = 'humanoid', 5, 100, 3, 5
    eng, dm, theta, pdims, pool = make_engine(env, K, (1024, 1024), (100, 50, 25), seed=91)
    a = eng.rollout(B, R * H, H, 'step_rand', pool, seed=7)
    if eng.last_rollout_kernel() != 'gemm-streamk':
        pytest.skip('fewer than 8 units per workgroup on this device (CU count): the tile GEMMs stay')
    keep = [x.clone() for x in (a.obs, a.act, a.mean, a.rew, a.done, a.tpath, a.last_obs)]
    again = eng.rollout(B, R * H, H, 'step_rand', pool, seed=7)
    for x, y in zip(keep, (again.obs, again.act, again.mean, again.rew, again.done, again.tpath, again.last_obs)):
        assert torch.equal(x, y)
    eng.set_option('STREAMK_LATE', '0')
    tile = eng.rollout(B, R * H, H, 'step_rand', pool, seed=7)
    assert eng.last_rollout_kernel() == 'gemm-stepwise'
    assert torch.equal(tile.done, keep[4]) and torch.equal(tile.tpath, keep[5])
    for x, y in zip(keep[:4], (tile.obs, tile.act, tile.mean, tile.rew)):
        np.testing.assert_allclose(cpu(x), cpu(y), **TOL.CROSS_KERNEL)
    eng.set_option('STREAMK_LATE', None)
    eng.set_option('METRPO_SEQ_ROUNDS', '1')                     # the rounds one after the other: 20 tiles per launch, the tile GEMMs
    seq = eng.rollout(B, R * H, H, 'step_rand', pool, seed=7)
    for x, y in zip(keep[:4], (seq.obs, seq.act, seq.mean, seq.rew)):
        np.testing.assert_allclose(cpu(x), cpu(y), **TOL.CROSS_KERNEL)


def test_streamk_split_tiles_equal_the_oracle_and_do_not_depend_on_the_split():
    """More tiles than CUs and not a multiple of them: workgroup ranges start and end inside tiles, accumulators are handed over between
    workgroups.  K = 5 heads x 32 row blocks x 2 column blocks = 320 tiles on the chip's CUs; the rollout picks the path by itself."""
    env, K, B, T, H = 'ant', 5, 4000, 3, 3
    eng, dm, theta, pdims, pool = make_engine(env, K, (512, 512), (32, 32), seed=81)
    traj, dr32 = _rollout_vs_oracle(eng, dm, theta, pdims, pool, env, K, B, T, H)
    assert eng.last_rollout_kernel() == 'gemm-streamk'
    again = eng.rollout(B, T, H, 'step_rand', pool, **dr32)
    assert torch.equal(again.obs, traj.obs) and torch.equal(again.rew, traj.rew) and torch.equal(again.mean, traj.mean)


def test_streamk_xcd_aware_ranges_are_bitwise_the_plain_split(monkeypatch):
    """The workgroups of one XCD take consecutive ranges of the (tile, chunk) sequence (mlp_streamk.h: SkArgs::xcd; parts cut at tile boundaries): who
    computes what changes, the k-ordered chains do not -- every tensor bit for bit the plain blockIdx-ordered split."""
    env, K, B, T, H = 'half_cheetah', 5, 2600, 3, 3                # 5 x 21 x 4 = 420 tiles of 32 + 1 units
    eng, dm, theta, pdims, pool = make_engine(env, K, (1024, 1024), (32, 32), seed=85)
    xcd = eng.rollout(B, T, H, 'step_rand', pool, seed=3)
    assert eng.last_rollout_kernel() == 'gemm-streamk'
    keep = [x.clone() for x in (xcd.obs, xcd.rew, xcd.mean, xcd.done)]
    eng.set_option('STREAMK_PLACE', 'flat')
    plain = eng.rollout(B, T, H, 'step_rand', pool, seed=3)
    for a, b in zip(keep, (plain.obs, plain.rew, plain.mean, plain.done)):
        assert torch.equal(a, b)


def test_streamk_random_shapes_equal_the_tile_gemms(monkeypatch):
    """Shapes nobody tuned for: head counts, batches and layer widths drawn at random (>= one tile per CU, so the path is chosen by itself; tile counts that
    are not multiples of the 8 XCD parts, ragged last row blocks, 1-3 column blocks, contraction lengths from 128 to 1024) -- stream-K against the tile
    GEMMs on the same draws, two steps free-running."""
    rs = np.random.RandomState(4321)
    envs = ['swimmer', 'half_cheetah', 'ant', 'hopper', 'snake']
    for case in range(8):
        env = envs[case % len(envs)]
        n1 = int(rs.choice([256, 512, 768])); k1 = int(rs.choice([128, 160, 256, 352, 512, 1024]))
        K = int(rs.randint(2, 8))
        tiles_per_row_block = K * (n1 // 256)
        B = 128 * int(np.ceil(300.0 / tiles_per_row_block)) + int(rs.randint(1, 128))          # > 256 tiles, ragged last row block
        eng, dm, theta, pdims, pool = make_engine(env, K, (k1, n1), (32, 32), seed=500 + case)
        msg = str((case, env, K, (k1, n1), B))
        sk = eng.rollout(B, 2, 2, 'step_rand', pool, seed=case)
        assert eng.last_rollout_kernel() == 'gemm-streamk', msg
        keep = [x.clone() for x in (sk.obs, sk.rew, sk.mean, sk.done)]
        eng.set_option('METRPO_NO_STREAMK', '1')
        tile = eng.rollout(B, 2, 2, 'step_rand', pool, seed=case)
        eng.set_option('METRPO_NO_STREAMK', None)
        assert eng.last_rollout_kernel() == 'gemm-stepwise', msg
        assert torch.equal(keep[3], tile.done), msg
        for a, b in zip(keep[:3], (tile.obs, tile.rew, tile.mean)):
            np.testing.assert_allclose(cpu(a), cpu(b), **TOL.CROSS_KERNEL, err_msg=msg)
        del eng


def test_streamk_side_by_side_pieces_on_random_small_shapes():
    """Below one tile per CU (SkArgs::late): a tile is cut into pieces that run at the same time, every piece but the last exports its own sums and the last adds
    them all -- it finds the earlier pieces from the split's arithmetic (mlp_streamk.h: the workgroups in front of it whose ranges start inside the same tile).
    Head counts, batches and widths drawn at random so that a tile is 2 ... 13 pieces, ranges end inside the epilogue units, last row blocks are ragged; forced
    launches (from 2 units per workgroup) against the tile GEMMs on the same draws, two steps free-running, and bitwise repeatable."""
    rs = np.random.RandomState(97531)
    envs = ['swimmer', 'half_cheetah', 'ant', 'hopper', 'snake', 'humanoid']
    ran = 0
    for case in range(10):
        env = envs[case % len(envs)]
        k1 = int(rs.choice([256, 512, 768, 1024])); n1 = int(rs.choice([256, 512, 1024]))
        K = int(rs.randint(1, 7))
        B = int(rs.randint(17, 900))
        ph = (100, 50, 25) if env == 'humanoid' else (32, 32)
        eng, dm, theta, pdims, pool = make_engine(env, K, (k1, n1), ph, seed=900 + case)
        msg = str((case, env, K, (k1, n1), B))
        eng.set_option('STREAMK', '1'); eng.set_rollout_variant(1)
        sk = eng.rollout(B, 2, 2, 'step_rand', pool, seed=case)
        if eng.last_rollout_kernel() != 'gemm-streamk':
            del eng; continue
        ran += 1
        keep = [x.clone() for x in (sk.obs, sk.rew, sk.mean, sk.done)]
        again = eng.rollout(B, 2, 2, 'step_rand', pool, seed=case)
        for a, b in zip(keep, (again.obs, again.rew, again.mean, again.done)):
            assert torch.equal(a, b), msg
        eng.set_option('STREAMK', None); eng.set_option('NO_STREAMK', '1')
        tile = eng.rollout(B, 2, 2, 'step_rand', pool, seed=case)
        assert eng.last_rollout_kernel() == 'gemm-stepwise', msg
        assert torch.equal(keep[3], tile.done), msg
        for a, b in zip(keep[:3], (tile.obs, tile.rew, tile.mean)):
            np.testing.assert_allclose(cpu(a), cpu(b), **TOL.CROSS_KERNEL, err_msg=msg)
        del eng
    assert ran >= 6, ran


@pytest.mark.parametrize('env,K,dh,B', [('humanoid', 6, (1024, 1024, 1024), 1500), ('ant', 5, (256, 512, 512), 3400), ('humanoid', 5, (512, 1024), 2600)])
def test_stored_layer0_kernel_equals_the_tile_gemm(env, K, dh, B, monkeypatch):
    """Layer 0 of the forms that store it (three hidden layers; two behind Humanoid's 77 inputs) on k_l0_rows -- the head's weight slice LDS-resident, bias as
    one more input row -- against the tile GEMM it replaces (METRPO_NO_L0_ROWS), and both against the oracle through the rollout."""
    ph = (100, 50, 25) if env == 'humanoid' else (32, 32)
    eng, dm, theta, pdims, pool = make_engine(env, K, dh, ph, seed=87)
    traj, dr32 = _rollout_vs_oracle(eng, dm, theta, pdims, pool, env, K, B, 2, 2)
    assert eng.last_rollout_kernel() == 'gemm-streamk'
    eng.set_option('METRPO_NO_L0_ROWS', '1')
    tile = eng.rollout(B, 2, 2, 'step_rand', pool, **dr32)
    assert eng.last_rollout_kernel() == 'gemm-streamk'
    assert torch.equal(tile.done, traj.done)
    # (both are k-ordered fmaf chains over the same 77 products, the bias last: on the fixtures the two kernels agree bit for bit)
    np.testing.assert_allclose(cpu(tile.obs), cpu(traj.obs), **TOL.CROSS_KERNEL)


def test_streamk_three_hidden_layers_split_tiles():
    env, K, B, T, H = 'humanoid', 6, 1500, 2, 2                # 6 x 12 x 4 = 288 tiles per launch
    eng, dm, theta, pdims, pool = make_engine(env, K, (1024, 1024, 1024), (100, 50, 25), seed=83)
    traj, dr32 = _rollout_vs_oracle(eng, dm, theta, pdims, pool, env, K, B, T, H)
    assert eng.last_rollout_kernel() == 'gemm-streamk'
    again = eng.rollout(B, T, H, 'step_rand', pool, **dr32)
    assert torch.equal(again.obs, traj.obs)


@pytest.mark.parametrize('env,K,dh,B,sam_mode,streamk', [
    ('ant', 4, (512, 512), 333, 'step_rand', True), ('ant', 4, (256, 256), 100, 'model_med', False), ('half_cheetah', 3, (256, 512), 77, 'model_mean_std', True),
    ('hopper', 3, (256, 256), 130, 'eps_rand', False), ('snake', 5, (512, 512), 64, 'model_mean', True), ('swimmer', 5, (512, 512), 100, 'one_model', False),
    ('humanoid', 4, (1024, 1024), 150, 'step_rand', False), ('humanoid', 3, (256, 512, 256), 77, 'model_med', True), ('humanoid', 2, (128, 256), 500, 'model_mean_std', False)])
def test_step_closed_in_the_next_launch_is_bitwise_the_two_launch_sequence(env, K, dh, B, sam_mode, streamk, monkeypatch):
    """k_big_post(t - 1) folded into the pre-step launch of step t (rollout_gemm.hip: k_big_pre_mfma<ENV, true>): same arithmetic in the same
    order -- every trajectory tensor bit for bit what the two-launch sequence writes, on the stream-K path and on the tile GEMMs, for every
    selection mode, with early termination (Ant) and horizon resets inside the rollout."""
    T, H = 7, 3
    eng, dm, theta, pdims, pool = make_engine(env, K, dh, (100, 50, 25) if env == 'humanoid' else (32, 32), seed=91)      # (Humanoid: k_big_pre_mfma3<.., true>)
    if env == 'ant':
        pool[::3, 2] = 0.21; dm.diff_mean[2] = -0.02
        eng.set_dynamics_layers(dm.Ws, dm.bs, dm.in_mean, dm.in_std, dm.diff_mean, dm.diff_std)
    eng.set_option('METRPO_STREAMK' if streamk else 'METRPO_NO_STREAMK', '1')
    eng.set_option('METRPO_STEP_MERGE', '1')                # Humanoid behind the tile GEMMs: merged only on request
    eng.set_rollout_variant(1)
    dr = Hh.draws(np.random.RandomState(8), K, B, T, dm.ns, dm.na, len(pool))
    dr32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in dr.items()}
    merged = eng.rollout(B, T, H, sam_mode, pool, **dr32)
    eng.set_option('METRPO_STEP_MERGE', '0')
    two = eng.rollout(B, T, H, sam_mode, pool, **dr32)
    for name in ('obs', 'act', 'rew', 'mean', 'done', 'tpath', 'last_obs'):
        assert torch.equal(getattr(merged, name), getattr(two, name)), name
    # and with the library's own draws (Philox): same streams in both forms
    eng.set_option('METRPO_STEP_MERGE', '1')
    m2 = eng.rollout(B, T, H, sam_mode, pool, seed=5)
    eng.set_option('METRPO_STEP_MERGE', '0')
    t2 = eng.rollout(B, T, H, sam_mode, pool, seed=5)
    for name in ('obs', 'act', 'rew', 'mean', 'done', 'tpath', 'last_obs'):
        assert torch.equal(getattr(m2, name), getattr(t2, name)), name


def test_streamk_xcd_teams_are_bitwise_the_consecutive_ranges():
    """Three hidden layers (activations read from memory) with enough tiles for the XCD-team split (mlp_streamk.h: SkArgs::team -- the workgroups of an XCD walk
    super-tiles in lock step, pieces handed to the same slot of the next XCD): who computes what changes, every output stays one k-ordered chain -- bit for bit the
    consecutive-range split, which tests above hold against the oracle."""
    env, K, B, T, H = 'humanoid', 8, 2200, 3, 3                   # 8 heads x 4 column blocks x 18 row blocks = 576 tiles per layer launch (>= 2 per workgroup)
    eng, dm, theta, pdims, pool = make_engine(env, K, (1024, 1024, 1024), (100, 50, 25), seed=89)
    team = eng.rollout(B, T, H, 'step_rand', pool, seed=3)
    assert eng.last_rollout_kernel() == 'gemm-streamk'
    keep = [x.clone() for x in (team.obs, team.rew, team.mean, team.done)]
    eng.set_option('STREAMK_PLACE', 'xcd')
    plain = eng.rollout(B, T, H, 'step_rand', pool, seed=3)
    for a, b in zip(keep, (plain.obs, plain.rew, plain.mean, plain.done)):
        assert torch.equal(a, b)
    assert np.isfinite(cpu(team.obs)).all() and float(np.abs(cpu(team.obs[1]) - cpu(team.obs[0])).max()) > 1e-3
