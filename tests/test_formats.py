"""On-disk formats at the boundary: host logic, runs without a GPU."""
import os
import pickle
import numpy as np
from conftest import load_golden, GOLDEN as GOLDEN_DIR
from oracle import dynamics_oracle as D


def _cases():
    d = load_golden('collect_split')
    lens = d['lens']; off = np.concatenate([[0], np.cumsum(lens)])
    Os = [d['O'][off[i]:off[i + 1]] for i in range(len(lens))]; As = [d['A'][off[i]:off[i + 1]] for i in range(len(lens))]
    return Os, As


def test_rollouts_pickle_written_by_the_reference_is_read(tmp_path):
    """tests/golden/new_rollouts_0.pkl is the file the reference's collect_data wrote (model_based_rl.py:809-811)."""
    from metrpo_amd import formats
    x, y = formats.load_rollouts(os.path.join(GOLDEN_DIR, 'new_rollouts_0.pkl'))
    Os, As = _cases()
    xo, yo = D.trajectories_to_pairs(Os, As)
    np.testing.assert_array_equal(x, xo); np.testing.assert_array_equal(y, yo)
    p = formats.save_rollouts(str(tmp_path), 3, x, y)
    assert os.path.basename(p) == 'new_rollouts_3.pkl'
    with open(p, 'rb') as f:
        x2, y2 = pickle.load(f)                              # what the reference's own loader does (:425-433 style tuple unpack)
    np.testing.assert_array_equal(x2, x); np.testing.assert_array_equal(y2, y)
    with open(p, 'rb') as f, open(os.path.join(GOLDEN_DIR, 'new_rollouts_0.pkl'), 'rb') as g:
        assert f.read() == g.read()                          # same pickle bytes as the reference produced


def test_validation_init_pickles(tmp_path):
    from metrpo_amd import formats
    states = [np.arange(5.0) + i for i in range(4)]          # the vip == vrip branch pickles a python LIST of arrays (:453-457)
    p = str(tmp_path / 'sub' / 'val_init.pkl')
    formats.save_validation_init(p, states)
    with open(p, 'rb') as f:
        assert isinstance(pickle.load(f), list)
    v = formats.load_validation_init(p)
    assert v.dtype == np.float32 and v.shape == (4, 5)
    formats.save_validation_init(p, np.array(states))        # the other branch pickles an ndarray (:478-482)
    np.testing.assert_array_equal(formats.load_validation_init(p), v)


def test_progress_csv_round_trip(tmp_path):
    from metrpo_amd import formats
    cols = formats.progress_columns()
    assert cols[0] == 'collect_data_time' and '# policy updates' in cols and 'real_policy_mean_min_validation_cost' in cols
    log = formats.TabularLog(str(tmp_path / 'progress.csv'))
    for it in range(3):
        for j, c in enumerate(cols):
            log.record_tabular(c, it * 100 + j)
        log.dump_tabular()
    got = formats.read_progress(str(tmp_path / 'progress.csv'))
    assert list(got.keys()) == cols
    np.testing.assert_array_equal(got['# model updates'], [1, 101, 201])
    assert formats.POLICY_AND_MODELS_CKPT % 7 == 'policy-and-models-7.ckpt' and formats.MODEL_CKPT % ('training_dynamics', 2) == 'training_dynamics_2.ckpt'
    assert formats.dynamics_variable_names(2, 3) == ['model3/layer0/weights', 'model3/layer0/biases', 'model3/layer1/weights', 'model3/layer1/biases']
