"""Dynamics-training oracle: pinned parts against the reference's own Python (tests/golden/dyn_data.npz),
unpinned parts (loss, gradient, Adam) against torch autograd and a hand-rolled TF-style Adam.  CPU only."""
import numpy as np
import pytest
import torch
from conftest import load_golden
from oracle import metrpo_oracle as O
from oracle import dynamics_oracle as D


def test_data_collection_matches_reference():
    d = load_golden('dyn_data')
    dc = D.DataCollectionOracle(max_size=50)
    np.random.seed(int(d['seed']))
    ai = 0
    for i, (op, n) in enumerate(zip(d['ops'], d['ns'])):
        if op == 0:
            dc.add_data(d['addx%d' % ai], d['addy%d' % ai]); ai += 1
            xb, yb = dc.x, dc.y
        elif op == 1:
            xb, yb = dc.get_next_batch(int(n))
        else:
            xb, yb = dc.sample(int(n))
        np.testing.assert_array_equal(xb, d['x%d' % i]); np.testing.assert_array_equal(yb, d['y%d' % i])
        assert dc.n_data == int(d['n_data'][i]) and dc.cur_idx == int(d['cur_idx'][i])


def test_batch_split_and_baseline_loss_match_reference():
    d = load_golden('dyn_data')
    K, bs = int(d['split_K']), int(d['split_bs'])
    xs, ys = D.split_batch(d['split_x'], d['split_y'], bs, K)
    for i in range(K):
        np.testing.assert_array_equal(xs[i], d['mx%d' % i]); np.testing.assert_array_equal(ys[i], d['my%d' % i])
        np.testing.assert_array_equal(xs[i], d['split_x'][i::K])          # model i trains on samples i, K+i, 2K+i, ...
    assert abs(D.compute_baseline_loss(d['split_x'], d['split_y'], K) - float(d['baseline_loss'])) < 1e-12


def _problem(env='swimmer', K=3, hidden=(16, 12), seed=0, n=40):
    dm, _, _, _ = O.make_problem(env, K=K, dyn_hidden=hidden, pol_hidden=(8, 8), seed=seed)
    rng = np.random.RandomState(seed)
    dm.in_mean = rng.randn(dm.ns + dm.na) * 0.1; dm.in_std = np.abs(1 + rng.randn(dm.ns + dm.na) * 0.2)
    dm.diff_mean = rng.randn(dm.ns) * 0.01; dm.diff_std = np.abs(0.1 + rng.randn(dm.ns) * 0.02)
    x = rng.randn(n * K, dm.ns + dm.na); y = x[:, :dm.ns] + rng.randn(n * K, dm.ns) * 0.1
    return dm, x, y, n


def _torch_loss(dm, params, k, x, y):
    Ws, bs = params
    h = ((torch.from_numpy(x) - torch.from_numpy(dm.in_mean)) / torch.from_numpy(dm.in_std))[:, dm.n_drop:]
    L = len(Ws)
    for l in range(L):
        h = h @ Ws[l][k] + bs[l][k]
        if l < L - 1:
            h = torch.relu(h)
    pred = torch.from_numpy(dm.diff_mean) + torch.from_numpy(dm.diff_std) * h + torch.from_numpy(x[:, :dm.ns])
    return ((pred - torch.from_numpy(y)) ** 2).sum(1).mean()


def test_loss_and_gradient_match_autograd():
    dm, x, y, n = _problem()
    xs, ys = D.split_batch(x, y, n, dm.K)
    Ws = [torch.tensor(w, requires_grad=True) for w in dm.Ws]; bs = [torch.tensor(b, requires_grad=True) for b in dm.bs]
    total = sum(_torch_loss(dm, (Ws, bs), k, xs[k], ys[k]) for k in range(dm.K))
    total.backward()
    np.testing.assert_allclose(D.prediction_losses(dm, xs, ys).sum(), total.item(), rtol=1e-12)
    for k in range(dm.K):
        gW, gb = D.model_gradients(dm, k, xs[k], ys[k])
        for l in range(len(dm.Ws)):
            np.testing.assert_allclose(gW[l], Ws[l].grad[k].numpy(), rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(gb[l], bs[l].grad[k].numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('reg', [0.0, 1e-3])
def test_train_steps_match_tf_style_adam_on_autograd(reg):
    dm, x, y, n = _problem(seed=1)
    Ws = [torch.tensor(w.copy(), requires_grad=True) for w in dm.Ws]; bs = [torch.tensor(b.copy(), requires_grad=True) for b in dm.bs]
    state = {}
    adam = D.AdamState(dm)
    rng = np.random.RandomState(2)
    for it in range(5):
        idx = rng.randint(len(x), size=n * dm.K)
        xb, yb = x[idx], y[idx]
        xs, ys = D.split_batch(xb, yb, n, dm.K)
        for p in Ws + bs:
            p.grad = None
        sum(_torch_loss(dm, (Ws, bs), k, xs[k], ys[k]) for k in range(dm.K)).backward()
        t = it + 1
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)              # tf.train.AdamOptimizer: epsilon outside the bias correction
        with torch.no_grad():
            for i, p in enumerate(Ws + bs):
                st = state.setdefault(i, dict(m=torch.zeros_like(p), v=torch.zeros_like(p)))
                st['m'] = 0.9 * st['m'] + 0.1 * p.grad; st['v'] = 0.999 * st['v'] + 0.001 * p.grad ** 2
                p -= lr_t * st['m'] / (st['v'].sqrt() + 1e-8) + 1e-3 * reg * p
        assert np.isfinite(D.train_step(dm, adam, xb, yb, n, 1e-3, reg_constant=reg))
    for l in range(len(dm.Ws)):
        np.testing.assert_allclose(dm.Ws[l], Ws[l].detach().numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(dm.bs[l], bs[l].detach().numpy(), rtol=1e-9, atol=1e-12)


def run_reference_buffer_tests(make, combine, to_np=np.asarray):
    """The bodies of the reference's OWN tests utils.test_data_collection / test_combine_data_collection (utils.py:145-176),
    replayed on `make(max_size)` collections; every get_next_batch result and the combined collections are compared with what the
    reference's classes produced (tests/golden/dyn_buffer_reftests.npz)."""
    d = load_golden('dyn_buffer_reftests')
    np.random.seed(int(d['seed']))
    got = []
    dc = make(3)
    x = np.array([[1, 2], [3, 4], [5, 6], [7, 8]])
    def nb(n):
        xb, yb = dc.get_next_batch(n)
        got.append((to_np(xb), to_np(yb), dc.cur_idx, dc.n_data))
    dc.set_data(x, x); nb(2); nb(2); nb(2)
    dc.set_data(x, x, True); nb(2); nb(2); nb(2)
    x_new = np.array([[0, 0]])
    dc.add_data(x_new, x_new); nb(2); nb(2); nb(2)
    x_new = np.array([[9, 9], [8, 8]])
    dc.add_data(x_new, x_new, True); nb(2); nb(2); nb(2)
    assert len(got) == int(d['n_batches'])
    for i, (xb, yb, cur, n) in enumerate(got):
        np.testing.assert_array_equal(xb, d['bx%d' % i]); np.testing.assert_array_equal(yb, d['by%d' % i])
        assert cur == int(d['cur%d' % i]) and n == int(d['n%d' % i])
    dc1 = make(10); xx = np.reshape(np.arange(20), (10, 2)); dc1.set_data(xx, xx)
    yy = np.reshape(-np.arange(10), (5, 2)); dc2 = make(5); dc2.set_data(yy, yy)
    a, b = combine(dc1, dc2), combine(dc2, dc1)
    for c, k in ((a, 'a'), (b, 'b')):
        np.testing.assert_array_equal(to_np(c.x), d['comb_%sx' % k]); np.testing.assert_array_equal(to_np(c.y), d['comb_%sy' % k])
        assert c.n_data == int(d['comb_%s_n' % k]) and c.cur_idx == int(d['comb_%s_cur' % k]) and c.max_size == int(d['comb_%s_max' % k])
    assert np.random.randint(1 << 30) == int(d['rng_after'])          # the shuffles consumed exactly the reference's share of np.random


def test_reference_own_buffer_tests_on_the_oracle():
    run_reference_buffer_tests(D.DataCollectionOracle, D.combine_data_collections)


class _Rec(object):
    def __init__(self): self.calls = []
    def update(self, x): self.calls.append(np.array(x))


def collect_split_cases():
    """Replays tests/golden/collect_split.npz: yields (case dict, make_collections, check) pieces shared with the GPU test."""
    d = load_golden('collect_split')
    lens = d['lens']; off = np.concatenate([[0], np.cumsum(lens)])
    Os = [d['O'][off[i]:off[i + 1]] for i in range(len(lens))]; As = [d['A'][off[i]:off[i + 1]] for i in range(len(lens))]
    for ci in range(int(d['n_cases'])):
        yield d, ci, Os, As, dict(mode='trajectory' if int(d['c%d_mode' % ci]) == 0 else 'triplet', same=bool(d['c%d_same' % ci]),
                                  ratio=float(d['c%d_ratio' % ci]), n_scopes=int(d['c%d_scopes' % ci]), seed=int(d['c%d_seed' % ci]))


def test_collect_split_matches_reference():
    """the reference's collect_data (model_based_rl.py:793-852) on synthetic trajectories vs the restatement."""
    from collections import OrderedDict
    for d, ci, Os, As, c in collect_split_cases():
        x_all, y_all = D.trajectories_to_pairs(Os, As)
        scopes = ['s%d' % i for i in range(c['n_scopes'])]
        data = OrderedDict((sc, D.DataCollectionOracle(1000)) for sc in scopes); val = OrderedDict((sc, D.DataCollectionOracle(1000)) for sc in scopes)
        irms, orms = _Rec(), _Rec()
        np.random.seed(c['seed'])
        D.collect_split(x_all, y_all, data, val, c['mode'], c['same'], c['ratio'], irms, orms)
        for si, sc in enumerate(scopes):
            np.testing.assert_array_equal(data[sc].x, d['c%d_s%d_tx' % (ci, si)]); np.testing.assert_array_equal(data[sc].y, d['c%d_s%d_ty' % (ci, si)])
            np.testing.assert_array_equal(val[sc].x, d['c%d_s%d_vx' % (ci, si)]); np.testing.assert_array_equal(val[sc].y, d['c%d_s%d_vy' % (ci, si)])
        assert len(irms.calls) == int(d['c%d_n_rms' % ci])
        for k in range(len(irms.calls)):
            np.testing.assert_array_equal(irms.calls[k], d['c%d_rms_in%d' % (ci, k)]); np.testing.assert_array_equal(orms.calls[k], d['c%d_rms_out%d' % (ci, k)])


def test_rms_update_is_mean_std_of_concatenation():
    """running_mean_std.py:44-60 test_runningmeanstd (epsilon 0 form): matches np.mean/np.std of the concatenation."""
    rng = np.random.RandomState(0)
    x, y = rng.randn(1000, 3) + 2.0, rng.randn(1000, 3) * 3.0 + 1.0
    r = D.RunningMeanStdOracle((3,), epsilon=0.0)
    r.update(x); r.update(y)
    z = np.concatenate([x, y])
    np.testing.assert_allclose(r.mean, z.mean(0)); np.testing.assert_allclose(r.std, z.std(0))
    r2 = D.RunningMeanStdOracle((3,))                  # shipped epsilon 1e-2 and the 0.1 std floor
    assert np.allclose(r2.std, 1.0) and np.allclose(r2.mean, 0.0)


def test_optimize_models_learns_and_restores_best():
    env, K = 'swimmer', 3
    dm, _, _, _ = O.make_problem(env, K=K, dyn_hidden=(16, 16), pol_hidden=(8, 8), seed=3)
    true, _, _, _ = O.make_problem(env, K=1, dyn_hidden=(16, 16), pol_hidden=(8, 8), seed=9)
    rng = np.random.RandomState(0)

    def gen(n):
        x = rng.randn(n, dm.ns + dm.na) * 0.5; x[:, dm.ns:] = np.clip(x[:, dm.ns:], -1, 1)
        return x, O.dynamics_forward(true, 0, x[:, :dm.ns], x[:, dm.ns:])
    data, val = D.DataCollectionOracle(10 ** 6), D.DataCollectionOracle(10 ** 6)
    data.add_data(*gen(600)); val.add_data(*gen(200))
    before = D.validation_losses(dm, val.x, val.y).copy()
    np.random.seed(0)
    out = D.optimize_models(dm, D.AdamState(dm), data, val, batch_size=50, lr_scratch=3e-3, lr_refine=1e-3, max_passes=30,
                            log_every_passes=1, num_passes_threshold=5, reinitialize=False)
    after = D.validation_losses(dm, val.x, val.y)
    assert (after < 0.5 * before).all()
    np.testing.assert_allclose(after, out['min_validation_losses'], rtol=1e-12)      # weights restored to each model's best
