"""CPU restatement (NumPy) of the ensemble dynamics TRAINING path -- SURVEY.md section 8f rank 1+2
("next" rows): optimize_models, the replay buffer and the running normalisers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pinned against the reference's own Python (tests/golden/dyn_*.npz, made by tests/golden/make_golden.py):
`utils.data_collection` (add_data / cap / get_next_batch / sample with the exact np.random stream),
`utils.get_ith_tensor` (which rows of a reshaped batch each model trains on), `compute_baseline_loss`, and the
pair building / train-validation split / normaliser feed of `collect_data` (collect_split.npz: the reference's own
collect_data run with synthetic trajectories in place of the simulator).  The reference's one known-answer test for
this path, running_mean_std.py:44-62 (epsilon=0: mean/std equal np.mean/np.std of all rows), is restated in the tests.
Also pinned, by the reference's own TF-graph code run unmodified on the eager tf stand-in (tests/golden/make_golden_tf.py ->
tfgraph_*.npz, tests/test_oracle_tfgraph.py): the per-model prediction loss and regulariser of build_dynamics_graph
(model_based_rl.py:23-104), the gradient of that loss graph (autograd over the reference's graph), RunningMeanStd.update, and three
optimiser steps of get_dynamics_optimizer (:154-183) -- where the Adam / SGD update RULES are the stand-in's restatement of the
TF 1.4 documentation (TensorFlow itself is not installable), not TF's code.
PARITY UNPINNED: the control flow of optimize_models (restated line by line from model_based_rl.py:881-1051; it needs a live
tf.Session loop).  Self-consistency: torch autograd and torch.optim.Adam in tests/test_oracle_dynamics.py.
"""
import numpy as np
from . import metrpo_oracle as O


# ----------------------------------------------------------------------------------------
# utils.py:44-131  data_collection (FIFO replay buffer)
# ----------------------------------------------------------------------------------------
class DataCollectionOracle(object):
    def __init__(self, max_size=int(5e4)):
        self.cur_idx, self.x, self.y, self.n_data, self.max_size = 0, None, None, None, max_size

    def cap_data_size(self):
        new_start_idx = self.x.shape[0] - self.max_size
        if new_start_idx > 0:
            self.x, self.y = self.x[new_start_idx:], self.y[new_start_idx:]
            self.n_data = self.max_size
            self.cur_idx -= new_start_idx

    def clone(self, dc, first_i=None):
        assert first_i is None or first_i <= dc.n_data
        self.set_data(dc.x[:first_i], dc.y[:first_i])

    def set_data(self, x, y, is_shuffled=False, rng=np.random):
        """utils.py:67-76.  Quirks kept: cur_idx is wrapped with the UNCAPPED row count, and shuffling only permutes idx_mapping,
        which remap() ignores (:110-111) -- it consumes np.random but never changes a batch."""
        assert x.shape[0] == y.shape[0]
        self.n_data, self.x, self.y = x.shape[0], x, y
        self.cur_idx %= self.n_data
        self.cap_data_size()
        self.idx_mapping = list(range(self.n_data))
        if is_shuffled:
            rng.shuffle(self.idx_mapping)

    def add_data(self, x_new, y_new, is_shuffled=False, rng=np.random):
        assert x_new.shape[0] == y_new.shape[0]
        if self.x is not None:
            self.cur_idx = self.x.shape[0]
            self.x = np.concatenate([self.x, x_new], axis=0)
            self.y = np.concatenate([self.y, y_new], axis=0)
        else:
            self.cur_idx, self.x, self.y = 0, x_new, y_new
        self.n_data = self.x.shape[0]
        self.cap_data_size()
        self.idx_mapping = list(range(self.n_data))
        if is_shuffled:
            rng.shuffle(self.idx_mapping)

    def get_num_data(self):
        return 0 if self.n_data is None else self.n_data

    def get_next_batch(self, batch_size):
        assert batch_size <= self.n_data
        start_idx, end_idx = self.cur_idx, self.cur_idx + batch_size
        if end_idx > self.n_data:
            indices = list(range(start_idx, self.n_data)) + list(range(0, batch_size - (self.n_data - start_idx)))
            self.cur_idx = batch_size - (self.n_data - start_idx)
        else:
            indices = list(range(start_idx, end_idx))
            self.cur_idx = end_idx
        return self.x[indices, :], self.y[indices, :]

    def sample(self, batch_size, rng=np.random):
        indices = np.floor(self.n_data * rng.uniform(0.0, 1.0, size=batch_size)).astype(np.intp)
        return self.x[indices, :], self.y[indices, :]


def trajectories_to_pairs(Os, As):
    """model_based_rl.py:793-807: x = [o_t, a_t], y = o_{t+1} for t < len-1 of every trajectory, in order."""
    x_all, y_all = [], []
    for o, a in zip(Os, As):
        for t in range(len(o) - 1):
            x_all.append(np.concatenate([o[t], a[t]])); y_all.append(o[t + 1])
    return np.array(x_all), np.array(y_all)


def collect_split(x_all, y_all, data, val, splitting_mode, use_same_dataset, split_ratio, input_rms=None, output_rms=None, rng=np.random):
    """model_based_rl.py:813-852: split new (x, y) pairs into the per-scope training / validation collections
    (`data`, `val`: ordered dicts scope -> collection) and feed the normalisers from the TRAINING part only
    (only in the use_same_dataset branch, as the reference does)."""
    indices = list(range(len(x_all)))
    if splitting_mode == 'triplet':
        rng.shuffle(indices)
    else:
        assert splitting_mode == 'trajectory'
    cur_i, total = 0, len(x_all)
    for scope in data.keys():
        if use_same_dataset:
            n = round(split_ratio * total)
            val[scope].add_data(x_all[indices[:n], :], y_all[indices[:n], :])
            data[scope].add_data(x_all[indices[n:], :], y_all[indices[n:], :])
            cur_i = len(indices)
            if input_rms is not None:
                input_rms.update(x_all[indices[n:], :])
                output_rms.update(y_all[indices[n:], :] - x_all[indices[n:], :y_all.shape[1]])
        else:
            n = int(split_ratio * total / len(data.keys()))
            val[scope].add_data(x_all[indices[cur_i:cur_i + n], :], y_all[indices[cur_i:cur_i + n], :])
            cur_i += n
            m = int(total / len(data.keys()) - n)
            data[scope].add_data(x_all[indices[cur_i:cur_i + m], :], y_all[indices[cur_i:cur_i + m], :])
            cur_i += m
    assert cur_i == total


def combine_data_collections(dc1, dc2):
    """utils.py:133-142: the collection with the SMALLER max_size goes last (its rows survive the FIFO cap)."""
    out = DataCollectionOracle(max(dc1.max_size, dc2.max_size))
    if dc2.max_size < dc1.max_size:
        x, y = np.concatenate([dc1.x, dc2.x], axis=0), np.concatenate([dc1.y, dc2.y], axis=0)
    else:
        x, y = np.concatenate([dc2.x, dc1.x], axis=0), np.concatenate([dc2.y, dc1.y], axis=0)
    out.set_data(x, y)
    return out


def compute_baseline_loss(x_batch, y_batch, n_models):
    """model_based_rl.py:858-865: loss of predicting 'no change'."""
    return n_models * np.sum(np.square(y_batch - x_batch[:, :y_batch.shape[1]])) / y_batch.shape[0]


def split_batch(x_batch, y_batch, batch_size, K):
    """model_based_rl.py:961-970 + utils.get_ith_tensor (utils.py:366-369): the (batch_size*K, d) sample block is
    reshaped to (batch_size, K*d); model i trains on columns [i*d, (i+1)*d) = samples i, K+i, 2K+i, ..."""
    d, dy = x_batch.shape[1], y_batch.shape[1]
    xf = np.reshape(x_batch, (batch_size, -1)); yf = np.reshape(y_batch, (batch_size, -1))
    return [xf[:, i * d:(i + 1) * d] for i in range(K)], [yf[:, i * dy:(i + 1) * dy] for i in range(K)]


# ----------------------------------------------------------------------------------------
# running_mean_std.py:35-42  RunningMeanStd.update  (rank 2)
# ----------------------------------------------------------------------------------------
class RunningMeanStdOracle(object):
    def __init__(self, shape, epsilon=1e-2):
        self.sum = np.zeros(shape); self.sumsq = np.full(shape, epsilon); self.count = epsilon

    def update(self, x):
        self.sum += np.sum(x, axis=0); self.sumsq += np.sum(np.square(x), axis=0); self.count += len(x)

    @property
    def mean(self):
        return self.sum / self.count

    @property
    def std(self):
        return np.sqrt(np.maximum(self.sumsq / self.count - np.square(self.mean), 1e-2))


# ----------------------------------------------------------------------------------------
# model_based_rl.py:23-104 (losses), :154-183 (optimizers)
# ----------------------------------------------------------------------------------------
def xavier_uniform(rng, shape):
    """tf.contrib.layers.xavier_initializer() (uniform): limit sqrt(6/(fan_in+fan_out)); 1-D shapes have
    fan_in = fan_out = n (training.py:179,191-194 initialise the biases with it too, quirk 5)."""
    fan_in, fan_out = (shape[0], shape[0]) if len(shape) == 1 else (shape[-2], shape[-1])
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def model_forward_cache(dm, k, x):
    """forward of head k keeping the layer inputs; x = [s, a] un-normalised, (n, ns+na)."""
    s = x[:, :dm.ns]
    h = ((x - dm.in_mean) / dm.in_std)[:, dm.n_drop:]
    hs = [h]
    L = len(dm.Ws)
    for l in range(L):
        h = h @ dm.Ws[l][k] + dm.bs[l][k]
        if l < L - 1:
            h = np.maximum(h, 0)
            hs.append(h)
    return dm.diff_mean + dm.diff_std * h + s, hs


def prediction_losses(dm, xs, ys):
    """_prediction_losses[i] = mean_b sum_d (y_pred - y)^2 (model_based_rl.py:58-71); xs/ys: per-model lists."""
    return np.array([np.mean(np.sum(np.square(model_forward_cache(dm, k, xs[k])[0] - ys[k]), axis=1)) for k in range(dm.K)])


def regularizer_loss(dm, k, constant):
    """constant * sum_l (l2_loss(W_l) + l2_loss(b_l)), tf.nn.l2_loss = sum(x^2)/2 (training.py:271-282)."""
    return constant * sum(0.5 * np.sum(dm.Ws[l][k] ** 2) + 0.5 * np.sum(dm.bs[l][k] ** 2) for l in range(len(dm.Ws)))


def model_gradients(dm, k, x, y):
    """d(prediction_loss_k)/d(W_l, b_l) by back-propagation (relu hidden layers)."""
    n = x.shape[0]
    pred, hs = model_forward_cache(dm, k, x)
    dz = 2.0 * (pred - y) / n * dm.diff_std
    L = len(dm.Ws)
    gW, gb = [None] * L, [None] * L
    for l in range(L - 1, -1, -1):
        gW[l] = hs[l].T @ dz
        gb[l] = dz.sum(0)
        if l > 0:
            dz = (dz @ dm.Ws[l][k].T) * (hs[l] > 0)
    return gW, gb


class AdamState(object):
    """tf.train.AdamOptimizer(learning_rate) defaults: beta1 0.9, beta2 0.999, epsilon 1e-8."""

    def __init__(self, dm):
        self.t = 0
        self.mW = [np.zeros_like(w) for w in dm.Ws]; self.vW = [np.zeros_like(w) for w in dm.Ws]
        self.mb = [np.zeros_like(b) for b in dm.bs]; self.vb = [np.zeros_like(b) for b in dm.bs]


def train_step(dm, adam, x_batch, y_batch, batch_size, lr, reg_constant=0.0, b1=0.9, b2=0.999, eps=1e-8):
    """One sess.run([dynamics_opt_op, dynamics_loss]) (model_based_rl.py:967-971): Adam on sum_i prediction_loss_i,
    SGD(lr) on the regulariser (zero with the shipped constant 0.0).  Returns the loss BEFORE the update."""
    K = dm.K
    xs, ys = split_batch(x_batch, y_batch, batch_size, K)
    loss = float(np.sum(prediction_losses(dm, xs, ys)) + sum(regularizer_loss(dm, k, reg_constant) for k in range(K)))
    adam.t += 1
    lr_t = lr * np.sqrt(1.0 - b2 ** adam.t) / (1.0 - b1 ** adam.t)
    grads = [model_gradients(dm, k, xs[k], ys[k]) for k in range(K)]
    for l in range(len(dm.Ws)):
        for (P, M, V, gi) in ((dm.Ws[l], adam.mW[l], adam.vW[l], 0), (dm.bs[l], adam.mb[l], adam.vb[l], 1)):
            for k in range(K):
                g = grads[k][gi][l]
                M[k] = b1 * M[k] + (1 - b1) * g
                V[k] = b2 * V[k] + (1 - b2) * g * g
                P[k] = P[k] - lr_t * M[k] / (np.sqrt(V[k]) + eps) - lr * reg_constant * P[k]
    return loss


def validation_losses(dm, x_val, y_val, reg_constant=0.0):
    """dynamics_losses on np.tile(val, n_models): every model sees the whole validation set (:933-945)."""
    xs, ys = [x_val] * dm.K, [y_val] * dm.K
    return prediction_losses(dm, xs, ys) + np.array([regularizer_loss(dm, k, reg_constant) for k in range(dm.K)])


def optimize_models(dm, adam, data, val, batch_size, lr_scratch, lr_refine, max_passes, log_every_passes, num_passes_threshold,
                    reinitialize, sample_mode='random', reg_constant=0.0, rng=np.random, init_fn=None):
    """Control flow of model_based_rl.py:optimize_models for one scope: per-model best-validation snapshot / restore
    (:998-1007, 871-878), patience (:1022-1031), optional scratch->refine learning-rate switch (:1024-1030)."""
    K = dm.K
    lr = lr_scratch if reinitialize else lr_refine
    if reinitialize and init_fn is not None:
        init_fn(dm)
    adam.__init__(dm)
    snap = [[w.copy() for w in dm.Ws], [b.copy() for b in dm.bs]]
    min_losses = validation_losses(dm, val.x, val.y, reg_constant)
    min_sum = float(np.sum(min_losses))
    recover_indices, refine_idx, best_j = np.zeros(K), -1, 0
    iter_const = data.n_data / batch_size
    max_iters, log_every = int(max_passes * iter_const), int(log_every_passes * iter_const)
    num_iters_threshold = int(num_passes_threshold * iter_const)
    j = 0
    for j in range(1, max_iters + 1):
        xb, yb = data.get_next_batch(batch_size * K) if sample_mode == 'next_batch' else data.sample(batch_size * K, rng)
        train_step(dm, adam, xb, yb, batch_size, lr, reg_constant)
        if j % log_every == 0:
            losses = validation_losses(dm, val.x, val.y, reg_constant)
            vsum = float(np.sum(losses))
            if min_sum > vsum:
                min_sum, best_j = vsum, j
            to_update = min_losses > losses
            min_losses[to_update] = losses[to_update]
            for i in np.nonzero(to_update)[0]:
                for l in range(len(dm.Ws)):
                    snap[0][l][i] = dm.Ws[l][i].copy(); snap[1][l][i] = dm.bs[l][i].copy()
                recover_indices[i] = j
            if j - max(np.amax(recover_indices), refine_idx) >= num_iters_threshold:
                if reinitialize and refine_idx < 0 and lr_scratch > lr_refine:
                    for l in range(len(dm.Ws)):
                        dm.Ws[l][:] = snap[0][l]; dm.bs[l][:] = snap[1][l]
                    lr, refine_idx = lr_refine, j
                    continue
                break
    for l in range(len(dm.Ws)):
        dm.Ws[l][:] = snap[0][l]; dm.bs[l][:] = snap[1][l]
    return dict(n_updates=j, best_index=best_j, min_validation_losses=min_losses, min_sum_validation_loss=min_sum,
                recover_indices=recover_indices)
