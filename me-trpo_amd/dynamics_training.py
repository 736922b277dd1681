"""Ensemble dynamics training, replay buffer and running normalisers -- the "next" rows of the scope table
(SURVEY.md 8f rank 1-2), mirroring the reference's interface:

    data_collection(max_size)                       utils.py:44-131     (FIFO replay buffer; add_data / get_next_batch / sample)
    RunningMeanStd(epsilon, shape)                  running_mean_std.py (sum / sumsq / count; mean; std floored at 0.1)
    optimize_models(...)                            model_based_rl.py:881-1051 (one scope)
    trajectories_to_pairs / add_rollouts            model_based_rl.py:793-852  (collect_data after the simulator call)

Data lives on the GPU; batch index selection keeps the reference's host-side NumPy semantics (np.random.uniform stream),
the gather, the K-head forward/backward/Adam step and the validation losses run in libmetrpo.so (dyn_train.hip)."""
import numpy as np
import torch


class data_collection(object):
    """FIFO replay buffer with the interface and the index stream of the reference's `data_collection` (utils.py:44-131),
    stored as a PREALLOCATED RING in HBM: two [max_size, d] device tensors, a physical `_head` (row of the oldest sample) and
    `n_data`.  Appending writes at most two contiguous slices and evicting the oldest rows moves `_head` -- nothing is
    re-allocated or shifted (the reference concatenates and re-slices the whole array on every add).  Logical row i (0 = oldest)
    lives at physical row (_head + i) % max_size; `cur_idx`, the batches of get_next_batch / sample and `.x` / `.y` are all in
    logical order, so every consumer sees exactly what the reference's arrays would hold
    (tests/golden/dyn_data.npz, dyn_buffer_reftests.npz)."""

    def __init__(self, max_size=int(5e4), device=None):
        self.max_size, self.device = int(max_size), device
        self.cur_idx, self.n_data = 0, None
        self._head, self._xs, self._ys = 0, None, None
        self.idx_mapping = []

    # ---- storage
    def _dev(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, torch.float32)
        return torch.as_tensor(np.asarray(a), dtype=torch.float32, device=self.device)

    def _ensure_storage(self, x, y):
        if self._xs is None:
            self._xs = torch.empty(self.max_size, x.shape[1], dtype=torch.float32, device=x.device)
            self._ys = torch.empty(self.max_size, y.shape[1], dtype=torch.float32, device=y.device)

    def _append(self, x, y):
        """Write rows after the newest sample; returns how many of the OLDEST logical rows (of old + new) fell out."""
        cap, held, m = self.max_size, self.n_data or 0, x.shape[0]
        if m >= cap:                                           # the new block alone fills the ring
            self._xs.copy_(x[m - cap:]); self._ys.copy_(y[m - cap:])
            self._head, self.n_data = 0, cap
            return held + m - cap
        w = (self._head + held) % cap                          # first free physical row
        first = min(m, cap - w)
        self._xs[w:w + first] = x[:first]; self._ys[w:w + first] = y[:first]
        if first < m:
            self._xs[:m - first] = x[first:]; self._ys[:m - first] = y[first:]
        dropped = max(0, held + m - cap)
        self._head = (self._head + dropped) % cap
        self.n_data = held + m - dropped
        return dropped

    def _physical(self, logical):
        idx = np.asarray(logical, dtype=np.int64)
        idx = np.where(idx < 0, idx + self.n_data, idx)        # NumPy wrap: the reference can hold cur_idx = -1 after a capped set_data
        if np.any((idx < 0) | (idx >= self.n_data)):
            raise IndexError("replay-buffer index out of range")
        return torch.as_tensor((idx + self._head) % self.max_size, device=self._xs.device)

    def _rows(self, logical):
        phys = self._physical(logical)
        return self._xs.index_select(0, phys), self._ys.index_select(0, phys)

    @property
    def x(self):
        return None if self._xs is None else self._rows(np.arange(self.n_data))[0]

    @property
    def y(self):
        return None if self._ys is None else self._rows(np.arange(self.n_data))[1]

    def _fresh_mapping(self, is_shuffled):
        self.idx_mapping = list(range(self.n_data))
        if is_shuffled:                                        # permutes idx_mapping only; batches never read it (utils.py:110-111)
            self.reshuffle_indices()

    # ---- reference interface
    def add_data(self, x_new, y_new, is_shuffled=False):
        """utils.py:78-90: the cursor moves to the first new row, then follows the eviction."""
        x_new, y_new = self._dev(x_new), self._dev(y_new)
        if x_new.shape[0] != y_new.shape[0]:
            raise AssertionError("x and y row counts differ")
        self._ensure_storage(x_new, y_new)
        self.cur_idx = self.n_data or 0
        self.cur_idx -= self._append(x_new, y_new)
        self._fresh_mapping(is_shuffled)

    def set_data(self, x, y, is_shuffled=False):
        """utils.py:67-76.  Kept quirk: the cursor is wrapped with the row count BEFORE the cap is applied."""
        x, y = self._dev(x), self._dev(y)
        if x.shape[0] != y.shape[0]:
            raise AssertionError("x and y row counts differ")
        self._ensure_storage(x, y)
        self.cur_idx %= x.shape[0]
        self.n_data, self._head = 0, 0
        self.cur_idx -= self._append(x, y)
        self._fresh_mapping(is_shuffled)

    def clone(self, dc, first_i=None):
        assert first_i is None or first_i <= dc.n_data, "Not enough data for first_i."
        self.set_data(dc.x[:first_i], dc.y[:first_i])

    def cap_data_size(self):
        """The ring never exceeds max_size; kept for interface parity (utils.py:55-61)."""

    def reshuffle_indices(self):
        np.random.shuffle(self.idx_mapping)

    def reshuffle_data(self):
        order = list(range(self.n_data))
        np.random.shuffle(order)
        x, y = self._rows(order)
        self._head = 0
        self._xs[:self.n_data] = x; self._ys[:self.n_data] = y

    def get_num_data(self):
        return self.n_data or 0

    def get_next_batch(self, batch_size, is_shuffled=False):
        """Sequential window of `batch_size` logical rows starting at the cursor, wrapping to row 0 (utils.py:113-125)."""
        n = self.n_data
        assert batch_size <= n, "Batch size %d is larger than n_data %d" % (batch_size, n)
        stop = self.cur_idx + batch_size
        if stop > n:
            window = np.concatenate([np.arange(self.cur_idx, n), np.arange(0, stop - n)])
            self.cur_idx = stop - n
        else:
            window = np.arange(self.cur_idx, stop)
            self.cur_idx = stop
        return self._rows(window)

    def sample(self, batch_size):
        """Uniform with replacement from one np.random.uniform call, floor(n*u) -- the draw of utils.py:127-129."""
        u = np.random.uniform(0.0, 1.0, size=batch_size)
        return self._rows(np.floor(self.n_data * u).astype(np.int64))


def combine_data_collections(dc1, dc2):
    """utils.py:133-142."""
    out = data_collection(max_size=max(dc1.max_size, dc2.max_size), device=dc1.device)
    if dc2.max_size < dc1.max_size:
        x, y = torch.cat([dc1.x, dc2.x]), torch.cat([dc1.y, dc2.y])
    else:
        x, y = torch.cat([dc2.x, dc1.x]), torch.cat([dc2.y, dc1.y])
    out.set_data(x, y)
    return out


class RunningMeanStd(object):
    def __init__(self, engine, epsilon=1e-2, shape=()):
        self.engine = engine
        n = int(np.prod(shape))
        self._sum = torch.zeros(n, dtype=torch.float64, device=engine.device)
        self._sumsq = torch.full((n,), float(epsilon), dtype=torch.float64, device=engine.device)
        self._count = float(epsilon)

    def update(self, x):
        x = torch.as_tensor(x, device=self.engine.device)
        self.engine.rms_accumulate(x, self._sum, self._sumsq)
        self._count += x.shape[0]

    @property
    def mean(self):
        return (self._sum / self._count).to(torch.float32)

    @property
    def std(self):
        m = self._sum / self._count
        return torch.sqrt(torch.clamp(self._sumsq / self._count - m * m, min=1e-2)).to(torch.float32)


def trajectories_to_pairs(Os, As):
    """model_based_rl.py:793-807: x = [o_t, a_t], y = o_{t+1} for every t < len-1 of every trajectory (host arrays in)."""
    x_all = np.concatenate([np.concatenate([np.asarray(o)[:-1], np.asarray(a)[:-1]], axis=1) for o, a in zip(Os, As)])
    y_all = np.concatenate([np.asarray(o)[1:] for o in Os])
    return x_all, y_all


def add_rollouts(x_all, y_all, dynamics_data, dynamics_validation, splitting_mode, use_same_dataset, split_ratio, input_rms=None,
                 output_rms=None):
    """The post-simulator half of collect_data (model_based_rl.py:813-852): np.random.shuffle of the indices in "triplet"
    mode, per-scope validation / training split, RunningMeanStd fed from the training part (shared-dataset branch only,
    as in the reference).  The simulator half (sample_trajectories) stays with the caller -- it needs MuJoCo."""
    x_all, y_all = np.asarray(x_all), np.asarray(y_all)
    indices = list(range(len(x_all)))
    if splitting_mode == "triplet":
        np.random.shuffle(indices)
    elif splitting_mode != "trajectory":
        raise AssertionError("splitting_mode must be 'trajectory' or 'triplet'")
    cur_i, total = 0, len(x_all)
    n_scopes = len(dynamics_data.keys())
    for scope in dynamics_data.keys():
        if use_same_dataset:
            n = round(split_ratio * total)
            dynamics_validation[scope].add_data(x_all[indices[:n], :], y_all[indices[:n], :])
            dynamics_data[scope].add_data(x_all[indices[n:], :], y_all[indices[n:], :])
            cur_i = len(indices)
            if input_rms is not None:
                input_rms.update(x_all[indices[n:], :])
                output_rms.update(y_all[indices[n:], :] - x_all[indices[n:], :y_all.shape[1]])
        else:
            n = int(split_ratio * total / n_scopes)
            dynamics_validation[scope].add_data(x_all[indices[cur_i:cur_i + n], :], y_all[indices[cur_i:cur_i + n], :])
            cur_i += n
            m = int(total / n_scopes - n)
            dynamics_data[scope].add_data(x_all[indices[cur_i:cur_i + m], :], y_all[indices[cur_i:cur_i + m], :])
            cur_i += m
    assert cur_i == total, "sample count must split evenly over the scopes (model_based_rl.py:852)"


def push_normalizers(engine, input_rms, diff_rms):
    """training.py:228,257: the dynamics model reads input_rms.mean/std and diff_rms.mean/std[:ns]."""
    ns = engine.ns
    engine.set_normalizers(input_rms.mean, input_rms.std, diff_rms.mean[:ns], diff_rms.std[:ns])


def xavier_reinitialize(engine, seed=None):
    """dynamics_initializer (model_based_rl.py:906-918): Xavier-uniform weights AND biases (training.py:179,191-194)."""
    rng = np.random.RandomState(seed)
    dims = [engine.ns + engine.na - engine.n_drop] + list(engine.dyn_hidden) + [engine.ns]
    parts = []
    for l in range(len(dims) - 1):
        lim_w = np.sqrt(6.0 / (dims[l] + dims[l + 1])); lim_b = np.sqrt(6.0 / (2 * dims[l + 1]))
        parts += [rng.uniform(-lim_w, lim_w, size=(engine.K, dims[l] * dims[l + 1])), rng.uniform(-lim_b, lim_b, size=(engine.K, dims[l + 1]))]
    params = torch.as_tensor(np.concatenate(parts, axis=1), dtype=torch.float32, device=engine.device)
    for k in range(engine.K):
        engine.set_dynamics_model(k, params[k])


def optimize_models(engine, dynamics_data, dynamics_validation, learning_rate, batch_size=1000, max_passes=2000, log_every=5,
                    num_passes_threshold=25, reinitialize=False, sample_mode='random', reg_constant=0.0, init_seed=None, logger=None):
    """model_based_rl.py:881-1051 for one scope.  learning_rate = {"scratch":..., "refine":...}; returns the reference's
    bookkeeping (training/validation losses, best index) plus '# model updates'."""
    K = engine.K
    lr = learning_rate
    if reinitialize:
        cur_lr = lr["scratch"]
        xavier_reinitialize(engine, init_seed)
    else:
        cur_lr = lr["refine"]
    engine.train_reset()                                                   # dynamics_adam_init
    snapshot = engine.get_dynamics().clone()                               # savers[scope][i].save(...), :927-930
    x_val, y_val = dynamics_validation.x, dynamics_validation.y           # np.tile(val, n_models): every model sees all of it
    min_validation_losses = engine.eval_losses(x_val, y_val, reg_constant).cpu().numpy()
    min_sum_validation_loss = float(np.sum(min_validation_losses))
    recover_indices, refine_idx, best_j = np.zeros(K), -1, 0
    training_losses, validation_losses = [], []
    iter_const = dynamics_data.n_data / batch_size
    max_iters = int(max_passes * iter_const)
    log_every_it = max(1, int(log_every * iter_const))
    num_iters_threshold = int(num_passes_threshold * iter_const)
    j = 0
    for j in range(1, max_iters + 1):
        if sample_mode == 'next_batch':
            x_batch, y_batch = dynamics_data.get_next_batch(batch_size * K, is_shuffled=False)
        else:
            assert sample_mode == 'random'
            x_batch, y_batch = dynamics_data.sample(batch_size * K)
        want = (j % log_every_it == 0)
        tl = engine.train_step(x_batch, y_batch, batch_size, cur_lr, reg_constant, want_loss=want)
        if want:                                                           # validation and logging, :974-1031
            training_losses.append(float(tl.sum().item()))
            _validation_losses = engine.eval_losses(x_val, y_val, reg_constant).cpu().numpy()
            validation_loss = float(np.sum(_validation_losses))
            validation_losses.append(validation_loss)
            if logger:
                logger('iter %d train %.5f val %.5f' % (j, training_losses[-1], validation_loss))
            if min_sum_validation_loss > validation_loss:
                min_sum_validation_loss, best_j = validation_loss, j
            to_update = min_validation_losses > _validation_losses
            min_validation_losses[to_update] = _validation_losses[to_update]
            if to_update.any():
                cur = engine.get_dynamics()
                for i in np.nonzero(to_update)[0]:
                    snapshot[i].copy_(cur[i])                              # per-model saver, :1002-1004
                    recover_indices[i] = j
            if j - max(np.amax(recover_indices), refine_idx) >= num_iters_threshold:
                if reinitialize and refine_idx < 0 and lr["scratch"] > lr["refine"]:
                    for i in range(K):                                     # recover_weights, then refine with the smaller rate
                        engine.set_dynamics_model(i, snapshot[i])
                    cur_lr, refine_idx = lr["refine"], j
                    continue
                break
    for i in range(K):                                                     # recover_weights (:1034), :871-878
        engine.set_dynamics_model(i, snapshot[i])
    return {'training_losses': training_losses, 'validation_losses': validation_losses, 'best_index': best_j,
            'n_model_updates': j, 'min_sum_validation_loss': min_sum_validation_loss,
            'min_validation_losses': min_validation_losses, 'recover_indices': recover_indices}
