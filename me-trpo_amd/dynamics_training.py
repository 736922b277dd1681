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
        elif x.shape[1] != self._xs.shape[1] or y.shape[1] != self._ys.shape[1]:
            raise AssertionError("feature width changed: buffer holds (%d, %d) columns, got (%d, %d)"
                                 % (self._xs.shape[1], self._ys.shape[1], x.shape[1], y.shape[1]))

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

    # .x / .y: the buffer's rows in logical order, as READ-ONLY snapshots (the reference exposes its arrays; writing through these
    # would not reach the ring once it has wrapped).  While the oldest row is physical row 0 they are views (no copy).
    def _logical(self, store):
        if store is None:
            return None
        if self._head == 0:
            return store[:self.n_data]
        return store.index_select(0, self._physical(np.arange(self.n_data)))

    @property
    def x(self):
        return self._logical(self._xs)

    @property
    def y(self):
        return self._logical(self._ys)

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


def _split_plan(total, n_scopes, shared, split_ratio):
    """Row ranges [lo, hi) of the (possibly shuffled) sample order that go to each scope's validation and training buffers.
    shared: every scope gets the same split (validation = the first round(ratio * total) rows); otherwise the order is cut into
    n_scopes consecutive (validation, training) pairs of int(ratio * total / n) and int(total / n - validation) rows."""
    if shared:
        cut = round(split_ratio * total)
        return [((0, cut), (cut, total))] * n_scopes, total
    n_val = int(split_ratio * total / n_scopes)
    n_trn = int(total / n_scopes - n_val)
    plan, at = [], 0
    for _ in range(n_scopes):
        plan.append(((at, at + n_val), (at + n_val, at + n_val + n_trn)))
        at += n_val + n_trn
    return plan, at


def add_rollouts(x_all, y_all, dynamics_data, dynamics_validation, splitting_mode, use_same_dataset, split_ratio, input_rms=None,
                 output_rms=None):
    """What collect_data does with the simulator's samples (model_based_rl.py:813-852): one np.random.shuffle of the sample
    order in "triplet" mode (none in "trajectory" mode), a validation / training cut per scope, and -- in the shared-dataset
    branch only, as in the reference -- the running normalisers fed from the training rows (inputs, and output minus state).
    The simulator call itself (sample_trajectories) stays with the caller: it needs MuJoCo."""
    if splitting_mode not in ("trajectory", "triplet"):
        raise AssertionError("splitting_mode must be 'trajectory' or 'triplet'")
    x_all, y_all = np.asarray(x_all), np.asarray(y_all)
    total = len(x_all)
    order = np.arange(total)
    if splitting_mode == "triplet":
        np.random.shuffle(order)
    scopes = list(dynamics_data.keys())
    plan, consumed = _split_plan(total, len(scopes), bool(use_same_dataset), split_ratio)
    assert consumed == total, "sample count must split evenly over the scopes (model_based_rl.py:852)"
    for scope, ((v_lo, v_hi), (t_lo, t_hi)) in zip(scopes, plan):
        val_rows, trn_rows = order[v_lo:v_hi], order[t_lo:t_hi]
        dynamics_validation[scope].add_data(x_all[val_rows], y_all[val_rows])
        dynamics_data[scope].add_data(x_all[trn_rows], y_all[trn_rows])
        if use_same_dataset and input_rms is not None:
            input_rms.update(x_all[trn_rows])
            output_rms.update(y_all[trn_rows] - x_all[trn_rows, :y_all.shape[1]])


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


class _BestPerModel(object):
    """Per-model best-validation bookkeeping of optimize_models, on device tensors: the lowest validation loss each of the K models
    has reached, the parameters it had then (the reference's per-model tf.train.Saver, model_based_rl.py:927-930, 1002-1004) and the
    update count at which that happened."""

    def __init__(self, engine, first_losses):
        self.engine = engine
        self.params = engine.get_dynamics().clone()                        # [K, P_dyn]
        self.loss = first_losses.clone()                                   # [K] float64
        self.when = torch.zeros(engine.K, dtype=torch.int64, device=first_losses.device)

    def offer(self, update, losses):
        """Keep the models that improved; returns nothing (no host sync)."""
        better = losses < self.loss
        self.loss = torch.where(better, losses, self.loss)
        self.params = torch.where(better[:, None], self.engine.get_dynamics(), self.params)
        self.when = torch.where(better, torch.full_like(self.when, update), self.when)

    def restore(self):
        """recover_weights (model_based_rl.py:871-878): every model goes back to its own best parameters."""
        for k in range(self.engine.K):
            self.engine.set_dynamics_model(k, self.params[k])


def optimize_models(engine, dynamics_data, dynamics_validation, learning_rate, batch_size=1000, max_passes=2000, log_every=5,
                    num_passes_threshold=25, reinitialize=False, sample_mode='random', reg_constant=0.0, init_seed=None, logger=None):
    """One scope of model_based_rl.py:881-1051.  Every model trains on its own slice of each K * batch_size draw (Adam on the
    prediction loss, SGD on the regulariser: dyn_train.hip); every `log_every` passes the K validation losses are evaluated, each
    model keeps the parameters of its own best validation loss, and training stops once no model has improved for
    `num_passes_threshold` passes -- after one switch from the "scratch" to the smaller "refine" learning rate (restarting from the
    best parameters) when the ensemble was re-initialised.  Returns the reference's bookkeeping plus the number of updates."""
    if sample_mode not in ('random', 'next_batch'):
        raise AssertionError("sample_mode must be 'random' or 'next_batch'")
    K = engine.K
    rate = learning_rate["scratch"] if reinitialize else learning_rate["refine"]
    may_refine = bool(reinitialize) and learning_rate["scratch"] > learning_rate["refine"]
    if reinitialize:
        xavier_reinitialize(engine, init_seed)
    engine.train_reset()
    x_val, y_val = dynamics_validation.x, dynamics_validation.y           # every model is validated on all of it
    best = _BestPerModel(engine, engine.eval_losses(x_val, y_val, reg_constant))
    best_sum, best_sum_at = float(best.loss.sum().item()), 0
    updates_per_pass = dynamics_data.n_data / batch_size
    n_updates_max = int(max_passes * updates_per_pass)
    eval_period = max(1, int(log_every * updates_per_pass))
    patience = int(num_passes_threshold * updates_per_pass)
    refined_at = -1
    history_train, history_val = [], []
    done = 0
    for done in range(1, n_updates_max + 1):
        if sample_mode == 'next_batch':
            xb, yb = dynamics_data.get_next_batch(batch_size * K, is_shuffled=False)
        else:
            xb, yb = dynamics_data.sample(batch_size * K)
        evaluate = (done % eval_period == 0)
        train_losses = engine.train_step(xb, yb, batch_size, rate, reg_constant, want_loss=evaluate)
        if not evaluate:
            continue
        val_losses = engine.eval_losses(x_val, y_val, reg_constant)
        best.offer(done, val_losses)
        # one read-back per evaluation: summed training loss, summed validation loss, update count of the latest improvement
        t_sum, v_sum, latest = (float(v) for v in torch.stack([train_losses.sum(), val_losses.sum(), best.when.max().double()]).cpu())
        history_train.append(t_sum); history_val.append(v_sum)
        if logger:
            logger('iter %d train %.5f val %.5f' % (done, t_sum, v_sum))
        if v_sum < best_sum:
            best_sum, best_sum_at = v_sum, done
        if done - max(latest, refined_at) < patience:
            continue
        if may_refine and refined_at < 0:                                  # stalled at the scratch rate: best parameters, smaller rate
            best.restore()
            rate, refined_at = learning_rate["refine"], done
            continue
        break
    best.restore()
    return {'training_losses': history_train, 'validation_losses': history_val, 'best_index': best_sum_at,
            'n_model_updates': done, 'min_sum_validation_loss': best_sum,
            'min_validation_losses': best.loss.cpu().numpy(), 'recover_indices': best.when.cpu().numpy().astype(np.float64)}
