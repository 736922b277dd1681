"""Wire / on-disk formats at the boundary of the inner loop, so that the accelerated loop can be driven by,
or compared against, artefacts of a reference run:

    new_rollouts_%d.pkl            pickle.dump((x_all, y_all))                                model_based_rl.py:810-811
    validation-init pickles        pickle of the list / array of reset states                 model_based_rl.py:444-487
    progress.csv                   rllab logger tabular output, one column per record_tabular  model_based_rl.py:590-733, 1037-1039, 1317-1320
    checkpoint names               policy-and-models-%d.ckpt, policy.ckpt, <scope>_<i>.ckpt    model_based_rl.py:728, 1128, 929

TensorFlow checkpoints (the .ckpt bundles themselves) and joblib pickles of rllab policy objects need TensorFlow / rllab
to read or write; only their NAMES are provided.  Dynamics / policy parameters cross the boundary as flat float32 vectors
(`Engine.get_dynamics / get_policy`) and are stored here as .npz with the TF variable names of training.py:183-194 as keys."""
import csv
import os
import pickle
from collections import OrderedDict

import numpy as np

ROLLOUTS_FILE = 'new_rollouts_%d.pkl'                      # model_based_rl.py:810
POLICY_AND_MODELS_CKPT = 'policy-and-models-%d.ckpt'       # :728
POLICY_CKPT = 'policy.ckpt'                                # :1128
MODEL_CKPT = '%s_%d.ckpt'                                  # :929  (scope, model index)

# columns the outer loop records per iteration, in the order of the record_tabular calls
PROGRESS_COLUMNS_OUTER = ['collect_data_time', 'model_opt_time', 'policy_opt_time', 'MaxPolicyWeightDiff', 'MinPolicyWeightDiff',
                          'AvgPolicyWeightDiff', 'save_and_log_time', 'Time', 'ItrTime']


def progress_columns(model_scopes=('training_dynamics',), modes=('real', 'trpo_mean', 'estimated')):
    """All columns of one progress.csv row: outer loop (:590-733) + optimize_models (:1037-1039) + optimize_policy (:1317-1320)."""
    cols = ['collect_data_time', '# model updates'] + ['%s_min_sum_validation_loss' % s for s in model_scopes] + ['model_opt_time']
    cols += ['%s_policy_mean_min_validation_cost' % k for k in modes] + ['real_current_validation_cost', '# policy updates']
    cols += PROGRESS_COLUMNS_OUTER[2:]
    return cols


def save_rollouts(log_dir, count, x_all, y_all):
    """:809-811 -- x_all [n][ns+na] = (o_t, a_t), y_all [n][ns] = o_{t+1}."""
    path = os.path.join(log_dir, ROLLOUTS_FILE % count)
    with open(path, 'wb') as f:
        pickle.dump((np.asarray(x_all), np.asarray(y_all)), f)
    return path


def load_rollouts(path):
    with open(path, 'rb') as f:
        x_all, y_all = pickle.load(f)
    x_all, y_all = np.asarray(x_all), np.asarray(y_all)
    assert x_all.ndim == 2 and y_all.ndim == 2 and x_all.shape[0] == y_all.shape[0]
    return x_all, y_all


def save_validation_init(path, states):
    """:456-457 / :481-482 -- the reference pickles a python list of reset states (vip == vrip case) or an ndarray."""
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    with open(path, 'wb') as f:
        pickle.dump(states, f)


def load_validation_init(path):
    """-> float32 [n][ns] array whatever container the reference pickled (:447-452)."""
    with open(path, 'rb') as f:
        v = pickle.load(f)
    return np.asarray(v, dtype=np.float32)


class TabularLog(object):
    """rllab logger's record_tabular / dump_tabular to progress.csv: header = keys of the first dumped row (insertion order),
    every later row must carry the same keys."""

    def __init__(self, path):
        self.path, self.row, self.header = path, OrderedDict(), None

    def record_tabular(self, key, val):
        self.row[str(key)] = val

    def dump_tabular(self):
        new = self.header is None
        if new:
            self.header = list(self.row.keys())
        assert list(self.row.keys()) == self.header, "progress.csv columns changed between iterations"
        with open(self.path, 'w' if new else 'a', newline='') as f:
            w = csv.DictWriter(f, fieldnames=self.header)
            if new:
                w.writeheader()
            w.writerow(self.row)
        self.row = OrderedDict()


def read_progress(path):
    """-> dict column -> float array (non-numeric cells become nan)."""
    with open(path, newline='') as f:
        rows = list(csv.DictReader(f))
    out = OrderedDict()
    for k in (rows[0].keys() if rows else []):
        col = []
        for r in rows:
            try:
                col.append(float(r[k]))
            except (TypeError, ValueError):
                col.append(np.nan)
        out[k] = np.array(col)
    return out


def dynamics_variable_names(n_layers, model):
    """TF variable names of one model's MLP (training.py:183-194): model%d/layer%d/weights, .../biases."""
    names = []
    for l in range(n_layers):
        names += ['model%d/layer%d/weights' % (model, l), 'model%d/layer%d/biases' % (model, l)]
    return names


def save_dynamics_npz(path, engine, input_rms=None, diff_rms=None, scope='training_dynamics'):
    """All K models AND the running normalisers under the reference's TF variable names -- the payload of the checkpoint the
    reference writes (model_based_rl.py:728; variables training.py:183-194 and running_mean_std.py:5-20):
        <scope>/model<k>/layer<l>/weights | biases,   input_rms/runningsum | runningsumsq | count,   diff_rms/...
    input_rms / diff_rms: dynamics_training.RunningMeanStd objects (their sums are what TF checkpoints hold); without them
    the file carries weights only and load_dynamics_npz needs the normalisers supplied."""
    flat = engine.get_dynamics().detach().cpu().numpy()
    dims = [engine.ns + engine.na - engine.n_drop] + list(engine.dyn_hidden) + [engine.ns]
    arrs, L = {}, len(dims) - 1
    for k in range(engine.K):
        o = 0
        for l in range(L):
            n = dims[l] * dims[l + 1]
            arrs['%s/model%d/layer%d/weights' % (scope, k, l)] = flat[k, o:o + n].reshape(dims[l], dims[l + 1]); o += n
            arrs['%s/model%d/layer%d/biases' % (scope, k, l)] = flat[k, o:o + dims[l + 1]]; o += dims[l + 1]
    for name, rms in (('input_rms', input_rms), ('diff_rms', diff_rms)):
        if rms is not None:
            arrs[name + '/runningsum'] = rms._sum.detach().cpu().numpy()
            arrs[name + '/runningsumsq'] = rms._sumsq.detach().cpu().numpy()
            arrs[name + '/count'] = np.array(rms._count, dtype=np.float64)
    np.savez(path, **arrs)


def load_dynamics_npz(path, engine, input_rms=None, diff_rms=None, scope='training_dynamics'):
    """Inverse of save_dynamics_npz; works on a FRESH engine (one set_dynamics call with all K models and the normalisers).
    If the file holds the running sums they are restored into input_rms / diff_rms (when given) and mean/std (0.1 floor,
    running_mean_std.py:22-27) are derived from them; otherwise the given objects supply the normalisers."""
    import torch
    z = np.load(path)
    dims = [engine.ns + engine.na - engine.n_drop] + list(engine.dyn_hidden) + [engine.ns]
    L = len(dims) - 1
    pref = scope + '/' if ('%s/model0/layer0/weights' % scope) in z.files else ''      # files written before the scope prefix
    rows = []
    for k in range(engine.K):
        parts = []
        for l in range(L):
            parts += [z['%smodel%d/layer%d/weights' % (pref, k, l)].reshape(-1), z['%smodel%d/layer%d/biases' % (pref, k, l)].reshape(-1)]
        rows.append(np.concatenate(parts))
    stats = {}
    for name, rms in (('input_rms', input_rms), ('diff_rms', diff_rms)):
        if name + '/runningsum' in z.files:
            rsum, rsq, cnt = z[name + '/runningsum'], z[name + '/runningsumsq'], float(z[name + '/count'])
            if rms is not None:
                rms._sum.copy_(torch.as_tensor(rsum)); rms._sumsq.copy_(torch.as_tensor(rsq)); rms._count = cnt
            mean = rsum / cnt
            stats[name] = (mean, np.sqrt(np.maximum(rsq / cnt - mean * mean, 1e-2)))
        elif rms is not None:
            stats[name] = (rms.mean.cpu().numpy(), rms.std.cpu().numpy())
        else:
            raise KeyError("%s holds no %s statistics and none were supplied" % (path, name))
    ns = engine.ns
    engine.set_dynamics(torch.as_tensor(np.stack(rows).astype(np.float32)), stats['input_rms'][0], stats['input_rms'][1],
                        stats['diff_rms'][0][:ns], stats['diff_rms'][1][:ns])
