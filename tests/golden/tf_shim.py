"""Eager stand-in for the TensorFlow-1.4 API subset the reference's graph builders use.

GENERATOR-SIDE TOOLING ONLY (imported by tests/golden/make_golden.py in the build container; never by the
product, the tests or the bench).  TensorFlow 1.4 cannot be installed here, so the reference's OWN graph code --
training.py:96-117 (policy_model), :125-269 (prepare_input / build_ff_neural_net / dynamics_model), :271-282
(get_regularizer_loss), model_based_rl.py:23-104 (build_dynamics_graph), :106-151 (build_policy_graph), :154-206
(optimizers), utils.py:262-276 (minimize_and_clip), running_mean_std.py, envs/*.cost_tf / is_done_tf -- is executed
UNMODIFIED against this module registered as `tensorflow`: every tf.* call computes immediately on torch tensors
(float64 by default so the fixtures carry no float32 noise; autograd supplies `compute_gradients`).

Two kinds of value exist:
  * eager tensors (`ET`, a torch.Tensor subclass carrying the few tf.Tensor attributes the reference touches);
  * `Lazy` nodes: variables, placeholders and expressions made ONLY of those (e.g. RunningMeanStd.mean, built in the
    constructor from the running sums).  A Lazy is evaluated when it meets an eager tensor in an op or when a
    Session.run fetches it, so assign/assign_add/update() behave as in graph mode.

What is this module's own arithmetic (not the reference's, restated from the TF 1.4 documentation):
tf.train.AdamOptimizer / GradientDescentOptimizer update rules, tf.clip_by_norm, tf.nn.l2_loss,
tf.contrib.layers.xavier_initializer.  Everything else is a 1:1 map onto a torch op of the same meaning.
"""
import sys
import types
import builtins
import math
import contextlib
from collections import OrderedDict

import numpy as np
import torch

DT = torch.float64


class ET(torch.Tensor):
    name = 'eager:0'

    def get_shape(self):
        return _Shape(self.shape)

    def eval(self, feed_dict=None):
        return self.detach().numpy()


class _Shape(tuple):
    def as_list(self):
        return list(self)


def _wrap(t):
    return t.as_subclass(ET) if isinstance(t, torch.Tensor) and not isinstance(t, ET) else t


# ------------------------------------------------------------------ lazy nodes
class Lazy(object):
    def __init__(self, fn, args=(), name='lazy:0'):
        self._fn, self._args, self.name = fn, args, name

    def _eval(self, feed):
        return self._fn(*[_ev(a, feed) for a in self._args])

    # graph-mode API
    def eval(self, feed_dict=None):
        return _np(self._eval(feed_dict or {}))

    @property
    def shape(self):
        return _Shape(self._eval({}).shape)

    def get_shape(self):
        return self.shape

    def __hash__(self):
        return id(self)

    def __getitem__(self, idx):
        return _op(lambda x: x[idx], self)

    def __neg__(self):
        return _op(torch.neg, self)

    def assign(self, v):
        return assign(self, v)

    def assign_add(self, v):
        return assign_add(self, v)


def _bin(name, fn, rfn=None):
    setattr(Lazy, '__%s__' % name, lambda a, b: _op(fn, a, b))
    setattr(Lazy, '__r%s__' % name, lambda a, b: _op(rfn or (lambda x, y: fn(y, x)), a, b))


_bin('add', lambda a, b: a + b)
_bin('sub', lambda a, b: a - b)
_bin('mul', lambda a, b: a * b)
_bin('truediv', lambda a, b: a / b)
_bin('pow', lambda a, b: a ** b)
for _n, _f in (('ge', lambda a, b: a >= b), ('le', lambda a, b: a <= b), ('gt', lambda a, b: a > b), ('lt', lambda a, b: a < b)):
    setattr(Lazy, '__%s__' % _n, (lambda f: lambda a, b: _op(f, a, b))(_f))


class _Var(Lazy):
    def __init__(self, value, name, trainable=True):
        self.tensor = torch.as_tensor(np.asarray(value), dtype=DT).clone().requires_grad_(builtins.bool(trainable))
        self.name, self.trainable = name + ':0', trainable
        self.initial = self.tensor.detach().clone()
        self.previous = None

    def _eval(self, feed):
        return _wrap(self.tensor)

    def set(self, value):
        """Rebinds (never mutates) the leaf: graphs already built keep the value they were evaluated with, exactly like a
        fetched TF tensor; later evaluations see the new value."""
        new = torch.as_tensor(np.asarray(_np(value)), dtype=DT).reshape(self.tensor.shape)
        self.previous = self.tensor
        self.tensor = new.detach().clone().requires_grad_(builtins.bool(self.trainable))

    def numpy(self):
        return self.tensor.detach().numpy().copy()


class Placeholder(Lazy):
    def __init__(self, dtype, shape=None, name='ph'):
        self.dtype, self._shape, self.name = dtype, shape, name + ':0'

    @property
    def shape(self):
        return _Shape(self._shape)

    def _eval(self, feed):
        if feed is None or self not in feed:
            raise ValueError('placeholder %s was not fed' % self.name)
        return _t(feed[self])


def _t(x):
    if isinstance(x, torch.Tensor):
        return _wrap(x)
    a = np.asarray(x)
    if a.dtype == builtins.bool:
        return _wrap(torch.as_tensor(a))
    if a.dtype.kind in 'iu':
        return _wrap(torch.as_tensor(a, dtype=torch.int64))
    return _wrap(torch.as_tensor(a, dtype=DT))


def _ev(a, feed):
    if isinstance(a, Lazy):
        return a._eval(feed)
    if isinstance(a, (list, tuple)):
        return [_ev(x, feed) for x in a]
    if isinstance(a, np.ndarray):
        return _t(a)
    return a


def _has(args, cls):
    for a in args:
        if isinstance(a, cls) or (isinstance(a, (list, tuple)) and _has(a, cls)):
            return True
    return False


def _op(fn, *args):
    """Stay lazy iff some argument is Lazy and none is an eager tensor; otherwise compute now."""
    if _has(args, Lazy) and not _has(args, torch.Tensor):
        return Lazy(fn, args)
    return _wrap(fn(*[_ev(a, {}) for a in args]))


def _np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().numpy().copy()
    if isinstance(v, (list, tuple)):
        return [_np(x) for x in v]
    return v


def _tt(x):
    """operand of a torch binary function -> tensor"""
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), dtype=DT if np.asarray(x).dtype.kind == 'f' or isinstance(x, float) else None)


# ------------------------------------------------------------------ dtypes, graph keys, scopes, variables
float32, float64, int32, int64 = 'float32', 'float64', 'int32', 'int64'
bool = 'bool'  # noqa: A001  (tf.bool)


class GraphKeys(object):
    GLOBAL_VARIABLES, TRAINABLE_VARIABLES, SUMMARIES = 'variables', 'trainable_variables', 'summaries'


_VARS = OrderedDict()
_COLLECTIONS = {}
_SCOPE = []
_SESSION = [None]
_RNG = [np.random.RandomState(0)]
_UNIQ = [0]


def reset_default_graph():
    _VARS.clear(); _COLLECTIONS.clear(); del _SCOPE[:]; _UNIQ[0] = 0


def set_random_seed(i):
    _RNG[0] = np.random.RandomState(i)
    torch.manual_seed(i)


class _VarScope(object):
    def __init__(self, name):
        self.name = name

    def reuse_variables(self):
        pass


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    _SCOPE.append(name)
    try:
        yield _VarScope('/'.join(_SCOPE))
    finally:
        _SCOPE.pop()


@contextlib.contextmanager
def name_scope(name=None):
    yield name


def _full(name):
    return '/'.join(_SCOPE + [name])


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    """AUTO_REUSE semantics: an existing variable of that scoped name is returned, else it is created."""
    full = _full(name)
    if full in _VARS:
        return _VARS[full]
    if shape is None:
        raise ValueError('Variable %s does not exist' % full)
    shp = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
    v = _Var(initializer(shp), full, trainable)
    _VARS[full] = v
    return v


def Variable(value, trainable=True, name=None, dtype=None):
    _UNIQ[0] += 1
    full = _full(name or 'Variable_%d' % _UNIQ[0])
    v = _Var(value, full, trainable)
    _VARS[full] = v
    return v


def constant_initializer(value=0.0):
    return lambda shape: np.full(shape, value, dtype=np.float64)


def xavier_initializer(uniform=True, seed=None):
    """tf.contrib.layers.xavier_initializer: U(-l, l), l = sqrt(6/(fan_in+fan_out)); rank-1 shapes: fan_in = fan_out = n."""
    def init(shape):
        fan_in, fan_out = (shape[0], shape[0]) if len(shape) == 1 else (shape[-2], shape[-1])
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return _RNG[0].uniform(-lim, lim, size=shape)
    return init


def get_collection(key, scope=None):
    if key == GraphKeys.GLOBAL_VARIABLES:
        items = list(_VARS.values())
    elif key == GraphKeys.TRAINABLE_VARIABLES:
        items = [v for v in _VARS.values() if v.trainable]
    else:
        items = list(_COLLECTIONS.get(key, []))
    if scope is not None:
        items = [v for v in items if getattr(v, 'name', '').startswith(scope)]
    return items


def add_to_collection(key, value):
    _COLLECTIONS.setdefault(key, []).append(value)


def variables_initializer(var_list, name=None):
    def run():
        for v in var_list:
            v.set(v.initial)
    return Lazy(lambda: run() or 0.0)


def global_variables_initializer():
    return Lazy(lambda: 0.0)


def placeholder(dtype, shape=None, name='ph'):
    return Placeholder(dtype, shape, name)


def constant(value, dtype=None, name=None):
    return _t(np.asarray(value, dtype=np.float64))


# ------------------------------------------------------------------ assignment (side effects, deferred like graph ops)
def _assign(var, value, mode):
    def run(val):
        val = torch.as_tensor(np.asarray(_np(val)), dtype=DT)
        cur = var.tensor.detach()
        var.set(val.expand_as(cur) if mode == 'set' else cur + val if mode == 'add' else cur - val)
        return _wrap(var.tensor.detach().clone())
    return Lazy(run, (value,))


def assign(var, value):
    return _assign(var, value, 'set')


def assign_add(var, value):
    return _assign(var, value, 'add')


def assign_sub(var, value):
    return _assign(var, value, 'sub')


# ------------------------------------------------------------------ sessions
class Session(object):
    graph = None

    def __init__(self, *a, **k):
        _SESSION[0] = self

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass

    def run(self, fetches, feed_dict=None):
        feed = feed_dict or {}
        if isinstance(fetches, (list, tuple)):
            return [self.run(f, feed) for f in fetches]
        if isinstance(fetches, Lazy):
            return _np(fetches._eval(feed))
        return _np(fetches)

    def close(self):
        pass


InteractiveSession = Session


def get_default_session():
    return _SESSION[0]


def ConfigProto(*a, **k):
    return None


def GPUOptions(*a, **k):
    return None


# ------------------------------------------------------------------ ops
def _red(fn):
    def f(x, axis=None, name=None, keep_dims=False):
        def g(x):
            if isinstance(x, (list, tuple)):
                x = torch.stack([_tt(v).to(DT) for v in x])
            x = _tt(x)
            if axis is None:
                return fn(x)
            ax = tuple(axis) if isinstance(axis, (list, tuple)) else axis
            return fn(x, ax)
        return _op(g, x)
    return f


reduce_mean = _red(lambda x, ax=None: x.mean() if ax is None else x.mean(dim=ax))
reduce_sum = _red(lambda x, ax=None: x.sum() if ax is None else x.sum(dim=ax))
reduce_max = _red(lambda x, ax=None: x.max() if ax is None else x.amax(dim=ax))
reduce_min = _red(lambda x, ax=None: x.min() if ax is None else x.amin(dim=ax))
reduce_all = _red(lambda x, ax=None: x.all() if ax is None else x.all(dim=ax))


def _un(fn):
    return lambda x, name=None: _op(lambda v: fn(_tt(v)), x)


square, sqrt, exp, abs, tanh, identity = _un(torch.square), _un(torch.sqrt), _un(torch.exp), _un(torch.abs), _un(torch.tanh), _un(lambda v: v)  # noqa: A001
is_finite, logical_not = _un(torch.isfinite), _un(torch.logical_not)
to_float = _un(lambda v: v.to(DT))


def _bi(fn):
    return lambda a, b, name=None: _op(lambda x, y: fn(*torch.broadcast_tensors(_tt(x).to(DT) if _tt(x).dtype != torch.bool else _tt(x),
                                                                                 _tt(y).to(DT) if _tt(y).dtype != torch.bool else _tt(y))), a, b)


maximum, minimum, add = _bi(torch.maximum), _bi(torch.minimum), _bi(torch.add)
logical_and = _bi(torch.logical_and)
equal = _bi(torch.eq)


def matmul(a, b, name=None):
    return _op(lambda x, y: torch.matmul(_tt(x), _tt(y)), a, b)


def concat(values, axis=0, name=None):
    return _op(lambda vs: torch.cat([_tt(v) for v in vs], dim=axis), values)


def stack(values, axis=0, name=None):
    return _op(lambda vs: torch.stack([_tt(v) for v in vs], dim=axis), values)


def reshape(x, shape, name=None):
    return _op(lambda v: _tt(v).reshape(tuple(shape)), x)


def clip_by_value(x, lo, hi, name=None):
    return _op(lambda v, l, h: torch.minimum(torch.maximum(_tt(v), _tt(l).to(DT)), _tt(h).to(DT)), x, lo, hi)


def clip_by_norm(t, clip_norm, name=None):
    """t * clip_norm / max(||t||_2, clip_norm)   (TF 1.4 documentation)."""
    return _op(lambda v, c: v * c / torch.maximum(torch.sqrt(torch.sum(v * v)), torch.as_tensor(float(c), dtype=DT)), t, clip_norm)


def norm(x, name=None):
    return _t(0.0)


def cast(x, dtype, name=None):
    def g(v):
        v = _tt(v)
        if dtype in ('float32', 'float64'):
            return v.to(DT)
        if dtype in ('int32', 'int64'):
            return v.to(torch.int64)
        return v.to(torch.bool)
    return _op(g, x)


def shape(x):
    return list(_ev(x, {}).shape)


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return _t(_RNG[0].normal(mean, stddev, size=tuple(int(s) for s in shape)))


class _NN(object):
    relu = staticmethod(_un(torch.relu))
    tanh = staticmethod(_un(torch.tanh))
    sigmoid = staticmethod(_un(torch.sigmoid))
    l2_loss = staticmethod(_un(lambda v: torch.sum(v * v) / 2))


nn = _NN()


class _Summary(object):
    @staticmethod
    def scalar(*a, **k): return None
    @staticmethod
    def histogram(*a, **k): return None
    @staticmethod
    def merge(*a, **k): return None
    @staticmethod
    def merge_all(*a, **k): return None

    class FileWriter(object):
        def __init__(self, *a, **k): pass
        def add_summary(self, *a, **k): pass
        def flush(self): pass
        def close(self): pass


summary = _Summary()


# ------------------------------------------------------------------ optimizers (TF 1.4 documented update rules)
class _Optimizer(object):
    def compute_gradients(self, loss, var_list=None):
        var_list = list(var_list)
        loss = _ev(loss, {})
        if not (isinstance(loss, torch.Tensor) and loss.requires_grad):
            return [(None, v) for v in var_list]
        # a loss built before the last assignment refers to the value the variable had then (TF: the gradient is taken
        # w.r.t. the variable, whatever was assigned since)
        cur = [v.tensor for v in var_list]
        old = [v.previous if (v.previous is not None and v.previous.requires_grad) else v.tensor for v in var_list]
        grads = torch.autograd.grad(loss, cur + old, allow_unused=True, retain_graph=True)
        n = len(var_list)
        grads = [grads[i] if grads[i] is not None else grads[n + i] for i in range(n)]
        return [(None if g is None else _wrap(g), v) for g, v in zip(grads, var_list)]

    def minimize(self, loss, var_list=None):
        return self.apply_gradients(self.compute_gradients(loss, var_list))


class GradientDescentOptimizer(_Optimizer):
    def __init__(self, learning_rate, name='GradientDescent'):
        self.lr = learning_rate

    def apply_gradients(self, grads_and_vars):
        lr = float(_np(_ev(self.lr, {})))
        for g, v in grads_and_vars:
            if g is not None:
                v.set(v.tensor.detach() - lr * g.detach())
        return Lazy(lambda: 0.0)


class AdamOptimizer(_Optimizer):
    """lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; var -= lr_t m/(sqrt(v)+eps).
    Slot variables and the beta powers live in the variable store under the enclosing variable_scope (as TF's do), so a
    second optimizer built in the same scope continues the same state: one eager `apply_gradients` per rebuilt graph
    equals one sess.run(opt_op) of the persistent graph."""

    def __init__(self, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, name='Adam'):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.scope = '/'.join(_SCOPE)

    def _slot(self, key, like):
        full = '%s/%s' % (self.scope, key)
        if full not in _VARS:
            _VARS[full] = _Var(np.zeros(like), full, trainable=False)
        return _VARS[full]

    def apply_gradients(self, grads_and_vars):
        t_var = self._slot('adam_step', ())
        t_var.set(t_var.tensor + 1.0)
        t = float(t_var.tensor)
        lr = float(_np(_ev(self.lr, {})))
        lr_t = lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        for g, v in grads_and_vars:
            if g is None:
                continue
            g = g.detach().as_subclass(torch.Tensor)
            m = self._slot(v.name[:-2] + '/Adam', tuple(v.tensor.shape))
            s = self._slot(v.name[:-2] + '/Adam_1', tuple(v.tensor.shape))
            m.set(self.b1 * m.tensor + (1 - self.b1) * g)
            s.set(self.b2 * s.tensor + (1 - self.b2) * g * g)
            v.set(v.tensor.detach() - lr_t * m.tensor / (torch.sqrt(s.tensor) + self.eps))
        return Lazy(lambda: 0.0)


class _Saver(object):
    def __init__(self, *a, **k): pass
    def save(self, *a, **k): pass
    def restore(self, *a, **k): pass


class _Train(object):
    AdamOptimizer = AdamOptimizer
    GradientDescentOptimizer = GradientDescentOptimizer
    Saver = _Saver

    @staticmethod
    def latest_checkpoint(*a, **k): return None


train = _Train()


# ------------------------------------------------------------------ registration as `tensorflow`
def install():
    """Register this module as `tensorflow` (+ the sub-modules the reference imports by dotted name)."""
    me = sys.modules[__name__]
    me.__path__ = []
    sys.modules['tensorflow'] = me
    contrib = types.ModuleType('tensorflow.contrib')
    layers = types.ModuleType('tensorflow.contrib.layers')
    layers.xavier_initializer = xavier_initializer
    contrib.layers = layers
    me.contrib = contrib
    sys.modules['tensorflow.contrib'] = contrib
    sys.modules['tensorflow.contrib.layers'] = layers
    return me
