#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE'S OWN Python (read-only at /root/reference).

Runs only in the build container (the reference does not exist on the GPU box); the .npz
files it writes next to this script are the committed fixtures.  Nothing from the reference is
copied: its modules are imported in place (sys.dont_write_bytecode) under an import-stub finder
that fabricates the third-party packages it needs at import time but that are absent here
(tensorflow 1.4, rllab, gym, mujoco_py ...).  Only NumPy-level reference code is executed:

  envs/*: cost_np_vec, is_done                      -> rewards_*.npz, ant_done.npz
  env_helpers.NeuralNetEnv / VecSimpleEnv           -> vecenv_<sam_mode>.npz
  samplers.vectorized_sampler.VectorizedSampler     -> sampler_*.npz
  samplers.base.BaseSampler.process_samples         -> process_*.npz
  utils.stop_critereon, model_based_rl.is_done / update_stats -> stoplogic.npz

The TF session the reference calls (env_helpers.py:610-616) is replaced by a fake whose
run() evaluates oracle.metrpo_oracle.dynamics_forward -- that MLP arithmetic is therefore NOT
pinned by these fixtures (see the oracle header); everything downstream of it is.
The tiny rllab helpers the samplers call (tensor_utils, special.discount_cumsum,
util.center_advantages, Box.flatten_n, ProgBarCounter) are restated here from rllab master.

Usage:  python tests/golden/make_golden.py        (from the repo root)
"""
import sys
sys.dont_write_bytecode = True
import os, types, importlib.abc, importlib.machinery
from types import SimpleNamespace as NS
import numpy as np
import scipy.signal

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

# ---------------------------------------------------------------- 1. import stubs
PREFIXES = ('tensorflow', 'rllab', 'sandbox', 'gym', 'private_examples', 'cached_property',
            'colored_traceback', 'mujoco_py', 'joblib', 'matplotlib')


class Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name[0].isupper():
            obj = type(name, (object,), {'__init__': lambda s, *a, **k: None,
                                         'quick_init': staticmethod(lambda *a, **k: None)})
        else:
            obj = sys.modules.setdefault('%s.%s' % (self.__name__, name), Stub('%s.%s' % (self.__name__, name)))
        setattr(self, name, obj)
        return obj

    def __call__(self, *a, **k):
        return self


class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)

    def create_module(self, spec):
        return Stub(spec.name)

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, Finder())
import rllab.misc.overrides, rllab.misc.autoargs, cached_property  # noqa: E402
rllab.misc.overrides.overrides = lambda f: f
rllab.misc.autoargs.arg = lambda *a, **k: (lambda f: f)
cached_property.cached_property = property
if not hasattr(np, 'cast'):                      # env_helpers.py:589 uses np.cast (removed in NumPy 2)
    np.cast = {'bool': lambda x: np.asarray(x, dtype=bool)}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition('.')
    __import__(parent)
    setattr(sys.modules[parent], leaf, m)
    return m


# ---------------------------------------------------------------- 2. rllab helper restatements
def stack_tensor_list(tl):
    return np.array(tl)


def stack_tensor_dict_list(tdl):
    if not tdl:
        return {}
    keys = list(tdl[0].keys())
    ret = dict()
    for k in keys:
        ex = tdl[0][k]
        ret[k] = stack_tensor_dict_list([x[k] for x in tdl]) if isinstance(ex, dict) else stack_tensor_list([x[k] for x in tdl])
    return ret


def split_tensor_dict_list(td):
    keys = list(td.keys())
    ret = None
    for k in keys:
        vals = td[k]
        if isinstance(vals, dict):
            vals = split_tensor_dict_list(vals)
        if ret is None:
            ret = [{k: v} for v in vals]
        else:
            for v, cur in zip(vals, ret):
                cur[k] = v
    return ret


def concat_tensor_list(tl):
    return np.concatenate(tl, axis=0)


def concat_tensor_dict_list(tdl):
    if not tdl or not tdl[0]:
        return {}
    keys = list(tdl[0].keys())
    ret = dict()
    for k in keys:
        ex = tdl[0][k]
        ret[k] = concat_tensor_dict_list([x[k] for x in tdl]) if isinstance(ex, dict) else concat_tensor_list([x[k] for x in tdl])
    return ret


_mod('rllab.misc.tensor_utils', stack_tensor_list=stack_tensor_list, stack_tensor_dict_list=stack_tensor_dict_list,
     split_tensor_dict_list=split_tensor_dict_list, concat_tensor_list=concat_tensor_list,
     concat_tensor_dict_list=concat_tensor_dict_list)
_mod('rllab.misc.special',
     discount_cumsum=lambda x, d: scipy.signal.lfilter([1], [1, float(-d)], x[::-1], axis=0)[::-1],
     explained_variance_1d=lambda yp, y: 0.0)
_mod('rllab.algos.util', center_advantages=lambda a: (a - np.mean(a)) / (a.std() + 1e-8),
     shift_advantages_to_positive=lambda a: (a - np.min(a)) + 1e-8)


class ProgBarCounter(object):
    def __init__(self, *a, **k): pass
    def inc(self, n): pass
    def stop(self): pass


_mod('rllab.sampler.stateful_pool', ProgBarCounter=ProgBarCounter, singleton_pool=None)

# ---------------------------------------------------------------- 3. import the reference
import env_helpers                                               # noqa: E402
from samplers.vectorized_sampler import VectorizedSampler        # noqa: E402
import utils as ref_utils                                        # noqa: E402
import model_based_rl as ref_mbrl                                # noqa: E402
from envs.com_swimmer_env import SwimmerEnv                      # noqa: E402
from envs.com_half_cheetah_env import HalfCheetahEnv             # noqa: E402
from envs.com_ant_env import AntEnv                              # noqa: E402
from envs.com_simple_humanoid_env import SimpleHumanoidEnv       # noqa: E402
from envs.com_hopper_env import HopperEnv                        # noqa: E402
from envs.com_snake_env import SnakeEnv                          # noqa: E402
import tensorflow as tf                                          # noqa: E402  (stub)

from oracle import metrpo_oracle as O                            # noqa: E402

REF_ENVS = {
    'swimmer': (SwimmerEnv, NS(ctrl_cost_coeff=1e-2)),          # com_swimmer_env.py:43
    'half_cheetah': (HalfCheetahEnv, NS(ctrl_cost_coeff=1e-1)), # com_half_cheetah_env.py:21
    'ant': (AntEnv, NS()),
    'humanoid': (SimpleHumanoidEnv, NS(ctrl_cost_coeff=1e-3)),  # com_simple_humanoid_env.py:27
    'hopper': (HopperEnv, NS(ctrl_cost_coeff=0.01)),            # com_hopper_env.py:30
    'snake': (SnakeEnv, NS(ctrl_cost_coeff=1e-2)),              # com_snake_env.py:21
}


def ref_cost(env):
    cls, self_ = REF_ENVS[env]
    return lambda x, u, xn: cls.cost_np_vec(self_, x, u, xn)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote %-28s %6.1f KB' % (name + '.npz', os.path.getsize(path) / 1024.0))


# ---------------------------------------------------------------- 4. reward / done KATs
def gen_rewards():
    rng = np.random.RandomState(100)
    for env, (ns, na, _) in O.ENV_SPECS.items():
        n = 64
        x = rng.randn(n, ns)
        xn = rng.randn(n, ns) * 2.0
        u = rng.uniform(-1, 1, size=(n, na))
        if env == 'half_cheetah':
            xn[:8, 9] = rng.uniform(-40, 40, size=8)            # exercise the +-10 clip
        if env == 'hopper':
            xn[:8, 0] = rng.uniform(0.0, 0.6, size=8)           # height penalty
            xn[8:16, 1] = rng.uniform(-0.5, 0.5, size=8)        # angle penalty
            xn[16:20, 4] = rng.uniform(90, 130, size=4) * np.array([1, -1, 1, -1])  # |x|>100 penalty
        save('rewards_' + env, x=x, u=u, x_next=xn, cost=ref_cost(env)(x, u, xn))
    # Ant termination incl. boundaries and non-finite rows (com_ant_env.py:88-101)
    xn = rng.randn(40, 29)
    xn[:, 2] = rng.uniform(0.0, 1.2, size=40)
    xn[0, 2], xn[1, 2], xn[2, 2], xn[3, 2] = 0.2, 1.0, 0.19999999, 1.0000001
    xn[4, 2] = 0.5; xn[4, 7] = np.nan
    xn[5, 2] = 0.5; xn[5, 20] = np.inf
    xn[6, 2] = 0.5; xn[6, 0] = -np.inf
    xn[7, 2] = np.nan
    done = AntEnv.is_done(NS(), xn, xn)
    save('ant_done', x_next=xn, done=done)


# ---------------------------------------------------------------- 5. fake session + recording RNG
class FakeSession(object):
    """Stands in for tf.get_default_session() at env_helpers.py:610: fetches = list of K model
    handles (ints), feed = {placeholder: (B, ns+na)} -> list of K (B, ns) arrays."""

    def __init__(self, dm):
        self.dm = dm
        self.log = []

    def run(self, fetches, feed_dict):
        (xin,) = feed_dict.values()
        s, a = xin[:, :self.dm.ns], xin[:, self.dm.ns:]
        outs = [O.dynamics_forward(self.dm, k, s, a) for k in fetches]
        self.log.append(np.array(outs))
        return outs


class Recorder(object):
    """Wraps np.random.randint / normal and records every draw in call order."""

    def __init__(self):
        self.calls = []
        self._ri, self._no = np.random.randint, np.random.normal

    def __enter__(self):
        def randint(*a, **k):
            v = self._ri(*a, **k); self.calls.append(('randint', np.array(v))); return v

        def normal(*a, **k):
            v = self._no(*a, **k); self.calls.append(('normal', np.array(v))); return v
        np.random.randint, np.random.normal = randint, normal
        return self

    def __exit__(self, *exc):
        np.random.randint, np.random.normal = self._ri, self._no


def pack_calls(calls):
    """-> (kinds uint8[n], offsets int64[n+1], flat float64) so a test can replay the exact stream."""
    kinds = np.array([0 if k == 'randint' else 1 for k, _ in calls], dtype=np.uint8)
    flats = [np.asarray(v, dtype=np.float64).reshape(-1) for _, v in calls]
    offs = np.concatenate([[0], np.cumsum([f.size for f in flats])]).astype(np.int64)
    return kinds, offs, (np.concatenate(flats) if flats else np.zeros(0))


class PoolEnv(object):
    """Stand-in for the real simulator: reset() hands out rows of a fixed pool in order."""

    def __init__(self, pool, na):
        self.pool, self.i = pool, 0
        ns = pool.shape[1]
        self.observation_space = NS(shape=(ns,), flatten_n=lambda xs: np.asarray(xs).reshape((len(xs), -1)))
        self.action_space = NS(shape=(na,), bounds=(-np.ones(na), np.ones(na)),
                               flatten_n=lambda xs: np.asarray(xs).reshape((len(xs), -1)))

    def reset(self):
        s = self.pool[self.i % len(self.pool)].copy()
        self.i += 1
        return s


def make_nne(env, dm, pool, sam_mode):
    inner = NS()
    if env == 'ant':
        inner.is_done = lambda x, xn: AntEnv.is_done(NS(), x, xn)
    nne = env_helpers.NeuralNetEnv(env=PoolEnv(pool, dm.na), inner_env=inner, cost_np=ref_cost(env),
                                   dynamics_in='ph', dynamics_outs=list(range(dm.K)), sam_mode=sam_mode)
    nne.spec = NS(observation_space=nne.env.observation_space, action_space=nne.env.action_space)
    return nne


def small_problem(env, K, seed, hidden=(8, 8)):
    return O.make_problem(env, K=K, dyn_hidden=hidden, pol_hidden=(8, 8), seed=seed, n_pool=64)


def dm_arrays(dm):
    d = {}
    for l, (W, b) in enumerate(zip(dm.Ws, dm.bs)):
        d['dynW%d' % l], d['dynb%d' % l] = W, b
    d.update(in_mean=dm.in_mean, in_std=dm.in_std, diff_mean=dm.diff_mean, diff_std=dm.diff_std,
             n_drop=np.array(dm.n_drop), dyn_act=np.array(dm.acts[0]))
    return d


# ---------------------------------------------------------------- 6. VecSimpleEnv traces
def gen_vecenv():
    for sam_mode in O.SAM_MODES:
        for env in (('swimmer', 'ant') if sam_mode in ('step_rand', 'eps_rand') else ('swimmer',)):
            K, B, H, T = 3, 6, 4, 10
            dm, theta, pdims, pool = small_problem(env, K, seed=7)
            if env == 'ant':   # make some envs leave the healthy band quickly
                pool[::3, 2] = 0.21
                dm.diff_mean[2] = -0.02
            sess = FakeSession(dm)
            tf.get_default_session = lambda: sess
            np.random.seed(1234)
            arng = np.random.RandomState(5)
            actions = arng.randn(T, B, dm.na) * 0.8          # some exceed +-1 -> clip path exercised
            with Recorder() as rec:
                nne = make_nne(env, dm, pool, sam_mode)
                ve = nne.vec_env_executor(n_envs=B, max_path_length=H)
                first = ve.reset().copy()
                tr = dict(states=[], rewards=[], dones=[], ts=[], cur_idx=[])
                for t in range(T):
                    s, r, d, _ = ve.step(actions[t])
                    tr['states'].append(np.array(s)); tr['rewards'].append(np.array(r))
                    tr['dones'].append(np.array(d)); tr['ts'].append(ve.ts.copy())
                    tr['cur_idx'].append(ve.cur_model_idx.copy())
            kinds, offs, flat = pack_calls(rec.calls)
            save('vecenv_%s_%s' % (env, sam_mode), env=np.array(env), sam_mode=np.array(sam_mode),
                 K=np.array(K), B=np.array(B), H=np.array(H), pool=pool, actions=actions, first_obs=first,
                 states=np.array(tr['states']), rewards=np.array(tr['rewards']), dones=np.array(tr['dones']),
                 ts=np.array(tr['ts']), cur_idx=np.array(tr['cur_idx']), next_all=np.array(sess.log),
                 rng_kinds=kinds, rng_offs=offs, rng_flat=flat, **dm_arrays(dm))


# ---------------------------------------------------------------- 7. sampler + process_samples
class RefPolicy(object):
    """What the sampler needs from rllab's GaussianMLPPolicy (reset, get_actions, recurrent,
    distribution.entropy); arithmetic = oracle.policy_get_actions (unpinned, [rllab])."""
    recurrent = False

    def __init__(self, theta, dims):
        self.theta, self.dims = theta, dims
        self.distribution = NS(entropy=lambda info: np.sum(info['log_std'] + np.log(np.sqrt(2 * np.pi * np.e)), axis=-1))

    def reset(self, dones=None):
        pass

    def get_actions(self, obs):
        eps = np.random.normal(size=(len(obs), self.dims[-1]))
        return O.policy_get_actions(self.theta, self.dims, np.asarray(obs), eps)


def paths_arrays(paths, prefix=''):
    d = {prefix + 'n_paths': np.array(len(paths)), prefix + 'lengths': np.array([len(p['rewards']) for p in paths])}
    for k in ('observations', 'actions', 'rewards'):
        d[prefix + k] = np.concatenate([p[k] for p in paths])
    d[prefix + 'mean'] = np.concatenate([p['agent_infos']['mean'] for p in paths])
    d[prefix + 'log_std'] = np.concatenate([p['agent_infos']['log_std'] for p in paths])
    return d


def gen_sampler():
    cases = [('swimmer', 'step_rand', 5, 6, 24, False, 1.0, 1.0),      # exact multiple: 1 round... (4 envs x 6)
             ('swimmer', 'step_rand', 5, 6, 50, False, 0.99, 0.95),    # overshoot to whole rounds (quirk 6)
             ('swimmer', 'eps_rand', 5, 6, 30, True, 1.0, 1.0),        # determ=True (model_based_rl.py:1221)
             ('ant', 'step_rand', 4, 8, 60, False, 0.99, 0.95),        # early termination, ragged paths
             ('half_cheetah', 'model_mean_std', 4, 5, 20, False, 0.99, 1.0)]
    for ci, (env, sam_mode, B, H, batch, determ, gamma, lam) in enumerate(cases):
        K = 3
        dm, theta, pdims, pool = small_problem(env, K, seed=20 + ci)
        if env == 'ant':
            pool[::2, 2] = 0.22
            dm.diff_mean[2] = -0.015
        sess = FakeSession(dm)
        tf.get_default_session = lambda: sess
        np.random.seed(4321 + ci)
        policy = RefPolicy(theta, pdims)
        # previous-iteration baseline coefficients (quirk 8): a fitted-looking vector, or None (first iteration)
        base = O.LinearFeatureBaselineOracle()
        if ci != 0:
            base._coeffs = np.random.RandomState(9 + ci).randn(2 * dm.ns + 4) * 0.05
        coeffs_before = None if base._coeffs is None else base._coeffs.copy()
        with Recorder() as rec:
            nne = make_nne(env, dm, pool, sam_mode)
            algo = NS(env=nne, policy=policy, baseline=base, batch_size=batch, max_path_length=H,
                      discount=gamma, gae_lambda=lam, center_adv=True, positive_adv=False)
            smp = VectorizedSampler(algo, n_envs=B)
            smp.start_worker()
            paths = smp.obtain_samples(0, determ=determ)
        kinds, offs, flat = pack_calls(rec.calls)
        arrs = paths_arrays(paths)
        samples = smp.process_samples(0, paths)
        save('sampler_%d_%s_%s' % (ci, env, sam_mode), env=np.array(env), sam_mode=np.array(sam_mode),
             K=np.array(K), B=np.array(B), H=np.array(H), batch_size=np.array(batch), determ=np.array(determ),
             gamma=np.array(gamma), lam=np.array(lam), pool=pool, theta=theta, pdims=np.array(pdims),
             rng_kinds=kinds, rng_offs=offs, rng_flat=flat,
             has_coeffs=np.array(coeffs_before is not None),
             coeffs_before=(coeffs_before if coeffs_before is not None else np.zeros(0)),
             coeffs_after=base._coeffs,
             s_observations=samples['observations'], s_actions=samples['actions'], s_rewards=samples['rewards'],
             s_returns=samples['returns'], s_advantages=samples['advantages'],
             s_mean=samples['agent_infos']['mean'], s_log_std=samples['agent_infos']['log_std'],
             **arrs, **dm_arrays(dm))


# ---------------------------------------------------------------- 8. early-stopping decision tables
def gen_stoplogic():
    rng = np.random.RandomState(77)
    stop = ref_utils.stop_critereon(threshold=0.10, offset=1e-5, percent_models_threshold=0.30)  # params-swimmer.json:63-67
    olds = rng.randn(40, 5)
    news = olds + rng.randn(40, 5) * 0.5
    news[0] = olds[0]                      # ties are not "worse"
    news[1, :2] = olds[1, :2] + 1.0; news[1, 2:] = olds[1, 2:] - 1.0   # exactly 40% worse > 30%
    news[2, :1] = olds[2, :1] + 1.0; news[2, 1:] = olds[2, 1:] - 1.0   # 20% worse
    vec = np.array([stop(o, n, mode='vector') for o, n in zip(olds, news)])
    sc_old, sc_new = rng.randn(40), rng.randn(40)
    scal = np.array([stop(o, n) for o, n in zip(sc_old, sc_new)])
    # is_done over modes, update_stats with whole in {True, False}
    logger = NS(info=lambda *a, **k: None)
    modes = ['real', 'trpo_mean', 'one_model', 'no_early', 'estimated']
    rows = []
    for mode in modes:
        for j in range(12):
            mins = {'real': float(rng.randn()), 'trpo_mean': float(rng.randn()), 'estimated': rng.randn(5)}
            cand = {'real': float(rng.randn()), 'trpo_mean': float(rng.randn()), 'estimated': mins['estimated'] + rng.randn(5)}
            pop = NS(mode=mode, stop_critereon=stop)
            done = ref_mbrl.is_done(pop, {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in mins.items()}, cand, logger)
            upd = {}
            for whole in (False, True):
                m2 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in mins.items()}
                ref_mbrl.update_stats(m2, cand, whole)
                upd[whole] = m2
            rows.append((modes.index(mode), mins['real'], mins['trpo_mean'], mins['estimated'], cand['real'],
                         cand['trpo_mean'], cand['estimated'], bool(done),
                         upd[False]['real'], upd[False]['trpo_mean'], upd[False]['estimated'],
                         upd[True]['real'], upd[True]['trpo_mean'], upd[True]['estimated']))
    cols = list(zip(*rows))
    save('stoplogic', olds=olds, news=news, vec=vec, sc_old=sc_old, sc_new=sc_new, scal=scal,
         modes=np.array(modes), mode_idx=np.array(cols[0]), min_real=np.array(cols[1]), min_tm=np.array(cols[2]),
         min_est=np.array(cols[3]), cand_real=np.array(cols[4]), cand_tm=np.array(cols[5]), cand_est=np.array(cols[6]),
         done=np.array(cols[7]), upd0_real=np.array(cols[8]), upd0_tm=np.array(cols[9]), upd0_est=np.array(cols[10]),
         upd1_real=np.array(cols[11]), upd1_tm=np.array(cols[12]), upd1_est=np.array(cols[13]))


# ---------------------------------------------------------------- 9. replay buffer / batch slicing
def gen_dynamics_data():
    rng = np.random.RandomState(303)
    dc = ref_utils.data_collection(max_size=50)
    np.random.seed(99)
    log = []
    ops = [('add', 20), ('next', 8), ('next', 8), ('add', 25), ('next', 30), ('sample', 12), ('add', 30), ('next', 45), ('sample', 7),
           ('next', 10)]
    adds = []
    for op, n in ops:
        if op == 'add':
            x, y = rng.randn(n, 5), rng.randn(n, 3)
            adds.append((x, y))
            dc.add_data(x, y)
            log.append((0, n, dc.n_data, dc.cur_idx, dc.x.copy(), dc.y.copy()))
        elif op == 'next':
            xb, yb = dc.get_next_batch(n)
            log.append((1, n, dc.n_data, dc.cur_idx, np.array(xb), np.array(yb)))
        else:
            xb, yb = dc.sample(n)
            log.append((2, n, dc.n_data, dc.cur_idx, np.array(xb), np.array(yb)))
    arrs = dict(ops=np.array([l[0] for l in log]), ns=np.array([l[1] for l in log]), n_data=np.array([l[2] for l in log]),
                cur_idx=np.array([l[3] for l in log]), seed=np.array(99))
    for i, l in enumerate(log):
        arrs['x%d' % i], arrs['y%d' % i] = l[4], l[5]
    for i, (x, y) in enumerate(adds):
        arrs['addx%d' % i], arrs['addy%d' % i] = x, y
    # which samples of a (batch*K, d) block each model sees (model_based_rl.py:961-970 + utils.get_ith_tensor)
    K, bs, d, dy = 3, 4, 5, 3
    xb, yb = rng.randn(bs * K, d), rng.randn(bs * K, dy)
    xf, yf = np.reshape(xb, (bs, -1)), np.reshape(yb, (bs, -1))
    arrs.update(split_x=xb, split_y=yb, split_K=np.array(K), split_bs=np.array(bs),
                **{'mx%d' % i: np.array(ref_utils.get_ith_tensor(xf, i, d)) for i in range(K)},
                **{'my%d' % i: np.array(ref_utils.get_ith_tensor(yf, i, dy)) for i in range(K)})
    arrs['baseline_loss'] = np.array(ref_mbrl.compute_baseline_loss(xb, yb, K))
    save('dyn_data', **arrs)



# ---------------------------------------------------------------- 10. collect_data: triplets, train/validation split, normaliser feed
class _RecRms(object):
    def __init__(self): self.calls = []
    def update(self, x): self.calls.append(np.array(x))


def gen_collect_split():
    """Drives the reference's own collect_data (model_based_rl.py:758-857) with sample_trajectories replaced by synthetic
    trajectories (the real one needs MuJoCo); everything after the simulator call is the reference's code."""
    import tempfile
    from collections import OrderedDict
    rng = np.random.RandomState(515)
    ns, na = 4, 2
    lens = [6, 9, 5, 14]                      # 30 triplets: the non-shared split asserts exact divisibility (:852)
    Os = [rng.randn(n, ns) for n in lens]
    As = [rng.randn(n, na) for n in lens]
    ref_mbrl.sample_trajectories = lambda *a, **k: (Os, As, [np.zeros(n) for n in lens], {})
    arrs = dict(lens=np.array(lens), O=np.concatenate(Os), A=np.concatenate(As))
    logger = NS(info=lambda *a, **k: None)
    cases = [('trajectory', True, 0.1, 1), ('triplet', True, 0.33, 1), ('triplet', False, 0.2, 2), ('trajectory', False, 0.25, 3)]
    for ci, (mode, same, ratio, n_scopes) in enumerate(cases):
        scopes = ['training_dynamics%d' % i for i in range(n_scopes)]
        data = OrderedDict((sc, ref_utils.data_collection(max_size=1000)) for sc in scopes)
        val = OrderedDict((sc, ref_utils.data_collection(max_size=1000)) for sc in scopes)
        irms, orms = _RecRms(), _RecRms()
        rp = NS(exploration=None, is_monitored=False, monitorpath='', max_timestep=20, render_every=None, splitting_mode=mode,
                use_same_dataset=same, split_ratio=ratio)
        np.random.seed(1000 + ci)
        with tempfile.TemporaryDirectory() as td:
            ref_mbrl.collect_data(None, 10, data, val, None, None, None, None, None, None, td, rp, 0, logger, None, irms, orms)
            if ci == 0:                                # the file the reference itself wrote (model_based_rl.py:809-811): a data fixture
                import shutil
                shutil.copy(os.path.join(td, 'new_rollouts_0.pkl'), os.path.join(HERE, 'new_rollouts_0.pkl'))
        arrs['c%d_mode' % ci] = np.array(0 if mode == 'trajectory' else 1)
        arrs['c%d_same' % ci] = np.array(int(same)); arrs['c%d_ratio' % ci] = np.array(ratio); arrs['c%d_scopes' % ci] = np.array(n_scopes)
        arrs['c%d_seed' % ci] = np.array(1000 + ci)
        for si, sc in enumerate(scopes):
            arrs['c%d_s%d_tx' % (ci, si)], arrs['c%d_s%d_ty' % (ci, si)] = np.array(data[sc].x), np.array(data[sc].y)
            arrs['c%d_s%d_vx' % (ci, si)], arrs['c%d_s%d_vy' % (ci, si)] = np.array(val[sc].x), np.array(val[sc].y)
        arrs['c%d_n_rms' % ci] = np.array(len(irms.calls))
        for k, (xi, xo) in enumerate(zip(irms.calls, orms.calls)):
            arrs['c%d_rms_in%d' % (ci, k)], arrs['c%d_rms_out%d' % (ci, k)] = xi, xo
    arrs['n_cases'] = np.array(len(cases))
    save('collect_split', **arrs)



# ---------------------------------------------------------------- 11. the reference's own replay-buffer tests (utils.py:145-176)
def gen_reference_buffer_tests():
    """Runs utils.test_data_collection / test_combine_data_collection with get_next_batch recorded: their known answers."""
    log = []
    orig = ref_utils.data_collection.get_next_batch
    def rec(self, batch_size, is_shuffled=False):
        xb, yb = orig(self, batch_size, is_shuffled)
        log.append((np.array(xb), np.array(yb), self.cur_idx, self.n_data))
        return xb, yb
    ref_utils.data_collection.get_next_batch = rec
    np.random.seed(4242)
    try:
        ref_utils.test_data_collection()
    finally:
        ref_utils.data_collection.get_next_batch = orig
    a, b = ref_utils.test_combine_data_collection()
    arrs = dict(n_batches=np.array(len(log)), seed=np.array(4242), comb_ax=np.array(a.x), comb_ay=np.array(a.y), comb_bx=np.array(b.x),
                comb_by=np.array(b.y), comb_a_n=np.array(a.n_data), comb_b_n=np.array(b.n_data), comb_a_cur=np.array(a.cur_idx),
                comb_b_cur=np.array(b.cur_idx), comb_a_max=np.array(a.max_size), comb_b_max=np.array(b.max_size),
                rng_after=np.array(np.random.randint(1 << 30)))
    for i, (xb, yb, cur, n) in enumerate(log):
        arrs['bx%d' % i], arrs['by%d' % i], arrs['cur%d' % i], arrs['n%d' % i] = xb, yb, np.array(cur), np.array(n)
    save('dyn_buffer_reftests', **arrs)


if __name__ == '__main__':
    gen_reference_buffer_tests()
    gen_collect_split()
    gen_dynamics_data()
    gen_rewards()
    gen_vecenv()
    gen_sampler()
    gen_stoplogic()
