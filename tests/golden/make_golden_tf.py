#!/usr/bin/env python3
"""Golden vectors for the TF-GRAPH half of the path, produced by the reference's OWN graph-building code.

Build container only (reads /root/reference; the .npz it writes are the committed fixtures).  TensorFlow 1.4 is not
installable here, so `tensorflow` is tests/golden/tf_shim.py (an eager torch-backed stand-in for the ~60 tf.* calls
the reference makes) and the reference code below runs UNMODIFIED on it:

  training.train(variant)              run for real up to its train_models(...) call (which is intercepted: it needs
                                       MuJoCo); this executes build_policy_from_rllab / build_dynamics_model and hands
                                       over the reference's closures  policy_model (training.py:96-117),
                                       dynamics_model (:218-269 via prepare_input :125-169, build_ff_neural_net
                                       :171-214), get_regularizer_loss (:271-282), RunningMeanStd objects (:320-323)
  running_mean_std.RunningMeanStd      update() + the mean/std expressions (:22-27, :35-42)
  model_based_rl.build_dynamics_graph  (:23-104)  K-head outputs, per-model prediction / regulariser losses
  model_based_rl.build_policy_graph    (:106-151) per-model validation costs incl. Ant's dones mask, n_saturates
  model_based_rl.get_dynamics_optimizer / get_policy_optimizer (:154-206) + utils.minimize_and_clip (:262-276)
                                       gradients of the reference's loss graphs by autograd; Adam / SGD / clip_by_norm
                                       steps use the shim's restatement of TF's documented update rules
  envs/*.cost_tf, AntEnv.is_done_tf

rllab's GaussianMLPPolicy (absent) is stood in for by `FakeGaussianMLPPolicy` below: it only OWNS the variables
(hidden W/b, tanh, output layer, log_std); the mean-net arithmetic that is pinned is the reference's policy_model.

Usage:  python tests/golden/make_golden_tf.py        (from the repo root)
"""
import sys
sys.dont_write_bytecode = True
import os, json, tempfile, copy
from types import SimpleNamespace as NS
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tf_shim                                                    # noqa: E402
tf = tf_shim.install()                                            # must precede every reference import
import make_golden as MG                                          # noqa: E402  (stub finder + reference imports)
import torch                                                      # noqa: E402

import training as ref_training                                   # noqa: E402
import model_based_rl as ref_mbrl                                 # noqa: E402
import utils as ref_utils                                         # noqa: E402
from running_mean_std import RunningMeanStd                       # noqa: E402
import rllab.misc.logger as rllab_logger                          # noqa: E402
import rllab.config as rllab_config                               # noqa: E402

REF = MG.REF
ENV_CLASSES = {k: v[0] for k, v in MG.REF_ENVS.items()}
PARAM_FILE = {'swimmer': 'params-swimmer.json', 'half_cheetah': 'params-half-cheetah.json', 'ant': 'params-ant.json',
              'humanoid': 'params-humanoid.json', 'hopper': 'params-hopper.json', 'snake': 'params-snake.json'}
DIMS = {'swimmer': (10, 2), 'half_cheetah': (18, 6), 'ant': (29, 8), 'humanoid': (55, 21), 'hopper': (11, 3), 'snake': (14, 4)}


# ------------------------------------------------------------------ stand-ins for what is absent (rllab, MuJoCo)
class _Box(object):
    def __init__(self, n, bound=None):
        self.shape = (n,)
        self.low, self.high = -np.ones(n) * (bound or np.inf), np.ones(n) * (bound or np.inf)
        self.bounds = (self.low, self.high)
        self.flat_dim = n


class FakeGaussianMLPPolicy(object):
    """Variable container with the attribute surface training.py:84-117 touches."""

    def __init__(self, name, env_spec, hidden_sizes, init_std, output_nonlinearity, **kw):
        n_in, na = env_spec.observation_space.shape[0], env_spec.action_space.shape[0]
        xav = tf.contrib.layers.xavier_initializer()
        layers = [NS(shape=(None, n_in))]
        with tf.variable_scope(name):
            with tf.variable_scope('mean_network'):
                sizes = list(hidden_sizes) + [na]
                for i, n_out in enumerate(sizes):
                    lname = 'output' if i == len(sizes) - 1 else 'hidden_%d' % i
                    W = tf.get_variable(lname + '/W', shape=(n_in, n_out), initializer=xav)
                    b = tf.get_variable(lname + '/b', shape=(n_out,), initializer=tf.constant_initializer(0.0))
                    nonlin = output_nonlinearity if i == len(sizes) - 1 else tf.nn.tanh
                    layers.append(NS(W=W, b=b, nonlinearity=nonlin))
                    n_in = n_out
            log_std = tf.get_variable('output_std_param/param', shape=(na,), initializer=tf.constant_initializer(np.log(init_std)))
        self._mean_network = NS(layers=layers)
        self._l_std_param = NS(param=log_std)


def fake_env(env_name):
    cls = ENV_CLASSES[env_name]
    inner = cls()                                                 # the reference's own constructor defaults (ctrl_cost_coeff ...)
    ns, na = DIMS[env_name]
    inner.observation_space, inner.action_space = _Box(ns), _Box(na, 1.0)
    env = NS(_wrapped_env=NS(_wrapped_env=inner), observation_space=inner.observation_space, action_space=inner.action_space,
             spec=NS(observation_space=inner.observation_space, action_space=inner.action_space))
    return env, inner


def capture_reference_closures(env_name, n_models, dyn_hidden, pol_hidden, T, gamma, reg_constant, clip, seed):
    """training.train(variant) with the simulator / rllab pieces replaced and train_models intercepted."""
    with open(os.path.join(REF, 'params', PARAM_FILE[env_name])) as f:
        params = json.load(f)                                     # the reference's params file, sizes overridden
    params['env'] = env_name
    params['n_models'] = n_models
    params['policy']['hidden_layers'] = list(pol_hidden)
    params['dynamics_model']['hidden_layers'] = list(dyn_hidden)
    params['dynamics_model']['nonlinearity'] = ['tf.nn.relu'] * len(dyn_hidden)
    params['dynamics_model']['regularization'] = {'method': 'tf.nn.l2_loss', 'constant': reg_constant}
    params['policy_opt_params'].update(T=T, oracle_maxtimestep=T, gamma=gamma, grad_norm_clipping=clip)
    params['rollout_params']['max_timestep'] = T
    env, inner = fake_env(env_name)
    captured = {}
    tmp = tempfile.mkdtemp(prefix='metrpo_golden_')
    saved = (ref_training.get_env, ref_training.train_models)
    import sandbox.rocky.tf.policies.gaussian_mlp_policy as gmp
    import rllab.baselines.linear_feature_baseline as lfb
    import algos.trpo as ref_trpo
    gmp.GaussianMLPPolicy = FakeGaussianMLPPolicy
    lfb.LinearFeatureBaseline = lambda **k: NS(**k)
    saved_trpo = ref_trpo.TRPO
    ref_trpo.TRPO = lambda **k: NS(**k)
    rllab_logger.get_snapshot_dir = lambda: tmp
    rllab_logger.log = lambda *a, **k: None
    rllab_config.PROJECT_PATH = tmp
    ref_training.get_env = lambda name: env
    ref_training.train_models = lambda **kw: captured.update(kw)
    try:
        ref_training.train(dict(seed=seed, mode='batch', use_gpu=False, params=params))
    finally:
        ref_training.get_env, ref_training.train_models = saved
        ref_trpo.TRPO = saved_trpo
    assert 'dynamics_model' in captured, 'training.train did not reach train_models (it swallows exceptions): see above'
    captured['inner_env'], captured['params'] = inner, params
    return captured


# ------------------------------------------------------------------ helpers
def policy_theta():
    """rllab flat order: W0,b0,...,Wout,bout,log_std."""
    vs = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, scope='training_policy')
    byname = {v.name[:-2]: v for v in vs}
    names = sorted([n for n in byname if 'hidden_' in n and n.endswith('/W')], key=lambda n: int(n.split('hidden_')[1].split('/')[0]))
    parts = []
    for n in names:
        parts += [byname[n].numpy().ravel(), byname[n[:-2] + '/b'].numpy().ravel()]
    parts += [byname['training_policy/mean_network/output/W'].numpy().ravel(), byname['training_policy/mean_network/output/b'].numpy().ravel(),
              byname['training_policy/output_std_param/param'].numpy().ravel()]
    return np.concatenate(parts)


def grads_flat_policy(grads_and_vars):
    g = {v.name[:-2]: (np.zeros(v.tensor.shape) if gr is None else gr.detach().numpy()) for gr, v in grads_and_vars}
    names = sorted([n for n in g if 'hidden_' in n and n.endswith('/W')], key=lambda n: int(n.split('hidden_')[1].split('/')[0]))
    parts = []
    for n in names:
        parts += [g[n].ravel(), g[n[:-2] + '/b'].ravel()]
    parts += [g['training_policy/mean_network/output/W'].ravel(), g['training_policy/mean_network/output/b'].ravel(),
              g['training_policy/output_std_param/param'].ravel()]
    return np.concatenate(parts)


def dyn_weights(scope, K, L):
    byname = {v.name[:-2]: v for v in tf.get_collection(tf.GraphKeys.GLOBAL_VARIABLES, scope=scope + '/')}
    d = {}
    for l in range(L):
        d['dynW%d' % l] = np.stack([byname['%s/model%d/layer%d/weights' % (scope, k, l)].numpy() for k in range(K)])
        d['dynb%d' % l] = np.stack([byname['%s/model%d/layer%d/biases' % (scope, k, l)].numpy() for k in range(K)])
    return d


LOGGER = NS(info=lambda *a, **k: None, debug=lambda *a, **k: None)

CASES = [
    # name, env, K, dyn_hidden, pol_hidden, B, T, gamma, reg, clip
    ('swimmer_2x64', 'swimmer', 5, (64, 64), (32, 32), 24, 6, 0.99, 1e-4, 10.0),
    ('half_cheetah_2x64', 'half_cheetah', 3, (64, 64), (32, 32), 16, 5, 1.0, 0.0, 10.0),
    ('ant_2x64', 'ant', 3, (64, 64), (32, 32), 20, 6, 0.97, 0.0, 0.05),
    ('swimmer_2x512', 'swimmer', 2, (512, 512), (32, 32), 8, 3, 1.0, 0.0, 10.0),
    ('humanoid_3x128', 'humanoid', 2, (128, 128, 128), (100, 50, 25), 8, 3, 0.99, 0.0, 10.0),
    ('hopper_2x32', 'hopper', 2, (32, 32), (16, 16), 16, 4, 1.0, 0.0, 10.0),
    ('snake_2x32', 'snake', 2, (32, 32), (16, 16), 16, 4, 1.0, 0.0, 10.0),
]


def gen_case(name, env_name, K, dyn_hidden, pol_hidden, B, T, gamma, reg, clip, seed):
    light = max(dyn_hidden) > 128                                 # big nets: forward / losses / costs only (fixture size)
    tf.get_default_session = tf_shim.get_default_session
    rng = np.random.RandomState(seed)
    ns, na = DIMS[env_name]
    cap = capture_reference_closures(env_name, K, dyn_hidden, pol_hidden, T, gamma, reg, clip, seed)
    dynamics_model, policy_model = cap['dynamics_model'], cap['policy_model']
    get_reg, cost_tf, inner = cap['get_regularizer_loss'], cap['cost_tf'], cap['inner_env']
    input_rms, diff_rms = cap['input_rms'], cap['diff_rms']
    params = cap['params']
    pop = ref_training.Policy_opt_params(**{**params['policy_opt_params'], 'stop_critereon': None})
    dop = ref_training.Dynamics_opt_params(**{**params['dynamics_opt_params'], 'stop_critereon': None})
    sess = tf.get_default_session()
    scope = 'training_dynamics'
    L = len(dyn_hidden) + 1
    env = NS(observation_space=inner.observation_space, action_space=inner.action_space)

    # ---- normalisers: feed the reference's RunningMeanStd.update (running_mean_std.py:35-42); column 1 gets a tiny
    # variance so the 0.1 std floor (:23-27) is active, the rest generic
    xs = rng.randn(200, ns + na) * rng.uniform(0.3, 2.0, size=ns + na) + rng.randn(ns + na) * 0.3
    xs[:, 1] = 0.25 + 0.01 * rng.randn(200)
    ds = rng.randn(200, ns) * rng.uniform(0.05, 0.5, size=ns) + rng.randn(ns) * 0.02
    ds[:, 3] = 0.001 * rng.randn(200)
    input_rms.update(xs[:120]); input_rms.update(xs[120:])
    diff_rms.update(ds)
    in_mean, in_std = sess.run([input_rms.mean, input_rms.std])
    diff_mean, diff_std = sess.run([diff_rms.mean, diff_rms.std])
    out = dict(env=np.array(env_name), K=np.array(K), dyn_hidden=np.array(dyn_hidden), pol_hidden=np.array(pol_hidden),
               B=np.array(B), T=np.array(T), gamma=np.array(gamma), reg_constant=np.array(reg), clip=np.array(clip),
               in_mean=in_mean, in_std=in_std, diff_mean=diff_mean, diff_std=diff_std,
               rms_in_sum=input_rms._sum.numpy(), rms_in_sumsq=input_rms._sumsq.numpy(), rms_in_count=input_rms._count.numpy(),
               rms_diff_sum=diff_rms._sum.numpy(), rms_diff_sumsq=diff_rms._sumsq.numpy(), rms_diff_count=diff_rms._count.numpy(),
               n_drop=np.array(2 if params['dynamics_model'].get('ignore_xy_input') else 1 if params['dynamics_model'].get('ignore_x_input') else 0))

    # ---- a1 + f1 losses: model_based_rl.build_dynamics_graph with concrete tensors in place of the placeholders
    n_in = ns + na
    xu = np.concatenate([rng.randn(B, ns) * 0.5, rng.uniform(-1, 1, size=(B, na))], axis=1)
    if env_name == 'ant':
        xu[:, 2] = rng.uniform(0.3, 0.9, size=B)
    bs = 12
    x_full = np.concatenate([rng.randn(bs * K, ns) * 0.5, rng.uniform(-1, 1, size=(bs * K, na))], axis=1)
    y_full = x_full[:, :ns] + rng.randn(bs * K, ns) * 0.1
    xf, yf = np.reshape(x_full, (bs, -1)), np.reshape(y_full, (bs, -1))      # model_based_rl.py:961-966
    dyn_loss, pred_loss, reg_loss, dyn_outs, dyn_losses = ref_mbrl.build_dynamics_graph(
        scope, dynamics_model, tf_shim._t(xu), tf_shim._t(xf), tf_shim._t(yf), n_in, K, get_reg, ns, LOGGER)
    # dynamics weights: rescale the output layer (x0.3) so multi-step rollouts stay bounded, and make every weight exactly
    # float32-representable (the fixture stores them as float32; the graphs are then re-evaluated on the stored values)
    for v in tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, scope=scope + '/'):
        w = v.numpy() * (0.3 if '/layer%d/' % (L - 1) in v.name else 1.0)
        v.set(w.astype(np.float32).astype(np.float64))
    dyn_loss, pred_loss, reg_loss, dyn_outs, dyn_losses = ref_mbrl.build_dynamics_graph(
        scope, dynamics_model, tf_shim._t(xu), tf_shim._t(xf), tf_shim._t(yf), n_in, K, get_reg, ns, LOGGER)
    out.update({k_: v_.astype(np.float32) for k_, v_ in dyn_weights(scope, K, L).items()})
    out.update(xu=xu, dyn_out=np.stack(sess.run(dyn_outs)), train_x=x_full, train_y=y_full, train_bs=np.array(bs),
               dynamics_losses=np.array(sess.run(dyn_losses)), prediction_loss=sess.run(pred_loss), regularizer_loss=sess.run(reg_loss),
               dynamics_loss=sess.run(dyn_loss))
    # gradient of the reference's prediction loss (autograd over the reference's graph)
    tvars = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, scope=scope + '/')
    gv = tf.train.AdamOptimizer(1e-3).compute_gradients(pred_loss, var_list=tvars)
    for l in range(L if not light else 0):
        byname = {v.name[:-2]: g for g, v in gv}
        out['gradW%d' % l] = np.stack([byname['%s/model%d/layer%d/weights' % (scope, k, l)].detach().numpy() for k in range(K)])
        out['gradb%d' % l] = np.stack([byname['%s/model%d/layer%d/biases' % (scope, k, l)].detach().numpy() for k in range(K)])

    # ---- a3: policy mean (training.py:96-117), random theta incl. biases and log_std
    for v in tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, scope='training_policy'):
        if v.name.endswith('/b:0'):
            v.set(rng.randn(*v.tensor.shape) * 0.1)
        if 'output_std_param' in v.name:
            v.set(rng.randn(*v.tensor.shape) * 0.2 - 0.5)
        v.set(v.numpy().astype(np.float32).astype(np.float64))
    obs = rng.randn(B, ns) * 0.7
    out.update(theta=policy_theta(), obs=obs, policy_mean=sess.run(policy_model(tf_shim._t(obs))))

    # ---- a18 + f3: build_policy_graph (per-model costs), then mean over models (model_based_rl.py:365) and
    # get_policy_optimizer (Adam on the per-variable clip_by_norm'ed gradient)
    x0 = rng.randn(B, ns) * 0.3
    if env_name == 'ant':
        x0[:, 2] = rng.uniform(0.21, 0.6, size=B)               # some trajectories leave [0.2, 1.0] within T steps
    if env_name == 'humanoid':
        x0[:, -1] = 1.4 + 0.1 * rng.randn(B)
    if env_name == 'hopper':
        x0[:, 0] = rng.uniform(0.3, 0.7, size=B)
    is_done_tf = getattr(inner, 'is_done_tf', None)             # model_based_rl.py:260
    stochastic = tf.Variable(0.0, trainable=False)              # :345
    humanoid = env_name == 'humanoid'                           # its cost_tf applies np.square to a tensor: forward only
    ctx = torch.no_grad() if humanoid else torch.enable_grad()
    with ctx:
        costs, n_sat = ref_mbrl.build_policy_graph('training_policy', scope, tf_shim._t(x0), K, pop, policy_model, dynamics_model, env,
                                                   cost_tf, LOGGER, is_done_tf, stochastic)
    out.update(x0=x0, policy_costs=np.array(sess.run(costs)), n_saturates=np.asarray(sess.run(n_sat)))
    if not humanoid:
        training_policy_cost = tf.reduce_mean(costs)            # model_based_rl.py:365
        out['training_policy_cost'] = sess.run(training_policy_cost)
        opt_op, adam_init, grads_and_vars = ref_mbrl.get_policy_optimizer('training_policy', training_policy_cost, pop, LOGGER)
        out['bptt_grad'] = grads_flat_policy(grads_and_vars)     # unclipped gradient (policy_grads_and_vars, :196-199)
        thetas = [policy_theta()]                                # get_policy_optimizer applied one Adam step already
        for _ in range(2):                                       # two more sess.run(policy_opt_op): rebuild = re-evaluate
            costs, _ = ref_mbrl.build_policy_graph('training_policy', scope, tf_shim._t(x0), K, pop, policy_model, dynamics_model, env,
                                                   cost_tf, LOGGER, is_done_tf, stochastic)
            ref_mbrl.get_policy_optimizer('training_policy', tf.reduce_mean(costs), pop, LOGGER)
            thetas.append(policy_theta())
        out['bptt_thetas'] = np.stack(thetas)
        out['bptt_lr'] = np.array(pop.learning_rate)

    # ---- f1: three optimiser steps on the dynamics (Adam on the prediction loss, SGD on the regulariser, :154-183)
    lr = dop.learning_rate['scratch'] if isinstance(dop.learning_rate, dict) else dop.learning_rate
    steps_W = []
    losses = []
    for it in range(3):
        xb = np.concatenate([rng.randn(bs * K, ns) * 0.5, rng.uniform(-1, 1, size=(bs * K, na))], axis=1)
        yb = xb[:, :ns] + rng.randn(bs * K, ns) * 0.1
        out['step%d_x' % it], out['step%d_y' % it] = xb, yb
        dl, pl, rl, _, _ = ref_mbrl.build_dynamics_graph(scope, dynamics_model, tf_shim._t(xu), tf_shim._t(np.reshape(xb, (bs, -1))),
                                                          tf_shim._t(np.reshape(yb, (bs, -1))), n_in, K, get_reg, ns, LOGGER)
        losses.append(sess.run(dl))
        ref_mbrl.get_dynamics_optimizer(scope, {scope: pl}, {scope: rl}, dop, LOGGER)
        if it == 2 and not light:
            for k_, v_ in dyn_weights(scope, K, L).items():
                out['step%d_%s' % (it, k_)] = v_
    out['step_losses'] = np.array(losses)
    out['train_lr'] = np.array(lr)
    MG.save('tfgraph_' + name, **out)


def gen_rms_reference_test():
    """running_mean_std.test_runningmeanstd (its own known-answer test) executed as written."""
    tf.get_default_session = tf_shim.get_default_session
    tf.reset_default_graph()
    import running_mean_std as rms_mod
    np.random.seed(0)
    rms_mod.test_runningmeanstd()                                 # asserts inside; passing = the shim's graph semantics are sane
    print('reference test_runningmeanstd passed under the shim')


if __name__ == '__main__':
    gen_rms_reference_test()
    for i, c in enumerate(CASES):
        tf.reset_default_graph()
        gen_case(*c, seed=900 + i)
