"""[rllab] sandbox.rocky.tf.policies.gaussian_mlp_policy.GaussianMLPPolicy, as the reference uses it
(training.py:84-90; samplers/vectorized_sampler.py:62-65; algos/npo.py:42-69,99-101): tanh MLP mean
network, free log_std vector clamped below at log(1e-6), flat parameter vector in rllab order
[W0,b0,...,Wout,bout,log_std].  Parameters live on the GPU inside the Engine."""
import numpy as np
import torch

from .engine import xavier_policy_theta


class DiagonalGaussian(object):
    dist_info_keys = ['mean', 'log_std']

    def __init__(self, dim):
        self.dim = dim

    def entropy(self, dist_info):
        return np.sum(np.asarray(dist_info['log_std']) + np.log(np.sqrt(2 * np.pi * np.e)), axis=-1)


class GaussianMLPPolicy(object):
    vectorized = True
    recurrent = False
    state_info_keys = []

    def __init__(self, engine, init_std=1.0, seed=0, name='training_policy'):
        self.engine, self.name, self.init_std = engine, name, init_std
        self.distribution = DiagonalGaussian(engine.na)
        self.engine.set_policy(xavier_policy_theta(engine.ns, engine.pol_hidden, engine.na, init_std, seed))

    # -- rllab Parameterized surface -----------------------------------------------------------
    def get_param_values(self, trainable=True):
        return self.engine.get_policy().double().cpu().numpy()

    def set_param_values(self, flat, trainable=True):
        self.engine.set_policy(np.asarray(flat, dtype=np.float32))

    def log_std(self):
        return torch.clamp(self.engine.get_policy()[-self.engine.na:], min=float(np.log(1e-6)))

    def reset_log_std(self):
        """training.py:368-370 `reset_opt`: log_std <- log(init_std) at the start of every outer sweep."""
        th = self.engine.get_policy()
        th[-self.engine.na:] = float(np.log(self.init_std))
        self.engine.set_policy(th)

    def reset(self, dones=None):
        pass

    def get_actions(self, observations):
        """-> actions (B,na), dict(mean, log_std); eps ~ np.random.normal(size=mean.shape) as rllab does."""
        obs = np.asarray(observations)
        eps = np.random.normal(size=(obs.shape[0], self.engine.na))
        actions, mean = self.engine.policy_actions(obs, eps)
        mean = mean.double().cpu().numpy()
        log_std = np.broadcast_to(self.log_std().double().cpu().numpy(), mean.shape).copy()
        return actions.double().cpu().numpy(), dict(mean=mean, log_std=log_std)

    def get_action(self, observation):
        a, info = self.get_actions(np.asarray(observation)[None])
        return a[0], {k: v[0] for k, v in info.items()}
