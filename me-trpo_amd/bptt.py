"""BPTT policy optimisation -- the 'bptt' branch of the reference's optimize_policy:

    model_based_rl.py:1181-1187   x_batch = np.array([env.reset() for i in range(batch_size)])
                                  _, training_cost = sess.run([policy_opt_op, training_policy_cost], {policy_training_init: x_batch})
    :106-151, :365                training_policy_cost = mean_i sum_t gamma^t cost_tf(x_t, clip(policy(x_t)), model_i(x_t, u_t))
    :186-195, utils.py:262-276    policy_opt_op = Adam(learning_rate) on the per-variable clip_by_norm'ed gradient

The unrolled forward, its reverse sweep, the parameter-gradient reduction and the Adam step all run in libmetrpo.so
(csrc/bptt.hip + the gradient kernels of the TRPO update); nothing is differentiated by a framework."""
import numpy as np
import torch


class BPTT(object):
    """policy_opt_params of the reference: T, gamma, learning_rate, grad_norm_clipping, batch_size."""

    def __init__(self, engine, T, gamma=1.0, learning_rate=1e-3, grad_norm_clipping=None, batch_size=100):
        self.engine, self.T, self.gamma = engine, int(T), float(gamma)
        self.learning_rate, self.grad_norm_clipping, self.batch_size = float(learning_rate), grad_norm_clipping, int(batch_size)
        engine.policy_adam_reset()                                   # sess.run(policy_adam_init)

    def reset_optimizer(self):
        self.engine.policy_adam_reset()

    def training_cost_and_grad(self, x_batch):
        costs, grad = self.engine.bptt_grad(x_batch, self.T, self.gamma)
        return costs, grad

    def step(self, x_batch):
        """One sess.run([policy_opt_op, training_policy_cost]): returns the training cost evaluated BEFORE the update (a 0-d device
        tensor; `float()` it to synchronise, as np.squeeze(training_cost) does in the reference)."""
        costs, grad = self.engine.bptt_grad(x_batch, self.T, self.gamma)
        self.engine.policy_adam_step(grad, self.learning_rate, self.grad_norm_clipping)
        return costs.mean()

    def optimize_policy_iteration(self, env_or_pool):
        """:1183: fresh initial states from the real env's reset() (or an InitStatePool), then one step."""
        if hasattr(env_or_pool, 'sample'):
            x_batch = env_or_pool.sample(self.batch_size)
        else:
            x_batch = np.array([env_or_pool.reset() for _ in range(self.batch_size)])
        return self.step(x_batch)
