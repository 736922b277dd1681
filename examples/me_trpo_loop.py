#!/usr/bin/env python3
"""End-to-end ME-TRPO outer loop on one MI355X with every stage on the GPU library (train_models, model_based_rl.py:231-755, reduced
to its data flow): collect -> split / normalise -> train the K-model ensemble -> TRPO on imagined rollouts with validation-cost early
stopping -> repeat.

The real simulator (MuJoCo) is not available here, so the "real environment" is a SURROGATE: one fixed, randomly initialised dynamics
network evaluated through a second Engine (K = 1).  Everything that the reference does between simulator calls is the real thing.

    python examples/me_trpo_loop.py [--outer 3] [--env swimmer]
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metrpo_amd as M                                                        # noqa: E402
from metrpo_amd import synthetic, early_stop                                  # noqa: E402
from metrpo_amd import dynamics_training as DT                                # noqa: E402


class SurrogateRealEnv(object):
    """reset() / step(a) of a 'real' env whose transition is a hidden MLP (one_model on a K=1 engine); cost is the env's analytic cost."""

    def __init__(self, env, hidden, seed, n_parallel):
        self.eng = M.Engine(env, 1, hidden, (32, 32))
        Ws, bs, norm = synthetic.make_dynamics(env, 1, hidden, seed=seed)
        self.eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
        self.eng.set_policy(M.xavier_policy_theta(self.eng.ns, (32, 32), self.eng.na))
        self.pool = synthetic.make_pool(env).astype(np.float32)
        self.rng = np.random.RandomState(seed)
        self.n = n_parallel

    def reset(self):
        return self.pool[self.rng.randint(len(self.pool))]

    def rollouts(self, policy_actions, T, n_traj):
        """n_traj trajectories of T+1 observations / T+1 actions (the reference's sample_trajectories shape) under `policy_actions`."""
        s = self.pool[self.rng.randint(len(self.pool), size=n_traj)]
        Os, As, cost = [s], [], 0.0
        for t in range(T + 1):
            a = policy_actions(s)
            As.append(a)
            if t == T:
                break
            sn, rew, _ = self.eng.step(s, np.clip(a, -1, 1), 'one_model')
            cost += float(-rew.mean())
            s = sn.cpu().numpy()
            Os.append(s)
        O = np.stack(Os, axis=1); A = np.stack(As, axis=1)                   # [n_traj][T+1][.]
        return [O[i] for i in range(n_traj)], [A[i] for i in range(n_traj)], cost


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--env', default='swimmer')
    ap.add_argument('--outer', type=int, default=3)
    ap.add_argument('--K', type=int, default=5)
    ap.add_argument('--T', type=int, default=50)
    ap.add_argument('--traj', type=int, default=60)
    ap.add_argument('--n-envs', type=int, default=1000)
    ap.add_argument('--policy-iters', type=int, default=20)
    ap.add_argument('--model-passes', type=int, default=30)
    ap.add_argument('--quiet', action='store_true')
    args = ap.parse_args(argv)
    say = (lambda *a: None) if args.quiet else print
    np.random.seed(0)
    env, K, T = args.env, args.K, args.T
    eng = M.Engine(env, K, (64, 64), (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, K, (64, 64), seed=1)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    policy = M.GaussianMLPPolicy(eng, init_std=1.0, seed=0)
    real = SurrogateRealEnv(env, (48, 48), seed=7, n_parallel=args.traj)
    ns, na = eng.ns, eng.na
    data = OrderedDict(training_dynamics=DT.data_collection(max_size=50000, device='cuda'))
    val = OrderedDict(training_dynamics=DT.data_collection(max_size=50000, device='cuda'))
    in_rms, out_rms = DT.RunningMeanStd(eng, shape=(ns + na,)), DT.RunningMeanStd(eng, shape=(ns,))
    validation_init = real.pool[:256]
    algo = M.TRPO(env=M.NeuralNetEnv(M.InitStatePool(real.pool, na), None, env, None, eng, 'step_rand'), policy=policy,
                  baseline=M.LinearFeatureBaseline(), batch_size=args.n_envs * T, max_path_length=T, discount=1.0, step_size=0.01,
                  sampler_args=dict(n_envs=args.n_envs))
    history = []
    for it in range(args.outer):
        t0 = time.time()
        # ---- collect_data (model_based_rl.py:758-857): the current policy in the (surrogate) real env, exploration = its own noise
        Os, As, real_cost = real.rollouts(lambda s: policy.get_actions(s)[0], T, args.traj)
        x_all, y_all = DT.trajectories_to_pairs(Os, As)
        DT.add_rollouts(x_all, y_all, data, val, 'triplet', True, 0.1, in_rms, out_rms)
        DT.push_normalizers(eng, in_rms, out_rms)
        t1 = time.time()
        # ---- optimize_models (:881-1051)
        info = DT.optimize_models(eng, data['training_dynamics'], val['training_dynamics'], dict(scratch=1e-3, refine=1e-3), batch_size=256,
                                  max_passes=args.model_passes, log_every=5, num_passes_threshold=10, reinitialize=(it == 0), init_seed=it)
        t2 = time.time()
        # ---- optimize_policy (:1082-1301): TRPO on the imagined env, early stopping on the per-model validation costs
        res = early_stop.optimize_policy(algo, validation_init, T, 1.0, mode='estimated', log_every=5, num_iters_threshold=10,
                                         max_iters=args.policy_iters, reset_log_std=True)
        t3 = time.time()
        history.append(dict(real_cost=real_cost, n_data=data['training_dynamics'].get_num_data(),
                            model_val=float(np.sum(info['min_validation_losses'])) if 'min_validation_losses' in info else float('nan'),
                            est_cost=float(np.mean(res['min_validation_costs']['estimated'])), best_index=res['best_index']))
        say("outer %d: real-cost/step %.4f | data %d | model val loss %.4g (%d updates) | policy est-cost %.4f (best iter %d of %d) | "
            "collect %.2fs models %.2fs policy %.2fs" % (it, real_cost / T, history[-1]['n_data'], history[-1]['model_val'],
                                                       info.get('n_model_updates', -1), history[-1]['est_cost'], res['best_index'],
                                                       res['last_index'], t1 - t0, t2 - t1, t3 - t2))
    return history


if __name__ == '__main__':
    main()
