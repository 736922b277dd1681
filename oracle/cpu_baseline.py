"""Structure-faithful CPU baseline of one ME-TRPO inner-loop iteration.

TEST/BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py): `bench.py`'s cpu_baseline leg times this
on the GPU box's host cores.  TensorFlow 1.4 / rllab cannot be installed here, so the reference's
path is restated with the SAME control flow: per time step one policy call and one call evaluating
all K dynamics heads on the (B, ns+na) batch (samplers/vectorized_sampler.py:60-108 +
env_helpers.py:597-635), NumPy selection/reward/done/reset, the per-env Python bookkeeping loop,
per-path GAE, linear-feature baseline, and a host CG with ~28 full-batch function evaluations
([rllab] ConjugateGradientOptimizer).  MLP arithmetic in float32 (the TF graph dtype); one thread,
as the reference configures (utils.py:229-232)."""
import time
import numpy as np
from . import metrpo_oracle as O


def run_iteration(env='swimmer', K=5, dyn_hidden=(64, 64), pol_hidden=(32, 32), B=5000, H=100, seed=0):
    dm, theta, pdims, pool = O.make_problem(env, K=K, dyn_hidden=dyn_hidden, pol_hidden=pol_hidden, seed=seed)
    dm32, theta32 = dm.astype(np.float32), theta.astype(np.float32)
    pool_iter = iter(range(10 ** 9))
    reset_fn = lambda: pool[next(pool_iter) % len(pool)].copy()
    t = {}
    t0 = time.perf_counter()
    ve = O.VecEnvOracle(env, lambda s, a: O.dynamics_forward_all(dm32, s.astype(np.float32), a.astype(np.float32)).astype(np.float64),
                        K, B, dm.ns, H, 'step_rand', reset_fn)

    def get_actions(obs):
        a, info = O.policy_get_actions(theta32, pdims, np.asarray(obs, dtype=np.float32),
                                       np.random.normal(size=(len(obs), pdims[-1])).astype(np.float32))
        return a.astype(np.float64), {k: v.astype(np.float64) for k, v in info.items()}

    paths = O.obtain_samples(ve, get_actions, B * H)
    t['obtain_samples'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    samples = O.process_samples(paths, O.LinearFeatureBaselineOracle(), 1.0, 1.0)
    t['process_samples'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    f32 = lambda x: np.asarray(x, dtype=np.float32)
    out = O.cg_optimize(theta32, pdims, f32(samples['observations']), f32(samples['actions']), f32(samples['advantages']),
                        f32(samples['agent_infos']['mean']), f32(samples['agent_infos']['log_std']))
    t['optimize_policy'] = time.perf_counter() - t0
    return dict(units=K * B * H, seconds=sum(t.values()), breakdown=t, accepted=bool(out['accepted']))


if __name__ == '__main__':
    # worker of bench.py's all-cores leg: python -m oracle.cpu_baseline env K dyn_hidden pol_hidden B H seed -> one JSON line
    import json
    import sys
    env, K, dh, ph, B, H, seed = sys.argv[1:8]
    r = run_iteration(env, int(K), tuple(int(x) for x in dh.split(',')), tuple(int(x) for x in ph.split(',')), B=int(B), H=int(H), seed=int(seed))
    print(json.dumps({'units': r['units'], 'seconds': r['seconds']}))
