"""Synthetic problem instances of the shapes BASELINE.md names: Xavier-uniform weights
(dynamics biases Xavier too -- training.py:179,191-194; policy biases zero), dynamics output layer x0.1
so rollouts stay bounded, identity input normaliser, diff_std = 0.1, init-state pool ~ N(0, 0.1^2)
(Ant z-column ~ U(0.4, 0.8)), log_std = log(init_std)."""
import numpy as np

ENV_SPECS = {'swimmer': (10, 2, 2), 'half_cheetah': (18, 6, 1), 'ant': (29, 8, 2), 'humanoid': (55, 21, 0),
             'hopper': (11, 3, 0), 'snake': (14, 4, 2)}

CONFIGS = {
    # name: env, K, dyn_hidden, pol_hidden, B (whole job), H, gpus BASELINE.json quotes the config on.  bench.py runs B / gpus
    # envs per GPU (the per-GPU share; weak scaling keeps that share fixed when --gpus differs).
    'C0': dict(env='swimmer', K=5, dyn_hidden=(64, 64), pol_hidden=(32, 32), B=100, H=50, gpus=1),
    'C0p': dict(env='swimmer', K=5, dyn_hidden=(512, 512), pol_hidden=(32, 32), B=100, H=200, gpus=1, batch_size=50000),    # params-swimmer.json shape: trpo.batch_size 50 000 -> 3 rounds of 100 envs x 200 steps
    # the other params files of the reference with a horizon-terminated env (2 x 1024 nets, T = 100 / 200): 5 / 3 rounds of 100 envs
    'C0hc': dict(env='half_cheetah', K=5, dyn_hidden=(1024, 1024), pol_hidden=(32, 32), B=100, H=100, gpus=1, batch_size=50000),   # params-half-cheetah.json
    'C0ho': dict(env='hopper', K=5, dyn_hidden=(1024, 1024), pol_hidden=(32, 32), B=100, H=100, gpus=1, batch_size=50000),          # params-hopper.json
    'C0sn': dict(env='snake', K=5, dyn_hidden=(1024, 1024), pol_hidden=(32, 32), B=100, H=200, gpus=1, batch_size=50000),           # params-snake.json
    'C0an': dict(env='ant', K=5, dyn_hidden=(1024, 1024), pol_hidden=(32, 32), B=100, H=100, gpus=1, batch_size=50000),            # params-ant.json (early termination: step-granular stop rule)
    'C0hu': dict(env='humanoid', K=5, dyn_hidden=(1024, 1024), pol_hidden=(100, 50, 25), B=100, H=100, gpus=1, batch_size=50000),  # params-humanoid.json
    'C1': dict(env='swimmer', K=5, dyn_hidden=(64, 64), pol_hidden=(32, 32), B=5000, H=100, gpus=1),
    'C2': dict(env='half_cheetah', K=5, dyn_hidden=(1024, 1024), pol_hidden=(32, 32), B=10000, H=200, gpus=4),  # params-half-cheetah.json nets
    'C2s': dict(env='half_cheetah', K=5, dyn_hidden=(64, 64), pol_hidden=(32, 32), B=10000, H=200, gpus=4),     # BASELINE leaves the MLP open: 2x64 variant
    'C3': dict(env='ant', K=10, dyn_hidden=(512, 512), pol_hidden=(32, 32), B=20000, H=500, gpus=8),
    'C4': dict(env='humanoid', K=20, dyn_hidden=(1024, 1024, 1024), pol_hidden=(100, 50, 25), B=50000, H=1000, gpus=8),
}


def config_from_params(path_or_dict):
    """A CONFIGS row from one of the reference's params files (params.shapes_from_params): `bench.py --params FILE`.  The C0p / C0hc / C0ho / C0sn /
    C0an / C0hu rows above are exactly these for the six shipped files (tests/test_params.py)."""
    from .params import shapes_from_params
    sh = shapes_from_params(path_or_dict)
    return dict(env=sh['env'], K=sh['K'], dyn_hidden=sh['dyn_hidden'], pol_hidden=sh['pol_hidden'], B=sh['n_envs'], H=sh['T'], gpus=1, batch_size=sh['batch_size'])


def make_dynamics(env, K, dyn_hidden, seed=0):
    ns, na, n_drop = ENV_SPECS[env]
    rng = np.random.RandomState(seed)
    dims = [ns + na - n_drop] + list(dyn_hidden) + [ns]
    Ws, bs = [], []
    for l in range(len(dims) - 1):
        lim = np.sqrt(6.0 / (dims[l] + dims[l + 1]))
        blim = np.sqrt(6.0 / (dims[l + 1] + 1))
        W = rng.uniform(-lim, lim, size=(K, dims[l], dims[l + 1]))
        b = rng.uniform(-blim, blim, size=(K, dims[l + 1]))
        if l == len(dims) - 2:
            W, b = W * 0.1, b * 0.1
        Ws.append(W.astype(np.float32)); bs.append(b.astype(np.float32))
    norm = dict(in_mean=np.zeros(ns + na, np.float32), in_std=np.ones(ns + na, np.float32),
                diff_mean=np.zeros(ns, np.float32), diff_std=np.full(ns, 0.1, np.float32))
    return Ws, bs, norm


def make_pool(env, n_pool=4096, seed=1):
    ns = ENV_SPECS[env][0]
    rng = np.random.RandomState(seed)
    pool = rng.randn(n_pool, ns) * 0.1
    if env == 'ant':
        pool[:, 2] = rng.uniform(0.4, 0.8, size=n_pool)
    return pool.astype(np.float32)
