"""The imagined (learned-model) environment, mirroring the reference's interface:

    NeuralNetEnv(env, inner_env, cost_np, dynamics_in, dynamics_outs, sam_mode)   env_helpers.py:532-572
        .vectorized  .observation_space  .action_space  .reset()  .vec_env_executor(n_envs, max_path_length)
    VecSimpleEnv(env, n_envs, max_path_length)                                    env_helpers.py:575-635
        .num_envs  .reset(dones=None)  .step(actions) -> (obs, rewards, dones, env_infos)  .terminate()

In the reference `dynamics_in` / `dynamics_outs` are a TF placeholder and the K output tensors and
`cost_np` is the env's cost_np_vec; here `dynamics_outs` is the Engine holding the K-head ensemble on
the GPU (dynamics_in is unused) and the analytic cost/termination of `env_name` is fused into the
step kernel.  `env` supplies initial states: anything with .reset() (the real simulator, as in the
reference, env_helpers.py:552-555) or an InitStatePool (device-resident, vectorised)."""
import numpy as np
import torch


class Box(object):
    """Minimal stand-in for rllab's Box space (bounds, shape, flatten_n)."""

    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)

    @property
    def shape(self):
        return self.low.shape

    @property
    def bounds(self):
        return self.low, self.high

    @property
    def flat_dim(self):
        return int(np.prod(self.low.shape))

    def flatten_n(self, xs):
        xs = np.asarray(xs)
        return xs.reshape((xs.shape[0], -1))


class EnvSpec(object):
    def __init__(self, observation_space, action_space):
        self.observation_space, self.action_space = observation_space, action_space


class InitStatePool(object):
    """A pool of initial states standing in for the real simulator's reset().  reset() hands out rows
    in order (deterministic, like a recorded list of real resets); the fused rollout indexes it on
    the device instead."""

    def __init__(self, states, na):
        self.states = np.asarray(states, dtype=np.float64)
        ns = self.states.shape[1]
        self.observation_space = Box(-np.inf * np.ones(ns), np.inf * np.ones(ns))
        self.action_space = Box(-np.ones(na), np.ones(na))           # rllab normalize(): bounds are +-1
        self._i = 0
        self._dev = None

    def reset(self):
        s = self.states[self._i % len(self.states)].copy()
        self._i += 1
        return s

    def device_tensor(self, device):
        if self._dev is None or self._dev.device != device:
            self._dev = torch.as_tensor(self.states, dtype=torch.float32, device=device).contiguous()
        return self._dev


class NeuralNetEnv(object):
    def __init__(self, env, inner_env, cost_np, dynamics_in, dynamics_outs, sam_mode):
        self.vectorized = True
        self.env = env
        self.inner_env = inner_env
        self.cost_np = cost_np                      # kept for interface parity; the fused kernel uses engine.env_name
        self.engine = dynamics_outs
        self.dynamics_in = dynamics_in
        self.n_models = self.engine.K
        self.sam_mode = sam_mode
        from ._lib import SAM_MODES
        assert sam_mode in SAM_MODES, "sam mode %s is not defined." % sam_mode      # env_helpers.py:634

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def spec(self):
        return EnvSpec(self.observation_space, self.action_space)

    def reset(self):
        self._state = np.asarray(self.env.reset(), dtype=np.float64)
        return np.copy(self._state)

    def step(self, action):
        """Single-env step (env_helpers.py:557-566); unused on the sampler path, kept for parity."""
        action = np.clip(action, *self.action_space.bounds)
        index = np.random.randint(self.n_models)
        s_next, rew, done = self.engine.step(self._state[None], action[None], 'eps_rand', np.array([index]))
        self._state = s_next[0].double().cpu().numpy()
        return self._state, float(rew[0].item()), bool(done[0].item()), {}

    def vec_env_executor(self, n_envs, max_path_length, fused=False):
        """fused=True (used by the fused sampler, which never steps the env from the host) skips the per-env host state."""
        return VecSimpleEnv(env=self, n_envs=n_envs, max_path_length=max_path_length, fused=fused)


class VecSimpleEnv(object):
    """Step-granular vectorised env.  States live on the device; per-step host work is the reset of
    finished envs (a host call per env when `env.env` is a real simulator, exactly as the reference)."""

    def __init__(self, env, n_envs, max_path_length, fused=False):
        self.env = env
        self.n_envs = self.num_envs = n_envs
        self.engine = env.engine
        self.max_path_length = max_path_length
        self.states = self.ts = self.cur_model_idx = None
        if not fused:                              # fused: the whole loop runs inside metrpo_rollout; host state made on demand
            self._materialise()

    def _materialise(self):
        n_envs, env = self.n_envs, self.env
        dev = self.engine.device
        self.states = torch.zeros(n_envs, self.engine.ns, dtype=torch.float32, device=dev)
        self.ts = np.zeros((n_envs,))
        self.cur_model_idx = np.random.randint(env.n_models, size=(n_envs,))      # env_helpers.py:583

    def reset(self, dones=None):
        if self.states is None:
            self._materialise()
        if dones is None:
            dones = np.asarray([True] * self.n_envs)
        else:
            dones = np.asarray(dones, dtype=bool)
        idx = np.nonzero(dones)[0]
        new = np.empty((len(idx), self.engine.ns))
        for j, i in enumerate(idx):                                  # index order, one reset + one randint each (:590-593)
            new[j] = self.env.reset()
            self.cur_model_idx[i] = np.random.randint(self.env.n_models)
        if len(idx):
            self.states[torch.as_tensor(idx, device=self.states.device)] = torch.as_tensor(new, dtype=torch.float32,
                                                                                           device=self.states.device)
        self.ts[dones] = 0
        return new

    def step(self, actions):
        if self.states is None:
            self._materialise()
        self.ts += 1
        sam_mode = self.env.sam_mode
        idx = noise = None
        if sam_mode == 'step_rand':
            idx = np.random.randint(self.env.n_models, size=self.n_envs)          # :619
        elif sam_mode == 'eps_rand':
            idx = self.cur_model_idx
        elif sam_mode == 'model_mean_std':
            noise = np.random.normal(size=(self.n_envs, self.engine.ns))          # :626
        s_next, rewards, dones = self.engine.step(self.states, np.asarray(actions), sam_mode, idx, noise)
        self.states = s_next
        dones = dones.cpu().numpy().astype(bool)
        dones[self.ts >= self.max_path_length] = True
        if np.any(dones):
            self.reset(dones)
        return self.states.double().cpu().numpy(), rewards.double().cpu().numpy(), dones, dict()

    def terminate(self):
        pass
