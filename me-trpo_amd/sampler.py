"""samplers/vectorized_sampler.py + samplers/base.py of the reference, on the GPU.

    VectorizedSampler(algo, n_envs=None)
        .start_worker()                       vectorized_sampler.py:23-40   (n_envs clamp <= 100 lifted when given)
        .obtain_samples(itr, determ=False)    :45-116  -> DevicePaths (time-major device tensors)
        .process_samples(itr, paths)          base.py:48-182 -> samples_data dict (device tensors)
        .shutdown_worker()

obtain_samples runs the whole while-loop of the reference as ONE fused kernel launch per
`max_path_length` steps.  The reference materialises a Python list of path dicts; here `DevicePaths`
keeps [T,B,.] tensors and produces that list only on request (`.to_paths()`), because building
5 000-50 000 dicts per iteration is the reference's own bottleneck."""
import numpy as np
import torch

from .parallel import Comm


class DevicePaths(object):
    """Result of obtain_samples: a Trajectory plus sampler metadata."""

    def __init__(self, traj, log_std, n_complete_samples):
        self.traj, self.log_std, self.n_complete_samples = traj, log_std, n_complete_samples

    def __len__(self):
        return int(self.traj.done.sum().item())

    def to_paths(self):
        """The reference's list of path dicts, in its completion order (time step, then env index);
        trailing unfinished paths are dropped (vectorized_sampler.py:60,104)."""
        tr = self.traj
        obs, act, rew, mean = (x.double().cpu().numpy() for x in (tr.obs, tr.act, tr.rew, tr.mean))
        done = tr.done.cpu().numpy().astype(bool)
        ls = self.log_std.double().cpu().numpy()
        T, B = rew.shape
        start = np.zeros(B, dtype=int)
        paths = []
        for t, b in zip(*np.nonzero(done)):
            sl = slice(start[b], t + 1)
            paths.append(dict(observations=obs[sl, b], actions=act[sl, b], rewards=rew[sl, b], env_infos={},
                              agent_infos=dict(mean=mean[sl, b], log_std=np.broadcast_to(ls, mean[sl, b].shape).copy())))
            start[b] = t + 1
        return paths


class BaseSampler(object):
    def __init__(self, algo):
        self.algo = algo

    def process_samples(self, itr, paths):
        """samplers/base.py:48-104,163-167 (non-recurrent branch).  Advantages use the baseline as fitted
        in the PREVIOUS iteration (:55); the baseline is refit afterwards (:164-167)."""
        algo, eng = self.algo, self.algo.engine
        comm = getattr(algo, 'comm', None) or Comm()
        tr = paths.traj
        self.finish_baseline_fit()             # deferred solve of the previous iteration (overlaps the rollout just launched)
        coeffs = algo.baseline.coeffs
        adv, ret, valid, stats = eng.gae(tr, coeffs, algo.discount, algo.gae_lambda)
        comm.allreduce_sum_(stats)
        # fixed-horizon envs: every sample is valid and the global count is known without a host sync
        static = (eng.env_name != 'ant') and (tr.T % tr.H == 0)
        n_global = comm.world * tr.T * tr.B if static and self._uniform_B(comm, tr.B) else int(stats[2].item())
        if algo.center_adv:
            eng.center_advantages(adv, valid, stats)
        if algo.positive_adv:
            adv.sub_(adv.min()).add_(1e-8)                      # [rllab] util.shift_advantages_to_positive
        AtA, Aty = eng.baseline_gram(tr.obs, ret, tr.tpath, valid)
        gram = torch.cat([AtA.reshape(-1), Aty])
        comm.allreduce_sum_(gram)
        self._pending_gram = gram.to('cpu', non_blocking=True) if gram.is_cuda else gram
        self._gram_event = torch.cuda.Event() if gram.is_cuda else None
        if self._gram_event is not None:
            self._gram_event.record()
        samples_data = dict(observations=tr.obs.reshape(-1, eng.ns), actions=tr.act.reshape(-1, eng.na),
                            rewards=tr.rew.reshape(-1), returns=ret.reshape(-1), advantages=adv.reshape(-1),
                            env_infos={}, agent_infos=dict(mean=tr.mean.reshape(-1, eng.na), log_std=paths.log_std),
                            valids=valid.reshape(-1), n_valid_global=n_global, paths=paths)
        if not getattr(algo, 'defer_baseline_fit', False):
            self.finish_baseline_fit()
        return samples_data

    def _uniform_B(self, comm, B):
        return getattr(self.algo, 'uniform_shards', True)

    def finish_baseline_fit(self):
        """Host solve of the (2ns+4)^2 normal equations.  Can be deferred until after optimize_policy
        (its line search synchronises anyway) so the fit costs no extra host sync."""
        gram = getattr(self, '_pending_gram', None)
        if gram is None:
            return
        if self._gram_event is not None:
            self._gram_event.synchronize()
        F = 2 * self.algo.engine.ns + 4
        g = gram.numpy()
        self.algo.baseline.solve(g[:F * F].reshape(F, F), g[F * F:])
        self._pending_gram = None


class VectorizedSampler(BaseSampler):
    def __init__(self, algo, n_envs=None):
        super(VectorizedSampler, self).__init__(algo)
        self.n_envs = n_envs
        self._itr_seed = 0

    def start_worker(self):
        n_envs = self.n_envs
        if n_envs is None:                                     # reference default (vectorized_sampler.py:24-27)
            n_envs = int(self.algo.batch_size / self.algo.max_path_length)
            n_envs = max(1, min(n_envs, 100))
        assert getattr(self.algo.env, 'vectorized', False), "the imagined env must be vectorized"
        self.vec_env = self.algo.env.vec_env_executor(n_envs=n_envs, max_path_length=self.algo.max_path_length, fused=True)
        self.env_spec = self.algo.env.spec
        self._n_envs = n_envs

    def shutdown_worker(self):
        self.vec_env.terminate()

    def obtain_samples(self, itr, determ=False, draws=None):
        """Loop `while n_samples < batch_size` (:60) with n_samples counting completed paths only (:104).
        Fixed-horizon envs: ceil(batch/(n_envs*H)) rounds of H steps in ONE launch.  Ant (early
        termination): launches of H steps until enough completed samples exist."""
        algo, eng = self.algo, self.algo.engine
        comm = getattr(algo, 'comm', None) or Comm()
        B, H = self._n_envs, algo.max_path_length
        nne = algo.env
        pool = nne.env.device_tensor(eng.device) if hasattr(nne.env, 'device_tensor') else \
            torch.as_tensor(np.stack([nne.env.reset() for _ in range(max(B, 1))]), dtype=torch.float32, device=eng.device)
        rounds = max(1, -(-algo.batch_size // (B * H)))
        T = rounds * H
        seed = (getattr(algo, 'seed', 0) * 1000003 + itr * 7919 + (1 if determ else 0)) & 0xFFFFFFFFFFFFFFFF
        offset = comm.rank * B
        draws = draws or {}
        ev = getattr(algo, 'rollout_events', None)          # optional [(start, end)] HIP-event pairs around the launch
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out = None
        if getattr(algo, 'reuse_trajectory_buffers', False):      # opt-in: later rollouts overwrite earlier DevicePaths
            key = (B, T, H)
            if getattr(self, '_traj_key', None) != key:
                self._traj_buf, self._traj_key = eng.alloc_trajectory(B, T, H), key
            out = self._traj_buf
        traj = eng.rollout(B, T, H, nne.sam_mode, pool, determ=determ, eval_all_heads=getattr(algo, 'eval_all_heads', True),
                           seed=seed, stream_offset=offset, out=out, **draws)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        if eng.env_name == 'ant' and not draws:
            # keep stepping in chunks of H until the completed-path sample count reaches batch_size
            while True:
                n_done = self._completed_samples(traj)
                if n_done >= algo.batch_size or traj.T >= 64 * H:
                    break
                T2 = traj.T + H
                traj = eng.rollout(B, T2, H, nne.sam_mode, pool, determ=determ, seed=seed, stream_offset=offset)
        log_std = algo.policy.log_std()
        return DevicePaths(traj, log_std, None)

    @staticmethod
    def _completed_samples(traj):
        done = traj.done.to(torch.int32)
        last = (done * torch.arange(1, traj.T + 1, device=done.device, dtype=torch.int32)[:, None]).max(dim=0).values
        return int(last.sum().item())
