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
        self.acc = None                      # zeroed float64 accumulators for process_samples (Engine.process_begin), consumed once

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
        coeffs = getattr(algo.baseline, 'coeffs_for_kernel', None) if hasattr(algo.baseline, 'coeffs_for_kernel') else algo.baseline.coeffs
        F = 2 * eng.ns + 4
        acc = getattr(paths, 'acc', None)                       # zeroed by obtain_samples' process_begin launch; used once
        if acc is None or acc.numel() != 3 + F * F + F:
            acc = torch.zeros(3 + F * F + F, dtype=torch.float64, device=eng.device)     # one fill: [advantage statistics | AtA | Aty]
        paths.acc = None
        adv, ret, valid, stats = eng.gae(tr, coeffs, algo.discount, algo.gae_lambda, stats=acc[:3])
        comm.allreduce_sum_(stats)
        # fixed-horizon envs: every sample is valid and the global count is known without a host sync
        static = (eng.env_name != 'ant') and (tr.T % tr.H == 0)
        n_global = comm.world * tr.T * tr.B if static and self._uniform_B(comm, tr.B) else int(stats[2].item())
        if algo.center_adv:
            eng.center_advantages(adv, valid, stats)
        if algo.positive_adv:                                   # [rllab] util.shift_advantages_to_positive, min over ALL ranks' valid samples
            lo = torch.where(valid.bool(), adv, torch.full_like(adv, float('inf'))).min().reshape(1)
            comm.allreduce_min_(lo)
            adv.sub_(lo).add_(1e-8)
        gram = acc[3:]
        eng.baseline_gram(tr.obs, ret, tr.tpath, valid, out=gram)
        comm.allreduce_sum_(gram)
        if getattr(algo, 'device_baseline_fit', False) and gram.is_cuda and hasattr(algo.baseline, 'solve_device'):
            # the fit's solve on the device (metrpo_baseline_solve): the coefficients never visit the host, nothing to wait for, no copy
            # between the rollout and the next GAE kernel (the host solve + its two transfers were where the host fell behind the GPU)
            algo.baseline.solve_device(eng, gram)
            self._pending_gram = None
            self._gram_event = None
        else:
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
        Fixed-horizon envs: ceil(batch/(n_envs*H)) rounds of H steps in ONE launch.  Early-terminating envs (Ant): the
        reference tests the condition after EVERY time step, see _obtain_until_enough."""
        algo, eng = self.algo, self.algo.engine
        comm = getattr(algo, 'comm', None) or Comm()
        B, H = self._n_envs, algo.max_path_length
        nne = algo.env
        pool = nne.env.device_tensor(eng.device) if hasattr(nne.env, 'device_tensor') else \
            torch.as_tensor(np.stack([nne.env.reset() for _ in range(max(B, 1))]), dtype=torch.float32, device=eng.device)
        # Philox key of this launch: a running per-sampler counter, NOT `itr` -- the outer ME-TRPO loop restarts itr at 1 every
        # sweep (model_based_rl.py:1171) and must not replay the same noise (the reference draws from one continuing np.random)
        seed = (getattr(algo, 'seed', 0) * 1000003 + self._itr_seed * 7919 + (1 if determ else 0)) & 0xFFFFFFFFFFFFFFFF
        self._itr_seed += 1
        offset = comm.rank * B
        draws = draws or {}
        ev = getattr(algo, 'rollout_events', None)          # optional [(start, end)] HIP-event pairs around the launch
        if ev is not None:
            from .tracing import timing_event
            e0, e1 = timing_event(), timing_event()
            e0.record()
        common = dict(determ=determ, eval_all_heads=getattr(algo, 'eval_all_heads', True), seed=seed, stream_offset=offset)
        if eng.env_name == 'ant':
            traj = self._obtain_until_enough(B, H, nne.sam_mode, pool, draws, common)
        else:
            rounds = max(1, -(-algo.batch_size // (B * H)))
            T = rounds * H
            out = None
            if getattr(algo, 'reuse_trajectory_buffers', False):      # opt-in: later rollouts overwrite earlier DevicePaths
                key = (B, T, H)
                if getattr(self, '_traj_key', None) != key:
                    self._traj_buf, self._traj_key = eng.alloc_trajectory(B, T, H), key
                out = self._traj_buf
            traj = eng.rollout(B, T, H, nne.sam_mode, pool, out=out, **common, **draws)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        if hasattr(eng, 'process_begin'):
            # one launch: the batch's agent_infos['log_std'] (what policy.log_std() returns) + the zeroed accumulators process_samples adds into
            F = 2 * eng.ns + 4
            log_std, acc = eng.process_begin(3 + F * F + F)
        else:
            log_std, acc = algo.policy.log_std(), None
        paths = DevicePaths(traj, log_std, None)
        paths.acc = acc
        return paths

    def _obtain_until_enough(self, B, H, sam_mode, pool, draws, common):
        """Early-terminating envs.  The reference steps all envs once, adds the lengths of the paths that just completed to
        n_samples and stops as soon as n_samples >= batch_size (vectorized_sampler.py:60,104); paths still open are dropped.
        Completed samples after t steps lie in (B*(t-H), B*t], so the stop step t* is in [ceil(batch/B), ceil(batch/B)+H):
        the first ceil(batch/B) steps are one launch, then chunks CONTINUE from the previous chunk's end state (same Philox
        streams as one long launch: counters use the global step).  metrpo_sampler_progress evaluates the loop condition on
        the device after every chunk and raises a flag that turns the chunks already enqueued behind it into no-ops, so
        there is no host round trip inside the loop (large-net rollouts, where a chunk is tens of ms of GEMMs, poll the flag
        instead of enqueuing dead GEMMs).  One read-back of t* at the end; the returned tensors are the [0, t*] prefix views.
        In sharded runs each rank applies the rule to its own envs and its own batch_size share (no collective)."""
        algo, eng = self.algo, self.algo.engine
        dev = eng.device
        batch = int(algo.batch_size)
        T_first = max(1, -(-batch // B))
        T_max = T_first + H
        chunk = max(1, int(getattr(algo, 'sampler_chunk', max(8, H // 8))))
        key = (B, T_max, H, chunk)                     # chunk sizes the per-step count buffer below
        if getattr(self, '_ant_key', None) != key or not getattr(algo, 'reuse_trajectory_buffers', False):
            self._ant_buf, self._ant_key = eng.alloc_trajectory(B, T_max, H), key
            self._ant_state = (torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
            self._ant_counts = torch.empty(max(T_first, chunk, T_max - T_first), dtype=torch.float64, device=dev)
        buf, (last_ts, last_model) = self._ant_buf, self._ant_state
        state = torch.tensor([0.0, -1.0], dtype=torch.float64, device=dev)
        stop = torch.zeros(1, dtype=torch.int32, device=dev)
        poll = eng.rollout_path() == 3

        def view(t_lo, t_hi):
            from .engine import Trajectory
            return Trajectory(buf.obs[t_lo:t_hi], buf.act[t_lo:t_hi], buf.rew[t_lo:t_hi], buf.mean[t_lo:t_hi], buf.done[t_lo:t_hi],
                              buf.tpath[t_lo:t_hi], buf.last_obs, B, t_hi - t_lo, H)

        def chunk_draws(t_lo, t_hi):
            d = {}
            for k, v in draws.items():
                if v is not None:
                    d[k] = v[t_lo:t_hi + 1] if k in ('reset_idx', 'reset_model') else v[t_lo:t_hi]
            return d

        t = 0
        one_launch = False                                         # the kernel family of the first chunk runs all steps of a call in one launch
        while t < T_max:
            t_hi = min(T_max, t + (T_first if t == 0 else (T_max - t if one_launch else chunk)))
            out = view(t, t_hi)
            # one-launch families apply the stop rule INSIDE the call (metrpo_rollout_args::stop_batch): everything behind the first T_first steps is one
            # call that stops stepping at the stop step -- no chunk granularity, no dead steps
            extra = dict(stop_batch=batch, stop_cum=state[0:1]) if one_launch else {}
            eng.rollout(B, t_hi - t, H, sam_mode, pool, out=out, t0=t, last_state=(last_ts, last_model), stop=stop,
                        resume=None if t == 0 else (buf.last_obs, last_ts, last_model), **extra, **common, **chunk_draws(t, t_hi))
            eng.sampler_progress(out.done, out.tpath, t, batch, self._ant_counts, state, stop)
            t = t_hi
            if poll and t < T_max:
                if eng.last_rollout_kernel() == 'streamk-persistent':
                    poll, one_launch = False, True                 # (nothing to wait for: the rest is one call)
                elif int(stop.item()):
                    break
        t_stop = int(state[1].item())
        if t_stop < 0:
            raise RuntimeError("obtain_samples: %d completed samples < batch_size %d after %d steps (draws too short?)"
                               % (int(state[0].item()), batch, t))
        return view(0, t_stop + 1)
