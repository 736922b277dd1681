"""algos/batch_polopt.py, algos/npo.py, algos/trpo.py of the reference with the same constructor
arguments, attributes and methods (start_worker / obtain_samples / process_samples /
optimize_policy), so that the reference's outer loop (model_based_rl.py:1171-1180) drives them
unchanged:

    algo.start_worker(); paths = algo.obtain_samples(j); samples_data = algo.process_samples(j, paths)
    algo.optimize_policy(j, samples_data)
"""
from .optimizer import ConjugateGradientOptimizer
from .parallel import Comm
from .sampler import VectorizedSampler
from .tracing import PhaseTimers


class BatchPolopt(object):
    def __init__(self, env, policy, baseline, scope=None, n_itr=500, start_itr=0, batch_size=5000,
                 max_path_length=500, discount=0.99, gae_lambda=1, plot=False, pause_for_plot=False,
                 center_adv=True, positive_adv=False, store_paths=False, whole_paths=True, fixed_horizon=False,
                 sampler_cls=None, sampler_args=None, force_batch_sampler=False, comm=None, seed=0, **kwargs):
        self.env, self.policy, self.baseline = env, policy, baseline
        self.scope, self.n_itr, self.start_itr = scope, n_itr, start_itr
        self.batch_size, self.max_path_length = batch_size, max_path_length
        self.discount, self.gae_lambda = discount, gae_lambda
        self.plot, self.pause_for_plot = plot, pause_for_plot
        self.center_adv, self.positive_adv = center_adv, positive_adv
        self.store_paths, self.whole_paths, self.fixed_horizon = store_paths, whole_paths, fixed_horizon
        self.kwargs = kwargs
        self.engine = policy.engine
        self.comm = comm or Comm()
        # Sharded runs: `batch_size` and the sampler's n_envs are PER RANK (the job collects world x batch_size samples; weak
        # scaling).  The stop rule of early-terminating envs (sampler._obtain_until_enough) is applied by every rank to its own
        # envs against this per-rank batch_size -- divide by comm.world beforehand for a fixed global batch.
        self.seed = seed
        self.timers = PhaseTimers()          # rollout / process / policy_opt GPU times (tracing.py); off until .enable()
        assert not force_batch_sampler, "BatchSampler is unreachable on this path (batch_polopt.py:86-90)"
        if sampler_cls is None:
            assert self.policy.vectorized
            sampler_cls = VectorizedSampler
        if sampler_args is None:
            sampler_args = dict()
        self.sampler = sampler_cls(self, **sampler_args)
        self.init_opt()

    def start_worker(self):
        self.sampler.start_worker()

    def shutdown_worker(self):
        self.sampler.shutdown_worker()

    def obtain_samples(self, itr, determ=False, **kw):
        with self.timers.phase('rollout'):
            paths = self.sampler.obtain_samples(itr, determ, **kw)
            # async_line_search: the previous optimize_policy only ENQUEUED its update (the accept test of the first line-search trials
            # runs on the device), so the rollout above went out without the host waiting for it.  Now the update is closed; in the rare
            # case that it was accepted at a later trial than the speculative ones, the policy changed after the rollout was enqueued
            # and the rollout is repeated -- results are those of the synchronous order, always.
            opt = getattr(self, 'optimizer', None)
            if opt is not None and getattr(opt, 'pending', False):
                if opt.finish().get('late'):
                    paths = self.sampler.obtain_samples(itr, determ, **kw)
            return paths

    def process_samples(self, itr, paths):
        with self.timers.phase('process'):
            return self.sampler.process_samples(itr, paths)

    def init_opt(self):
        raise NotImplementedError

    def optimize_policy(self, itr, samples_data):
        raise NotImplementedError


class NPO(BatchPolopt):
    """Natural Policy Optimization (algos/npo.py)."""

    def __init__(self, optimizer=None, optimizer_args=None, step_size=0.01, **kwargs):
        if optimizer is None:
            raise NotImplementedError("PenaltyLbfgsOptimizer (NPO's default, npo.py:24) is outside the hot path; "
                                      "use TRPO or pass an optimizer")
        self.optimizer = optimizer
        self.step_size = step_size
        super(NPO, self).__init__(**kwargs)

    def init_opt(self):
        # npo.py:85-91; the surrogate / mean-KL graph itself (npo.py:68-75) is what the HIP kernels compute
        self.optimizer.update_opt(loss=None, target=self.policy, leq_constraint=(None, self.step_size),
                                  inputs=None, constraint_name="mean_kl")
        return dict()

    def optimize_policy(self, itr, samples_data):
        """npo.py:95-121: inputs = (observations, actions, advantages, agent_infos[mean], agent_infos[log_std])."""
        agent_infos = samples_data["agent_infos"]
        batch = self.engine.make_batch(samples_data["observations"], samples_data["actions"], samples_data["advantages"],
                                       agent_infos["mean"], agent_infos["log_std"], valid=samples_data.get("valids"),
                                       n_global=samples_data.get("n_valid_global"))
        with self.timers.phase('policy_opt'):
            # async_line_search (off by default: the reference decides every trial on the host): see obtain_samples
            self.optimizer.optimize(self.engine, batch, comm=self.comm, defer=bool(getattr(self, 'async_line_search', False)))
        if hasattr(self.sampler, 'finish_baseline_fit') and not getattr(self, 'defer_baseline_fit', False):
            self.sampler.finish_baseline_fit()
        return dict()

    def get_itr_snapshot(self, itr, samples_data):
        return dict(itr=itr, policy=self.policy, baseline=self.baseline, env=self.env)


class TRPO(NPO):
    """Trust Region Policy Optimization (algos/trpo.py)."""

    def __init__(self, optimizer=None, optimizer_args=None, **kwargs):
        if optimizer is None:
            if optimizer_args is None:
                optimizer_args = dict()
            optimizer = ConjugateGradientOptimizer(**optimizer_args)
        super(TRPO, self).__init__(optimizer=optimizer, **kwargs)
