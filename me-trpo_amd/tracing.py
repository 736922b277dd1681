"""Phase timers of the inner loop -- the counterpart of the reference's wall-clock accounting: `policy_time / env_time /
process_time` of VectorizedSampler.obtain_samples (samplers/vectorized_sampler.py:54-56,67,70,106; logged as PolicyExecTime /
EnvExecTime / ProcessExecTime) and `policy_opt_time` (model_based_rl.py:694).

The reference times host code with time.time(); here the phases are GPU work enqueued asynchronously, so each phase is bracketed
by a pair of HIP events on the engine's stream and the elapsed times are read later, without adding a synchronisation to the loop:

    algo.timers.enable()
    ... iterations ...
    algo.timers.summary()    -> {'rollout_ms': ..., 'process_ms': ..., 'policy_opt_ms': ..., 'n': iterations measured}

`rollout_ms` = policy + env execution (one fused kernel here: PolicyExecTime + EnvExecTime), `process_ms` = process_samples
(baseline predict, GAE, centring, normal equations), `policy_opt_ms` = optimize_policy.  Inside libmetrpo.so the same three phases
are wrapped in roctx ranges (csrc/trace.h: metrpo_rollout / metrpo_gae ... / metrpo_trpo_update), visible in
`rocprofv3 --marker-trace`."""
import torch


class PhaseTimers(object):
    PHASES = ('rollout', 'process', 'policy_opt')

    def __init__(self, keep=256):
        self.enabled = False
        self.keep = int(keep)
        self._pairs = {p: [] for p in self.PHASES}

    def enable(self, on=True):
        self.enabled = bool(on)
        return self

    def reset(self):
        for v in self._pairs.values():
            del v[:]

    class _Range(object):
        def __init__(self, owner, phase):
            self.owner, self.phase = owner, phase

        def __enter__(self):
            if self.owner.enabled:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.owner.enabled and exc[0] is None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                lst = self.owner._pairs[self.phase]
                lst.append((self.e0, e1))
                if len(lst) > self.owner.keep:
                    del lst[0]
            return False

    def phase(self, name):
        """Context manager around the enqueue of one phase (no-op while disabled)."""
        return self._Range(self, name)

    def times_ms(self, name):
        """Elapsed GPU time of every recorded occurrence of the phase (synchronises on the last event)."""
        pairs = self._pairs[name]
        if not pairs:
            return []
        pairs[-1][1].synchronize()
        return [a.elapsed_time(b) for a, b in pairs]

    def summary(self):
        out = {}
        n = 0
        for p in self.PHASES:
            ts = self.times_ms(p)
            out[p + '_ms'] = float(sum(ts) / len(ts)) if ts else None
            n = max(n, len(ts))
        out['n'] = n
        return out
