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


class DeviceEvent(object):
    """A timing event that does not disturb what it measures.  torch.cuda.Event records with a system-scope release: the command processor
    writes the whole L2 back before the timestamp, and the NEXT kernel waits 6-15 us for it (five such records per bench iteration were
    2.6 % of it).  This one is created with hipEventReleaseToDevice: the timestamp is taken when the preceding work has finished, no flush.
    Same interface as the part of torch.cuda.Event used here (record / synchronize / elapsed_time in ms)."""
    _hip = None
    RELEASE_TO_DEVICE = 0x40000000

    def __init__(self):
        import ctypes as C
        if DeviceEvent._hip is None:
            h = C.CDLL('libamdhip64.so')
            vp = C.c_void_p                                   # declared signatures: no reliance on ctypes' default int / pointer marshalling
            h.hipEventCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]; h.hipEventCreateWithFlags.restype = C.c_int
            h.hipEventRecord.argtypes = [vp, vp]; h.hipEventRecord.restype = C.c_int
            h.hipEventSynchronize.argtypes = [vp]; h.hipEventSynchronize.restype = C.c_int
            h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]; h.hipEventElapsedTime.restype = C.c_int
            h.hipEventDestroy.argtypes = [vp]; h.hipEventDestroy.restype = C.c_int
            DeviceEvent._hip = h
        self._C = C
        self._h = C.c_void_p()
        rc = DeviceEvent._hip.hipEventCreateWithFlags(C.byref(self._h), C.c_uint(DeviceEvent.RELEASE_TO_DEVICE))
        if rc != 0:
            raise RuntimeError('hipEventCreateWithFlags failed: %d' % rc)

    def record(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = DeviceEvent._hip.hipEventRecord(self._h, self._C.c_void_p(s.cuda_stream))
        if rc != 0:
            raise RuntimeError('hipEventRecord failed: %d' % rc)

    def synchronize(self):
        DeviceEvent._hip.hipEventSynchronize(self._h)

    def elapsed_time(self, other):
        ms = self._C.c_float(0.0)
        rc = DeviceEvent._hip.hipEventElapsedTime(self._C.byref(ms), self._h, other._h)
        if rc != 0:
            raise RuntimeError('hipEventElapsedTime failed: %d (both events must have completed)' % rc)
        return float(ms.value)

    def __del__(self):
        try:
            if self._h:
                DeviceEvent._hip.hipEventDestroy(self._h)
        except Exception:
            pass


def timing_event():
    """DeviceEvent on a GPU, torch's event otherwise (CPU tests never record one)."""
    return DeviceEvent() if torch.cuda.is_available() else torch.cuda.Event(enable_timing=True)


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
                self.e0 = timing_event()
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.owner.enabled and exc[0] is None:
                e1 = timing_event()
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
