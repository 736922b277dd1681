import os, sys
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get('METRPO_TOL_REPORT'):
        _install_tolerance_report(os.environ['METRPO_TOL_REPORT'])


def _install_tolerance_report(path):
    """METRPO_TOL_REPORT=<file>: record, per assert_allclose call site, the worst observed |a - b| / (atol + rtol |b|) and max |a - b| -- how much of
    every stated tolerance the kernels actually use (the numbers behind DESIGN.md section 5's table).  Test behaviour is unchanged."""
    import atexit, json, traceback
    orig = np.testing.assert_allclose
    seen = {}

    def wrapped(actual, desired, rtol=1e-7, atol=0, *a, **k):
        try:
            x, y = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
            if x.shape == y.shape or x.size == 1 or y.size == 1:
                ok = np.isfinite(x) & np.isfinite(y)
                if ok.any():
                    diff = np.abs(x - y)[ok] if x.shape == y.shape else np.abs(x - y)
                    den = (atol + rtol * np.abs(y))[ok] if np.shape(y) == np.shape(ok) else atol + rtol * np.abs(y)
                    used = float(np.max(diff / np.maximum(den, 1e-300))) if np.all(den > 0) else float('nan')
                    fr = [f for f in traceback.extract_stack() if '/tests/' in f.filename and 'conftest' not in f.filename]
                    site = '%s:%d' % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else '?'
                    rec = seen.setdefault(site, {'rtol': rtol, 'atol': atol, 'used': 0.0, 'max_abs': 0.0, 'calls': 0})
                    rec['used'] = max(rec['used'], used) if used == used else rec['used']
                    rec['max_abs'] = max(rec['max_abs'], float(diff.max())); rec['calls'] += 1
        except Exception:
            pass
        return orig(actual, desired, rtol, atol, *a, **k)

    np.testing.assert_allclose = wrapped
    atexit.register(lambda: json.dump(seen, open(path, 'w'), indent=1, sort_keys=True))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


class ReplayRNG(object):
    """Replays the np.random.{randint,normal} stream recorded by tests/golden/make_golden.py,
    asserting that the consumer asks for the same kind and size at every call."""

    def __init__(self, kinds, offs, flat):
        self.kinds, self.offs, self.flat, self.i = kinds, offs, flat, 0

    def _next(self, kind, shape):
        assert self.i < len(self.kinds), "oracle drew more random numbers than the reference"
        assert self.kinds[self.i] == kind, "draw %d: kind mismatch" % self.i
        v = self.flat[self.offs[self.i]:self.offs[self.i + 1]]
        self.i += 1
        n = int(np.prod(shape)) if shape is not None else 1
        assert v.size == n, "draw %d: size %d != %d" % (self.i - 1, v.size, n)
        return v.reshape(shape) if shape is not None else v[0]

    def randint(self, high, size=None):
        v = self._next(0, size)
        return v.astype(np.int64) if size is not None else int(v)

    def normal(self, size=None):
        return self._next(1, size)

    def exhausted(self):
        return self.i == len(self.kinds)


def dm_from_golden(d, env):
    from oracle import metrpo_oracle as O
    ns, na, _ = O.ENV_SPECS[env]
    Ws, bs, l = [], [], 0
    while 'dynW%d' % l in d:
        Ws.append(d['dynW%d' % l]); bs.append(d['dynb%d' % l]); l += 1
    return O.DynamicsEnsemble(Ws, bs, [str(d['dyn_act'])] * (l - 1), d['in_mean'], d['in_std'],
                              d['diff_mean'], d['diff_std'], int(d['n_drop']), ns, na)


class PoolReset(object):
    def __init__(self, pool):
        self.pool, self.i = pool, 0

    def __call__(self):
        s = self.pool[self.i % len(self.pool)].copy()
        self.i += 1
        return s
