import os, sys
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
