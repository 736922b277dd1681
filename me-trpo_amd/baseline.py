"""[rllab] LinearFeatureBaseline(reg_coeff=1e-5), constructed at training.py:357 and used by
samplers/base.py:55 (predict) and :164-167 (fit).  Features [o, o^2, t/100, (t/100)^2, (t/100)^3, 1]
with o = clip(obs, -10, 10).  predict is fused into the GAE kernel (coefficients are handed to it);
fit reduces the normal equations on the GPU (metrpo_baseline_gram), all-reduces them across ranks
and solves the (2 ns + 4)^2 system in float64 as rllab does (x10 reg on NaN) -- on the host with lstsq (solve), or on the device
(solve_device: float64 elimination of the symmetric positive definite system, metrpo_baseline_solve) so that the coefficients go from the fit into the next GAE kernel
without a host round trip; `coeffs` then copies them to the host only when somebody asks."""
import numpy as np


class LinearFeatureBaseline(object):
    def __init__(self, env_spec=None, reg_coeff=1e-5):
        self._coeffs = None
        self._coeffs_dev = None              # device tensor [F] f64 of the last solve_device (None: the host array is the truth)
        self._reg_coeff = reg_coeff

    @property
    def coeffs(self):
        if self._coeffs_dev is not None and self._coeffs is None:
            self._coeffs = self._coeffs_dev.cpu().numpy()          # synchronises; only diagnostics / the path-dict API get here
        return self._coeffs

    @property
    def coeffs_for_kernel(self):
        """What metrpo_gae should be given: the device tensor when the last fit stayed on the device, else the host array (or None)."""
        return self._coeffs_dev if self._coeffs_dev is not None else self._coeffs

    def get_param_values(self, **tags):
        return self.coeffs

    def set_param_values(self, val, **tags):
        self._coeffs = val
        self._coeffs_dev = None

    def solve_device(self, engine, gram):
        """gram = [AtA | Aty] float64 device tensor, already summed over ranks; stream-ordered."""
        self._coeffs_dev = engine.baseline_solve(gram, self._reg_coeff, out=self._coeffs_dev)
        self._coeffs = None
        return self._coeffs_dev

    def solve(self, AtA, Aty):
        """AtA [F,F], Aty [F] float64 (already summed over ranks)."""
        AtA, Aty = np.asarray(AtA, dtype=np.float64), np.asarray(Aty, dtype=np.float64)
        reg = self._reg_coeff
        for _ in range(5):
            self._coeffs = np.linalg.lstsq(AtA + reg * np.identity(AtA.shape[0]), Aty, rcond=None)[0]
            if not np.any(np.isnan(self._coeffs)):
                break
            reg *= 10
        self._coeffs_dev = None
        return self._coeffs

    # host-side forms for path dicts (drop-in use with the list-of-paths representation)
    @staticmethod
    def _features(path):
        o = np.clip(path["observations"], -10, 10)
        l = len(path["rewards"])
        al = np.arange(l).reshape(-1, 1) / 100.0
        return np.concatenate([o, o ** 2, al, al ** 2, al ** 3, np.ones((l, 1))], axis=1)

    def predict(self, path):
        if self.coeffs is None:
            return np.zeros(len(path["rewards"]))
        return self._features(path).dot(self.coeffs)
