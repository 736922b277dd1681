"""[rllab] LinearFeatureBaseline(reg_coeff=1e-5), constructed at training.py:357 and used by
samplers/base.py:55 (predict) and :164-167 (fit).  Features [o, o^2, t/100, (t/100)^2, (t/100)^3, 1]
with o = clip(obs, -10, 10).  predict is fused into the GAE kernel (coefficients are handed to it);
fit reduces the normal equations on the GPU (metrpo_baseline_gram), all-reduces them across ranks
and solves the (2 ns + 4)^2 system on the host in float64 exactly as rllab does (lstsq, x10 reg on NaN)."""
import numpy as np


class LinearFeatureBaseline(object):
    def __init__(self, env_spec=None, reg_coeff=1e-5):
        self._coeffs = None
        self._reg_coeff = reg_coeff

    @property
    def coeffs(self):
        return self._coeffs

    def get_param_values(self, **tags):
        return self._coeffs

    def set_param_values(self, val, **tags):
        self._coeffs = val

    def solve(self, AtA, Aty):
        """AtA [F,F], Aty [F] float64 (already summed over ranks)."""
        AtA, Aty = np.asarray(AtA, dtype=np.float64), np.asarray(Aty, dtype=np.float64)
        reg = self._reg_coeff
        for _ in range(5):
            self._coeffs = np.linalg.lstsq(AtA + reg * np.identity(AtA.shape[0]), Aty, rcond=None)[0]
            if not np.any(np.isnan(self._coeffs)):
                break
            reg *= 10
        return self._coeffs

    # host-side forms for path dicts (drop-in use with the list-of-paths representation)
    @staticmethod
    def _features(path):
        o = np.clip(path["observations"], -10, 10)
        l = len(path["rewards"])
        al = np.arange(l).reshape(-1, 1) / 100.0
        return np.concatenate([o, o ** 2, al, al ** 2, al ** 3, np.ones((l, 1))], axis=1)

    def predict(self, path):
        if self._coeffs is None:
            return np.zeros(len(path["rewards"]))
        return self._features(path).dot(self._coeffs)
