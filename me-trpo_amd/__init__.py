"""metrpo_amd -- MI355X-native ME-TRPO policy-optimisation inner loop.

The package directory is `me-trpo_amd/`; import it as `metrpo_amd` (repo-root alias module).
Host code mirrors the reference's operator interface for this one path
(NeuralNetEnv / vec_env_executor / VectorizedSampler / BatchPolopt / NPO / TRPO /
ConjugateGradientOptimizer / LinearFeatureBaseline); all arithmetic runs in libmetrpo.so
(hand-written HIP for gfx950) through the C ABI of include/metrpo.h.  No CPU fallback exists.
"""
from . import _lib                      # raises ImportError if libmetrpo.so is missing
from .engine import Engine, Trajectory, xavier_policy_theta
from .parallel import Comm
from .imagined_env import NeuralNetEnv, VecSimpleEnv, InitStatePool, Box, EnvSpec
from .policy import GaussianMLPPolicy
from .baseline import LinearFeatureBaseline
from .sampler import VectorizedSampler, BaseSampler, DevicePaths
from .optimizer import ConjugateGradientOptimizer
from .algos import BatchPolopt, NPO, TRPO
from . import early_stop
from . import dynamics_training
from .bptt import BPTT
from . import formats
from .params import from_params, shapes_from_params

__all__ = ['Engine', 'Trajectory', 'xavier_policy_theta', 'Comm', 'NeuralNetEnv', 'VecSimpleEnv', 'InitStatePool',
           'Box', 'EnvSpec', 'GaussianMLPPolicy', 'LinearFeatureBaseline', 'VectorizedSampler', 'BaseSampler',
           'DevicePaths', 'ConjugateGradientOptimizer', 'BatchPolopt', 'NPO', 'TRPO', 'early_stop', 'dynamics_training', 'from_params', 'shapes_from_params']
