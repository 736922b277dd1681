"""[rllab] sandbox.rocky.tf.optimizers.conjugate_gradient_optimizer.ConjugateGradientOptimizer with
PerlmutterHvp and rllab.misc.krylov.cg, as wired by algos/trpo.py:18-20 (all defaults) and driven by
algos/npo.py:85-91 (update_opt) and :111 (optimize).

Two execution forms over the same kernels:
  fused=True  (default)  one C call (metrpo_trpo_update): CG vectors stay on the GPU in float64, the only
                         host synchronisation is the accept test of each line-search trial.
  fused=False            the reference-shaped host loop (NumPy float64 vectors, one kernel call per
                         f_loss / f_grad / f_Hx / f_loss_constraint evaluation).  Same arithmetic; used for
                         debugging and by the multi-process CPU tests with an injected evaluator.
Sharded runs: every evaluation returns this rank's pre-scaled share; a sum all-reduce (Comm) makes it the
global mean, so every rank walks the identical CG / line-search trajectory."""
import numpy as np
import torch

from .parallel import Comm


def cg(f_Ax, b, cg_iters=10, residual_tol=1e-10):
    """rllab.misc.krylov.cg"""
    p, r, x = b.copy(), b.copy(), np.zeros_like(b)
    rdotr = r.dot(r)
    for _ in range(cg_iters):
        z = f_Ax(p)
        v = rdotr / p.dot(z)
        x += v * p
        r -= v * z
        newrdotr = r.dot(r)
        mu = newrdotr / rdotr
        p = r + mu * p
        rdotr = newrdotr
        if rdotr < residual_tol:
            break
    return x


class EngineEvaluator(object):
    """f_loss/f_grad/f_Hx_plain/f_loss_constraint of the compiled graph, served by the HIP kernels."""

    def __init__(self, engine, batch):
        self.engine, self.batch = engine, batch

    def loss_grad(self):
        return self.engine.loss_grad(self.batch)                  # tensor [1+P] f64 (this rank's share)

    def hvp(self, v):
        return self.engine.fvp(self.batch, v)                     # tensor [P] f64

    def loss_constraint(self, theta):
        return self.engine.loss_kl(self.batch, theta)             # tensor [2] f64

    def get_params(self):
        return self.engine.get_policy().double().cpu().numpy()

    def set_params(self, theta):
        self.engine.set_policy(np.asarray(theta, dtype=np.float32))


class ConjugateGradientOptimizer(object):
    def __init__(self, cg_iters=10, reg_coeff=1e-5, subsample_factor=1.0, backtrack_ratio=0.8, max_backtracks=15,
                 accept_violation=False, hvp_approach=None, num_slices=1, fused=True):
        assert subsample_factor == 1.0, "subsampled Hx is not used on this path (algos/trpo.py:18-20 passes no args)"
        self._cg_iters, self._reg_coeff = cg_iters, reg_coeff
        self._backtrack_ratio, self._max_backtracks = backtrack_ratio, max_backtracks
        self._accept_violation, self._fused = accept_violation, fused
        self._max_constraint_val = None
        self._constraint_name = None
        self._last_diag = None
        self._open = None                    # engine whose update was only enqueued (optimize(..., defer=True)); finish() closes it
        self.spec_trials = 2                 # line-search trials decided on the device by a deferred update (the first two cover ~95 % of C1's updates)

    @property
    def pending(self):
        return self._open is not None

    @property
    def last_diag(self):
        """Diagnostics of the last optimize(); closes a deferred update first."""
        if self._open is not None:
            self.finish()
        return self._last_diag

    @last_diag.setter
    def last_diag(self, d):
        self._last_diag = d

    def finish(self):
        """Second half of optimize(..., defer=True): waits for the update, stores and returns its diagnostics.  diag['late'] is True
        when the policy changed only now (accepted at a later trial than the speculative ones): launches enqueued since optimize()
        used the previous policy."""
        if self._open is None:
            return self._last_diag
        eng, self._open = self._open, None
        self._last_diag = eng.trpo_update_end()
        return self._last_diag

    def update_opt(self, loss=None, target=None, leq_constraint=None, inputs=None, extra_inputs=None,
                   constraint_name="constraint", *args, **kwargs):
        """algos/npo.py:85-91: leq_constraint = (mean_kl, step_size); the symbolic loss/inputs of the
        reference have no counterpart here (the kernels implement that exact graph)."""
        constraint_term, constraint_value = leq_constraint
        self._max_constraint_val = float(constraint_value)
        self._constraint_name = constraint_name
        self._target = target

    # -- diagnostics the reference exposes (commented out at npo.py:106-120) ---------------------
    def loss(self, evaluator, comm=None):
        out = self._reduce(evaluator.loss_constraint(None), comm)
        return float(out[0])

    def constraint_val(self, evaluator, comm=None):
        out = self._reduce(evaluator.loss_constraint(None), comm)
        return float(out[1])

    @staticmethod
    def _reduce(t, comm):
        comm = comm or Comm()
        t = comm.allreduce_sum_(t)
        return t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    def optimize(self, engine_or_evaluator, batch=None, comm=None, defer=False):
        """defer=True (fused form only): enqueue the update with its first `spec_trials` line-search trials decided on the device and
        return at once (None); finish() / last_diag complete it.  The caller may enqueue the next rollout in between (algos.BatchPolopt)."""
        comm = comm or Comm()
        if self._open is not None:
            self.finish()
        if self._fused and batch is not None:
            # ranks > 1: the ctx's own RCCL communicator when one is attached (all-reduces issued from C), else a host callback
            need = comm.world > 1 or comm.always_reduce
            ar = (lambda t: comm.allreduce_sum_(t)) if (need and not getattr(engine_or_evaluator, 'comm_world', 0)) else None
            if defer and ar is None:
                engine_or_evaluator.trpo_update(
                    batch, max_kl=self._max_constraint_val, cg_iters=self._cg_iters, reg_coeff=self._reg_coeff,
                    backtrack_ratio=self._backtrack_ratio, max_backtracks=self._max_backtracks,
                    accept_violation=self._accept_violation, spec_trials=self.spec_trials)
                self._open = engine_or_evaluator
                return None
            self.last_diag = engine_or_evaluator.trpo_update(
                batch, max_kl=self._max_constraint_val, cg_iters=self._cg_iters, reg_coeff=self._reg_coeff,
                backtrack_ratio=self._backtrack_ratio, max_backtracks=self._max_backtracks,
                accept_violation=self._accept_violation, allreduce=ar)
            return self.last_diag
        ev = engine_or_evaluator if batch is None else EngineEvaluator(engine_or_evaluator, batch)
        self.last_diag = self._optimize_host(ev, comm)
        return self.last_diag

    def _optimize_host(self, ev, comm):
        """Line-by-line restatement of ConjugateGradientOptimizer.optimize."""
        prev_param = np.copy(ev.get_params())
        lg = self._reduce(ev.loss_grad(), comm)
        loss_before, flat_g = float(lg[0]), np.array(lg[1:], dtype=np.float64)

        def Hx(x):
            return self._reduce(ev.hvp(x), comm) + self._reg_coeff * x

        descent_direction = cg(Hx, flat_g, cg_iters=self._cg_iters)
        initial_step_size = np.sqrt(2.0 * self._max_constraint_val * (1. / (descent_direction.dot(Hx(descent_direction)) + 1e-8)))
        if np.isnan(initial_step_size):
            initial_step_size = 1.
        flat_descent_step = initial_step_size * descent_direction
        n_iter, loss, constraint_val = 0, np.nan, np.nan
        cur_param = prev_param
        for n_iter, ratio in enumerate(self._backtrack_ratio ** np.arange(self._max_backtracks)):
            cur_step = ratio * flat_descent_step
            cur_param = (prev_param - cur_step).astype(np.float32)
            lk = self._reduce(ev.loss_constraint(cur_param), comm)
            loss, constraint_val = float(lk[0]), float(lk[1])
            if loss < loss_before and constraint_val <= self._max_constraint_val:
                break
        accepted = True
        if (np.isnan(loss) or np.isnan(constraint_val) or loss >= loss_before or
                constraint_val >= self._max_constraint_val) and not self._accept_violation:
            accepted = False                       # "Line search condition violated. Rejecting the step!"
            ev.set_params(prev_param)
        else:
            ev.set_params(cur_param)
        return dict(loss_before=loss_before, loss=loss, kl=constraint_val, beta=float(initial_step_size),
                    n_backtrack=int(n_iter), accepted=accepted, g=flat_g, d=descent_direction)
