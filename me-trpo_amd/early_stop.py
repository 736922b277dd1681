"""Early-stopping state machine of the TRPO inner loop: utils.py:285-296 (stop_critereon),
model_based_rl.py:1339-1371 (is_done), :1403-1419 (update_stats) and the loop control of
optimize_policy (:1171-1180, 1209-1299, 1301): snapshot theta, iterate TRPO, every `log_every`
iterations evaluate the K per-model validation costs (build_policy_graph forward, :106-151),
keep/restore the best policy, stop after `num_iters_threshold` iterations without improvement.
Host control flow in NumPy exactly as the reference; the costs come from the validation kernel."""
import numpy as np
import torch


class StopCriterion(object):
    """Callable with the reference's signature `f(loss_old, loss_new, mode='scalar')` (utils.py:285-296).
    scalar: relative increase of the loss above `threshold`; vector: fraction of models that got worse above
    `percent_models_threshold` (ties are not "worse")."""

    def __init__(self, threshold, offset, percent_models_threshold=0.5):
        self.threshold, self.offset, self.percent_models_threshold = threshold, offset, percent_models_threshold

    def __call__(self, loss_old, loss_new, mode='scalar'):
        if mode == 'vector':
            if not isinstance(loss_new, np.ndarray):
                raise AssertionError("vector mode compares per-model cost arrays")
            worse = np.count_nonzero(np.asarray(loss_new) > np.asarray(loss_old))
            return worse / float(np.size(loss_new)) > self.percent_models_threshold
        if mode != 'scalar' or np.ndim(loss_new) != 0:
            raise AssertionError("scalar mode compares two numbers")
        return (loss_new - loss_old) / (abs(loss_old) + self.offset) > self.threshold


stop_critereon = StopCriterion          # the reference's (mis-spelt) factory name, same argument order


def _per_model(costs):
    """True for the K-vector trackers (handled element-wise), False for scalars and length-1 arrays (replaced whole)."""
    return np.ndim(costs) > 0 and np.size(costs) != 1


def is_done(mode, stop_fn, min_validation_costs, candidates):
    """Stop decision of model_based_rl.py:1339-1371 for one validation round: True = the new policy is worse."""
    if mode == 'no_early':
        return False
    if mode in ('real', 'trpo_mean'):
        return bool(min_validation_costs[mode] < candidates[mode])
    if mode == 'one_model':
        return bool(min_validation_costs['estimated'][0] < candidates['estimated'][0])
    if 'estimated' not in mode:
        raise AssertionError("unknown early-stopping mode %r" % (mode,))
    return any(bool(stop_fn(best, candidates[key], mode='vector'))
               for key, best in min_validation_costs.items() if 'estimated' in key)


def update_stats(min_validation_costs, candidates, whole=False):
    """Bookkeeping of model_based_rl.py:1403-1419: with `whole` every tracker takes the candidate; otherwise each entry keeps
    its minimum (per model for the K-vectors, which are updated in place)."""
    for key in list(min_validation_costs):
        best, new = min_validation_costs[key], candidates[key]
        if _per_model(best):
            take = np.ones(np.shape(best), dtype=bool) if whole else np.asarray(best) > np.asarray(new)
            best[take] = np.asarray(new)[take]
        elif whole or best > new:
            min_validation_costs[key] = new


def optimize_policy(algo, validation_init, T, gamma, mode='estimated', whole=True, log_every=5,
                    num_iters_threshold=25, max_iters=400, stop_fn=None, reset_log_std=True, real_cost_fn=None,
                    logger=None):
    """TRPO branch of model_based_rl.py:optimize_policy.  `real_cost_fn()` stands in for
    evaluate_fixed_init_trajectories on the real simulator (out of scope; None -> 0.0)."""
    eng = algo.engine
    stop_fn = stop_fn or stop_critereon(0.10, 1e-5, 0.30)
    if reset_log_std:
        algo.policy.reset_log_std()                                        # kwargs['reset_opt'], :1119-1121
    snapshot = eng.get_policy().clone()                                    # saver.save(policy.ckpt), :1127-1129
    # real_cost_fn must return the same value on every rank (evaluate it on rank 0 and broadcast, or on identical simulators)
    real = (lambda: float(real_cost_fn())) if real_cost_fn else (lambda: 0.0)
    est = lambda: eng.validation_cost(validation_init, T, gamma).cpu().numpy()
    min_costs = {'real': real(), 'trpo_mean': np.inf, 'estimated': est()}  # :1153-1163
    best_index, candidates, history = 0, {}, []
    j = 0
    for j in range(1, max_iters + 1):
        algo.start_worker()                                                # :1175
        paths = algo.obtain_samples(j)
        samples_data = algo.process_samples(j, paths)
        algo.optimize_policy(j, samples_data)
        if j % log_every == 0:                                             # :1209
            if mode == 'trpo_mean':                                        # :1221-1229
                determ = algo.obtain_samples(j, determ=True)
                tr = determ.traj
                comp = tr.done.flip(0).cummax(0).values.flip(0).bool()
                # every rank must take the same stop decision (else they issue different numbers of all-reduces): the mean
                # return is formed from the GLOBAL (sum of returns, number of paths) pair
                pair = torch.stack([(tr.rew * comp).sum().double(), tr.done.sum().double()])
                algo.comm.allreduce_sum_(pair)
                pair = pair.cpu()
                candidates['trpo_mean'] = float(-pair[0].item() / max(1.0, pair[1].item()))
            else:
                candidates['trpo_mean'] = 0.0
            candidates['estimated'] = est()                                # :1239-1241
            candidates['real'] = real()
            history.append((j, float(np.mean(candidates['estimated']))))
            if logger:
                logger('iter %d est=%s' % (j, np.array_str(candidates['estimated'], precision=3)))
            if not is_done(mode, stop_fn, min_costs, candidates):          # :1286-1295
                best_index = j
                snapshot = eng.get_policy().clone()
                update_stats(min_costs, candidates, whole)
            if j - best_index >= num_iters_threshold:                      # :1298
                break
    eng.set_policy(snapshot)                                               # log_and_restore, :1301/:1400
    return dict(best_index=best_index, last_index=j, min_validation_costs=min_costs, history=history)
