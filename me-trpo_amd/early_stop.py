"""Early-stopping state machine of the TRPO inner loop: utils.py:285-296 (stop_critereon),
model_based_rl.py:1339-1371 (is_done), :1403-1419 (update_stats) and the loop control of
optimize_policy (:1171-1180, 1209-1299, 1301): snapshot theta, iterate TRPO, every `log_every`
iterations evaluate the K per-model validation costs (build_policy_graph forward, :106-151),
keep/restore the best policy, stop after `num_iters_threshold` iterations without improvement.
Host control flow in NumPy exactly as the reference; the costs come from the validation kernel."""
import numpy as np


def stop_critereon(threshold, offset, percent_models_threshold=0.5):
    def f(loss_old, loss_new, mode='scalar'):
        if mode == 'scalar':
            assert not hasattr(loss_new, '__iter__')
            return (loss_new - loss_old) / (np.abs(loss_old) + offset) > threshold
        else:
            assert mode == 'vector'
            assert isinstance(loss_new, np.ndarray)
            out = loss_new > loss_old
            return np.mean(out) > percent_models_threshold
    return f


def is_done(mode, stop_fn, min_validation_costs, candidates):
    if mode == 'real':
        return min_validation_costs['real'] < candidates['real']
    elif mode == 'trpo_mean':
        assert 'trpo_mean' in min_validation_costs.keys()
        return min_validation_costs['trpo_mean'] < candidates['trpo_mean']
    elif mode == 'one_model':
        return min_validation_costs['estimated'][0] < candidates['estimated'][0]
    elif mode == 'no_early':
        return False
    else:
        assert 'estimated' in mode
        for _mode in min_validation_costs.keys():
            if 'estimated' in _mode and stop_fn(min_validation_costs[_mode], candidates[_mode], mode='vector'):
                return True
        return False


def update_stats(min_validation_costs, candidates, whole=False):
    for _mode in min_validation_costs.keys():
        costs = min_validation_costs[_mode]
        if hasattr(costs, '__iter__') and len(costs) != 1:
            if whole:
                min_validation_costs[_mode][:] = candidates[_mode][:]
            else:
                to_update = costs > candidates[_mode]
                min_validation_costs[_mode][to_update] = candidates[_mode][to_update]
        elif whole or costs > candidates[_mode]:
            min_validation_costs[_mode] = candidates[_mode]


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
                candidates['trpo_mean'] = float(-(tr.rew * comp).sum().item() / max(1, int(tr.done.sum().item())))
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
