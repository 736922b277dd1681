"""CPU restatement (NumPy, float64) of the BPTT policy update -- SURVEY.md section 8f rank 3 ("next" row):
gradient of  mean_k policy_cost_k  through the unrolled imagined rollout, and the clipped Adam step on it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: build_policy_graph (model_based_rl.py:106-151: x <- init; T times u = clip(policy(x)), x' = model_i(x,u),
cost += gamma^t * cost_tf(x,u,x') [Ant: masked by the running `dones`, updated after the cost]),
training_policy_cost = reduce_mean over the K models (:365), get_policy_optimizer (:186-206: Adam with per-variable
tf.clip_by_norm, utils.py:262-276), the 'bptt' branch of optimize_policy (:1181-1187).

PINNED (round 2): the reference's build_policy_graph + get_policy_optimizer run unmodified on the eager tf stand-in
(tests/golden/make_golden_tf.py); autograd through the reference's OWN unrolled graph gives policy_grads_and_vars, which the
hand-derived adjoint recursion below reproduces to 1e-9 relative, as do three clipped Adam steps (tests/test_oracle_tfgraph.py;
the Adam rule itself is the stand-in's restatement of the TF 1.4 documentation).  tests/test_oracle_bptt.py additionally checks the
recursion against torch autograd of an independently written unrolled computation and against finite differences.
"""
import numpy as np
from . import metrpo_oracle as O


def _cost_grads(env, u, xn, w):
    """d(sum_b w_b * cost_b)/d(u), /d(x_next) for the per-sample costs of metrpo_oracle.cost_np_vec (= cost_tf per sample).
    w: (B,) weights gamma^t * (1 - dones) / (B * K)."""
    gu = np.zeros_like(u); gx = np.zeros_like(xn)
    na = u.shape[1]
    if env == 'swimmer':
        gx[:, 5] = -w; gu = (w * 1e-2 * 2.0 / na)[:, None] * u
    elif env == 'half_cheetah':
        inner = xn[:, 9] - 1e-1 * 0.5 * np.sum(np.square(u), axis=1)
        p = ((inner >= -10) & (inner <= 10)).astype(u.dtype)            # tf.clip_by_value passes the gradient inside [min, max]
        gx[:, 9] = -w * p; gu = (w * p * 1e-1)[:, None] * u
    elif env == 'ant':
        gx[:, 15] = -w; gu = (w * 1e-2)[:, None] * u
    elif env == 'humanoid':
        gx[:, -1] = w * 2.0 * (xn[:, -1] - 1.5); gu = (w * 2e-5)[:, None] * u
    elif env == 'hopper':
        gx[:, 5] += -w; gu = (w * 0.01)[:, None] * u
        gx[:, 0] += w * 10.0 * (-(0.45 - xn[:, 0] > 0).astype(u.dtype))
        gx[:, 1] += w * 10.0 * np.sign(xn[:, 1]) * (np.abs(xn[:, 1]) - 0.2 > 0)
        gx[:, 2:] += w[:, None] * np.sign(xn[:, 2:]) * (np.abs(xn[:, 2:]) - 100 > 0)
    elif env == 'snake':
        gx[:, 7] = -w; gu = (w * 1e-2)[:, None] * u
    else:
        raise KeyError(env)
    return gu, gx


def _dyn_forward_cache(dm, k, s, a):
    xu = np.concatenate([s, a], axis=1)
    h = ((xu - dm.in_mean) / dm.in_std)[:, dm.n_drop:]
    hs = [h]
    L = len(dm.Ws)
    for l in range(L):
        h = h @ dm.Ws[l][k] + dm.bs[l][k]
        if l < L - 1:
            assert dm.acts[l] == 'relu'
            h = np.maximum(h, 0); hs.append(h)
    return dm.diff_mean[:dm.ns] + dm.diff_std[:dm.ns] * h + s, hs


def _dyn_vjp(dm, k, hs, g_next):
    """(d x_next / d [s, a])^T g_next for head k."""
    ns, na = dm.ns, dm.na
    d = g_next * dm.diff_std[:ns]
    L = len(dm.Ws)
    for l in range(L - 1, -1, -1):
        d = d @ dm.Ws[l][k].T
        if l > 0:
            d = d * (hs[l] > 0)
    gxu = np.zeros((g_next.shape[0], ns + na), dtype=g_next.dtype)
    gxu[:, dm.n_drop:] = d
    gxu = gxu / dm.in_std
    return g_next + gxu[:, :ns], gxu[:, ns:]


def policy_costs_and_grad(dm, theta, dims, env, x0, T, gamma):
    """Returns (costs [K], grad [P]): costs[i] = policy_cost of model i; grad = d mean_i costs[i] / d theta
    (log_std entries 0: `stochastic` is 0 in the 'bptt' branch)."""
    theta = np.asarray(theta, dtype=np.float64)
    Ws, bs, log_std = O.policy_unflatten(theta, dims)
    K, B = dm.K, x0.shape[0]
    costs = np.zeros(K)
    gWs = [np.zeros_like(W) for W in Ws]; gbs = [np.zeros_like(b) for b in bs]
    for k in range(K):
        xs, us, mus, dyn_hs, pol_hs, ws = [x0.astype(np.float64)], [], [], [], [], []
        dones = np.zeros(B)
        for t in range(T):
            x = xs[-1]
            mu, hs_p = O._policy_forward(Ws, bs, x)
            u = np.clip(mu, -1.0, 1.0)
            xn, hs_d = _dyn_forward_cache(dm, k, x, u)
            c = O.cost_np_vec(env, x, u, xn)
            w = (gamma ** t) * (1.0 - dones) / B if env == 'ant' else np.full(B, (gamma ** t) / B)
            costs[k] += np.sum(w * c)
            if env == 'ant':
                dones = np.maximum(dones, O.is_done(env, x, xn).astype(np.float64))
            xs.append(xn); us.append(u); mus.append(mu); dyn_hs.append(hs_d); pol_hs.append(hs_p); ws.append(w / K)
        lam = np.zeros_like(x0, dtype=np.float64)                # d J / d x_T
        for t in range(T - 1, -1, -1):
            gu_c, gx_c = _cost_grads(env, us[t], xs[t + 1], ws[t])
            gs, gu_d = _dyn_vjp(dm, k, dyn_hs[t], lam + gx_c)
            gmu = (gu_c + gu_d) * ((mus[t] >= -1.0) & (mus[t] <= 1.0))
            gW, gb = O._backward(Ws, pol_hs[t], gmu)
            for l in range(len(Ws)):
                gWs[l] += gW[l]; gbs[l] += gb[l]
            # input adjoint of the policy
            dh = gmu @ Ws[-1].T
            for l in range(len(Ws) - 2, -1, -1):
                dh = (dh * (1.0 - np.square(pol_hs[t][l + 1]))) @ Ws[l].T
            lam = gs + dh
    return costs, O.policy_flatten(gWs, gbs, np.zeros_like(log_std))


def clip_by_norm(g, clip_val):
    """tf.clip_by_norm: g * clip / max(||g||_2, clip)."""
    n = np.sqrt(np.sum(np.square(g)))
    return g * clip_val / max(n, clip_val)


class PolicyAdam(object):
    """tf.train.AdamOptimizer(learning_rate) on the policy variables (W_l, b_l, log_std), defaults beta1 0.9, beta2 0.999, eps 1e-8."""

    def __init__(self, P):
        self.t, self.m, self.v = 0, np.zeros(P), np.zeros(P)

    def step(self, theta, grad, dims, lr, clip_val=None, b1=0.9, b2=0.999, eps=1e-8):
        g = np.array(grad, dtype=np.float64)
        if clip_val is not None:                                   # per VARIABLE (utils.py:268-271)
            o = 0
            for i in range(len(dims) - 1):
                for n in (dims[i] * dims[i + 1], dims[i + 1]):
                    g[o:o + n] = clip_by_norm(g[o:o + n], clip_val); o += n
            g[o:] = clip_by_norm(g[o:], clip_val)
        self.t += 1
        lr_t = lr * np.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        self.m = b1 * self.m + (1 - b1) * g
        self.v = b2 * self.v + (1 - b2) * g * g
        return theta - lr_t * self.m / (np.sqrt(self.v) + eps)
