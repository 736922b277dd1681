"""BPTT restatement (oracle/bptt_oracle.py) against torch autograd of the same unrolled graph (an independent derivation:
TF1's tf.gradients on build_policy_graph cannot run here) -- SURVEY.md 8f rank 3."""
import numpy as np
import pytest
import torch
from oracle import metrpo_oracle as O
from oracle import bptt_oracle as Bp


def torch_costs_and_grad(dm, theta, dims, env, x0, T, gamma):
    """model_based_rl.py:106-151 literally, in torch float64; gradient by autograd."""
    th = torch.tensor(theta, dtype=torch.float64, requires_grad=True)
    Ws, bs, o = [], [], 0
    for i in range(len(dims) - 1):
        n = dims[i] * dims[i + 1]
        Ws.append(th[o:o + n].reshape(dims[i], dims[i + 1])); o += n
        bs.append(th[o:o + dims[i + 1]]); o += dims[i + 1]
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    in_mean, in_std, dmean, dstd = t64(dm.in_mean), t64(dm.in_std), t64(dm.diff_mean), t64(dm.diff_std)
    costs = []
    for k in range(dm.K):
        x = t64(x0); cost = 0.0; dones = torch.zeros(x0.shape[0], dtype=torch.float64)
        for t in range(T):
            h = x
            for l in range(len(Ws) - 1):
                h = torch.tanh(h @ Ws[l] + bs[l])
            u = torch.clamp(h @ Ws[-1] + bs[-1], -1.0, 1.0)
            z = ((torch.cat([x, u], dim=1) - in_mean) / in_std)[:, dm.n_drop:]
            for l in range(len(dm.Ws)):
                z = z @ t64(dm.Ws[l][k]) + t64(dm.bs[l][k])
                if l < len(dm.Ws) - 1:
                    z = torch.relu(z)
            xn = dmean + dstd * z + x
            su2 = torch.sum(u * u, dim=1)
            if env == 'swimmer': c = -(xn[:, 5] - 1e-2 * torch.mean(u * u, dim=1))
            elif env == 'half_cheetah': c = -torch.clamp(xn[:, 9] - 1e-1 * 0.5 * su2, -10, 10)
            elif env == 'ant': c = -(xn[:, 15] - 1e-2 * 0.5 * su2 + 0.05) * (1 - dones)
            elif env == 'humanoid': c = (xn[:, -1] - 1.5) ** 2 + 1e-2 * 1e-3 * su2
            elif env == 'hopper':
                c = -(xn[:, 5] - 0.01 * 0.5 * su2 - 10 * torch.clamp(0.45 - xn[:, 0], min=0) - 10 * torch.clamp(torch.abs(xn[:, 1]) - .2, min=0)
                      - torch.sum(torch.clamp(torch.abs(xn[:, 2:]) - 100, min=0), dim=1))
            else: c = -(xn[:, 7] - 1e-2 * 0.5 * su2)
            cost = cost + (gamma ** t) * torch.mean(c)
            if env == 'ant':
                nd = (xn[:, 2] >= 0.2) & (xn[:, 2] <= 1.0) & torch.isfinite(xn).all(dim=1)
                dones = torch.maximum(dones, (~nd).to(torch.float64))
            x = xn
        costs.append(cost)
    total = torch.stack(costs).mean()
    total.backward()
    return np.array([float(c) for c in costs]), th.grad.numpy()


@pytest.mark.parametrize('env,K,T,gamma,scale', [('swimmer', 3, 12, 1.0, 1.0), ('half_cheetah', 2, 8, 0.99, 3.0), ('ant', 2, 10, 0.95, 1.0),
                                                 ('hopper', 2, 9, 1.0, 4.0), ('snake', 2, 7, 0.9, 1.0), ('humanoid', 2, 5, 1.0, 1.0)])
def test_bptt_gradient_matches_autograd(env, K, T, gamma, scale):
    dm, theta, dims, pool = O.make_problem(env, K=K, dyn_hidden=(16, 12), pol_hidden=(8, 8), seed=11)
    rng = np.random.RandomState(3)
    theta = theta + 0.3 * rng.randn(theta.size)          # non-zero biases; some actions saturate the clip
    theta[-dims[-1]:] = 0.0
    x0 = pool[:24] * scale
    if env == 'ant':
        x0[:8, 2] = 0.15                                   # some envs are `done` from the first step on
    if env == 'hopper':
        x0[:6, 1] = 0.5; x0[6:10, 0] = 0.2; x0[10:12, 4] = 150.0     # every penalty branch active somewhere
    costs, grad = Bp.policy_costs_and_grad(dm, theta, dims, env, x0, T, gamma)
    tc, tg = torch_costs_and_grad(dm, theta, dims, env, x0, T, gamma)
    np.testing.assert_allclose(costs, tc, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(costs, O.validation_costs(dm, theta, dims, env, x0, T, gamma), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(grad, tg, rtol=1e-9, atol=1e-12)
    assert np.all(grad[-dims[-1]:] == 0.0) and np.linalg.norm(grad) > 0


def test_policy_adam_with_per_variable_clip_matches_torch():
    dims = [5, 4, 3]
    P = O.policy_num_params(dims)
    rng = np.random.RandomState(0)
    theta = rng.randn(P)
    adam = Bp.PolicyAdam(P)
    tth = torch.tensor(theta.copy(), dtype=torch.float64, requires_grad=True)
    topt = torch.optim.Adam([tth], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    segs, o = [], 0
    for i in range(len(dims) - 1):
        for n in (dims[i] * dims[i + 1], dims[i + 1]):
            segs.append((o, o + n)); o += n
    segs.append((o, P))
    for it in range(4):
        g = rng.randn(P) * (10.0 if it % 2 == 0 else 0.01)
        theta = adam.step(theta, g, dims, lr=1e-2, clip_val=0.5)
        gc = g.copy()
        for a, b in segs:
            n = np.linalg.norm(gc[a:b]); gc[a:b] = gc[a:b] * 0.5 / max(n, 0.5)
        tth.grad = torch.tensor(gc)
        # TF: lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)  ==  torch's form with eps/sqrt(1-b2^t)
        topt.param_groups[0]['eps'] = 1e-8 / np.sqrt(1.0 - 0.999 ** (it + 1))
        topt.step()
        np.testing.assert_allclose(theta, tth.detach().numpy(), rtol=1e-10, atol=1e-12)
    assert np.linalg.norm(Bp.clip_by_norm(np.ones(4) * 3.0, 2.0)) == pytest.approx(2.0)
    np.testing.assert_array_equal(Bp.clip_by_norm(np.ones(4) * 0.1, 2.0), np.ones(4) * 0.1)
