import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metrpo_amd
eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
for N in (16000, 62500, 125000, 250000, 500000, 1000000, 2000000):
    obs = torch.randn(N, 10, device='cuda'); act = torch.randn(N, 2, device='cuda'); adv = torch.randn(N, device='cuda')
    om = torch.randn(N, 2, device='cuda'); ols = torch.zeros(2, device='cuda')
    b = eng.make_batch(obs, act, adv, om, ols)
    v = torch.randn(eng.P, dtype=torch.float64, device='cuda')
    for _ in range(3): eng.fvp(b, v); eng.loss_grad(b); eng.loss_kl(b)
    torch.cuda.synchronize()
    res = []
    for fn in (lambda: eng.fvp(b, v), lambda: eng.loss_grad(b), lambda: eng.loss_kl(b)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print("N=%8d  fvp %7.1f us  grad %7.1f us  losskl %7.1f us   (each incl. finalize launch)" % (N, *res), flush=True)
