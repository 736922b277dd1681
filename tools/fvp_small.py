#!/usr/bin/env python3
"""FVP / grad / loss launches at a tiny sample count: kernel time ~= fixed per-block prologue + epilogue cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
obs = torch.randn(N, 10, device='cuda'); act = torch.randn(N, 2, device='cuda'); adv = torch.randn(N, device='cuda')
om = torch.randn(N, 2, device='cuda'); ols = torch.zeros(2, device='cuda')
b = eng.make_batch(obs, act, adv, om, ols)
v = torch.randn(eng.P, dtype=torch.float64, device='cuda')
for _ in range(10): eng.fvp(b, v)
eng.loss_grad(b); eng.loss_kl(b)
torch.cuda.synchronize()
