#!/usr/bin/env python3
"""C1-sized TRPO update with and without a (world-size-1) RCCL communicator attached to the ctx: the attached path issues ncclAllReduce at every
exchange point and runs the CG vector steps as stand-alone kernels -- what every rank of an N > 1 job pays on top of the fused single-rank path,
apart from the link latency itself."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
N = 500000
obs = torch.randn(N, 10, device='cuda'); act = torch.randn(N, 2, device='cuda') * 0.5; adv = torch.randn(N, device='cuda')
om = torch.zeros(N, 2, device='cuda'); ols = torch.zeros(2, device='cuda')
b = eng.make_batch(obs, act, adv, om, ols)
theta0 = eng.get_policy().clone()
def run(label):
    for _ in range(3):
        eng.set_policy(theta0); eng.trpo_update(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.set_policy(theta0); eng.trpo_update(b)
    torch.cuda.synchronize()
    print('%-28s %.3f ms per update' % (label, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
run('fused, no communicator')
eng.comm_init(metrpo_amd.Engine.comm_unique_id(), 1, 0)
run('RCCL communicator (world 1)')
eng.comm_destroy()
