#!/usr/bin/env python3
"""Cycles per steady-state iteration of the pipelined cached-activation FVP loop (policy_mfma.hip, -DPOL_TIMING variant copied over the library)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import _lib
eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
N = 500000
obs = torch.randn(N, 10, device='cuda'); act = torch.randn(N, 2, device='cuda') * 0.5; adv = torch.randn(N, device='cuda')
b = eng.make_batch(obs, act, adv, torch.zeros(N, 2, device='cuda'), torch.zeros(2, device='cuda'))
for _ in range(3):
    eng.trpo_update(b)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 128)()
assert lib.metrpo_debug_pol_phases(buf) == 0
for w in range(8):
    cyc, nit = buf[8 * (8 + w)], buf[8 * (8 + w) + 1]
    ph = [buf[8 * (8 + w) + 2 + i] / max(nit + 1, 1) for i in range(5)]
    print('wave %d: %d iterations, %.0f cycles per iteration; fetch+copy %.0f | run 1 %.0f | VALU 1 %.0f | run 2 %.0f | VALU 2 %.0f' % ((w, nit, cyc / max(nit, 1)) + tuple(ph)))
