#!/usr/bin/env python3
"""Cycle anatomy of k_fvpc_pipe (policy_fvpc.hip) for wave 0 of workgroup 0: needs the -DFVPC_TIMING variant
(SRC=policy_fvpc.hip tools/build_variant.sh ftiming -DFVPC_TIMING) copied over me-trpo_amd/libmetrpo.so, e.g.
   cp tools/_variants/ftiming.so me-trpo_amd/libmetrpo.so && python tools/fvpc_phases.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metrpo_amd
from metrpo_amd import _lib
eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
obs = torch.randn(N, 10, device='cuda'); act = torch.randn(N, 2, device='cuda') * 0.5; adv = torch.randn(N, device='cuda')
b = eng.make_batch(obs, act, adv, torch.zeros(N, 2, device='cuda'), torch.zeros(2, device='cuda'))
for _ in range(3):
    eng.trpo_update(b)
torch.cuda.synchronize()
fn = _lib.lib.metrpo_debug_fvpc_phases
fn.restype, fn.argtypes = C.c_int32, [C.c_void_p]
buf = (C.c_uint64 * 88)()
assert fn(buf) == 0
t = np.array(buf[:], dtype=np.int64)
print('entry -> weights + first tile: %d   prologue block A: %d' % (t[1] - t[0], t[2] - t[1]))
steps = t[8:].reshape(-1, 2)
n = int((steps[:, 1] > 0).sum())
prev = t[2]
a_, b_ = [], []
for j in range(n):
    a_.append(steps[j, 0] - prev); b_.append(steps[j, 1] - steps[j, 0]); prev = steps[j, 1]
print('steps timed: %d   block A mean %.0f (min %d max %d)   block B mean %.0f (min %d max %d)' % (n, np.mean(a_), min(a_), max(a_), np.mean(b_), min(b_), max(b_)))
print('block A per step:', ' '.join(str(x) for x in a_))
print('block B per step:', ' '.join(str(x) for x in b_))
print('last step -> after tail: %d   epilogue: %d   total: %d cycles' % (t[3] - prev, t[4] - t[3], t[4] - t[0]))
