#!/usr/bin/env python3
"""s_memtime marks of k_finalize with a fused CG step (needs a -DFIN_TIMING variant: SRC=policy_update.hip tools/build_variant.sh ftiming
-DFIN_TIMING; python tools/fin_phases.py ftiming).  Prints, relative to the earliest block start, when every block passed each mark."""
import sys, os, shutil, ctypes as C
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
shutil.copy(os.path.join(_root, 'tools', '_variants', sys.argv[1] + '.so'), os.path.join(_root, 'me-trpo_amd', 'libmetrpo.so'))
import torch, metrpo_amd
from metrpo_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)
hum = len(sys.argv) > 2 and sys.argv[2] == 'humanoid'          # the 100-50-25 policy: 12 492 columns, 391 workgroups per reduction
N = 50000 if hum else 500000
ns, na, ph = (55, 21, (100, 50, 25)) if hum else (10, 2, (32, 32))
eng = metrpo_amd.Engine('humanoid' if hum else 'swimmer', 5, (64, 64), ph)
eng.set_policy(metrpo_amd.xavier_policy_theta(ns, ph, na))
g = torch.Generator(device='cuda').manual_seed(0)
obs = torch.randn(N, ns, device='cuda', generator=g); act = torch.randn(N, na, device='cuda', generator=g); adv = torch.randn(N, device='cuda', generator=g)
mean = eng.policy_actions(obs, torch.zeros(N, na, device='cuda'))[1]
b = eng.make_batch(obs, act * 0.1 + mean, adv, mean, torch.zeros(na, device='cuda'))
for _ in range(3): eng.trpo_update(b)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 4096)()
assert lib.metrpo_debug_fin_phases(buf) == 0
nb = min(512, (eng.P + 31) // 32)
names = ['rows summed', 'out stored', 'store visible + barrier', 'ticket known', 'CG: z loaded, p.z partials', 'CG: p.z summed', 'CG: r.r summed']
print('cycles since the block\'s own start (s_memtime is per XCD: no cross-block comparison)')
for j, nme in enumerate(names, start=1):
    vals = [buf[8 * i + j] - buf[8 * i] for i in range(nb) if buf[8 * i + j] > buf[8 * i] and buf[8 * i + j] - buf[8 * i] < 10**7]
    if vals: print('%-28s min %6d  median %6d  max %6d   (%d blocks)' % (nme, min(vals), sorted(vals)[len(vals) // 2], max(vals), len(vals)))
