#!/usr/bin/env python3
"""Per-phase cycle breakdown of the cooperative rollout kernel (needs a library built with `make EXTRA=-DCOOP_TIMING`)."""
import sys, os, shutil, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:      # pre-built experiment variant (tools/_variants/<name>.so) replaces the library in this scratch copy
    shutil.copy(os.path.join(_root, 'tools', '_variants', sys.argv[1] + '.so'), os.path.join(_root, 'me-trpo_amd', 'libmetrpo.so'))
import torch, metrpo_amd
from metrpo_amd import synthetic, _lib
env, K, H = 'swimmer', 5, 100
eng = metrpo_amd.Engine(env, K, (64, 64), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (64, 64), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
lib = C.CDLL(_lib.LIB_PATH)
names = ['policy', 'rng+action', 'dyn L0 + H0 write', 'barrier 1', 'dyn L1 + L2 + PART write', 'barrier 2', 'selection',
         'reward/done/stores', 'reset/advance', 'obs store + loop', '  (stores + su2 after B1)', '  (layer 1)', '-', '-']
for B in (4096, 5000):
    out = eng.alloc_trajectory(B, H, H)
    for i in range(3):
        eng.rollout(B, H, H, 'step_rand', pool, seed=i, out=out)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert lib.metrpo_debug_coop_phases(buf) == 0
    steps = H if B == 4096 else None            # B = 5000: workgroup 0 runs 123 tile-steps (migration schedule)
    n = H if B == 4096 else -(-((B + 15) // 16) * H // 256)
    v = [[buf[16 * w + i] / n for i in range(14)] for w in range(4)]
    print('B=%d workgroup 0, cycles/step per wave: totals %s' % (B, ' '.join('%6.0f' % sum(x) for x in v)))
    for i, nme in enumerate(names):
        print('    %-28s %s' % (nme, ' '.join('%6.0f' % v[w][i] for w in range(4))))
