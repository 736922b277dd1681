#!/usr/bin/env python3
"""Stage times of the persistent CG solve (policy_mfma.hip MODE_CG): wall-clock marks of workgroup 0 and of the last workgroup, per iteration.
Needs SRC=policy_mfma.hip tools/build_variant.sh cgptiming -DCGP_TIMING; python tools/cgp_phases.py cgptiming [N]."""
import sys, os, shutil, ctypes as C
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
shutil.copy(os.path.join(_root, 'tools', '_variants', sys.argv[1] + '.so'), os.path.join(_root, 'me-trpo_amd', 'libmetrpo.so'))
import torch, numpy as np, metrpo_amd
from metrpo_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)
for N in ([int(sys.argv[2])] if len(sys.argv) > 2 else [500000, 60000, 5000]):
    eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
    eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
    g = torch.Generator(device='cuda').manual_seed(0)
    obs = torch.randn(N, 10, device='cuda', generator=g); act = torch.randn(N, 2, device='cuda', generator=g); adv = torch.randn(N, device='cuda', generator=g)
    mean = eng.policy_actions(obs, torch.zeros(N, 2, device='cuda'))[1]
    b = eng.make_batch(obs, act * 0.1 + mean, adv, mean, torch.zeros(2, device='cuda'))
    th0 = eng.get_policy().clone()
    for _ in range(3):
        eng.set_policy(th0); eng.trpo_update(b)
    torch.cuda.synchronize()
    assert eng.last_cg_persistent()
    buf = (C.c_ulonglong * 256)()
    assert lib.metrpo_debug_cgp_times(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(2, 16, 8)[:, 1:9, :7] * 10          # ns, iterations 1..8
    names = ['wait for the vector', 'image + per-lane copies', 'tile loop', 'epilogue + row packets', 'wait for rows + column sums', 'column sums out (+ wg 0: wait for all sums)', 'wg 0: CG step, vector out']
    print('N = %d: ns per stage, median over iterations 1..8   [workgroup 0 | last workgroup]' % N)
    for w in range(2):
        d = np.diff(t[w], axis=1)
        nxt = t[w][1:, 0] - t[w][:-1, 6]
        print('   ', 'wg0 ' if w == 0 else 'last', ' '.join('%6d' % np.median(d[:, i]) for i in range(6)), '| end of iteration -> start of next %6d' % np.median(nxt), '| iteration %6d' % np.median(t[w][1:, 0] - t[w][:-1, 0]))
    print('    stages:', ' | '.join(names[1:]))
