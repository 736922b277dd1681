#!/usr/bin/env python3
"""Phase boundaries (s_memrealtime, 10 ns ticks) of the persistent CG solve, thread 0 of the first and the last workgroup, per iteration (needs
tools/_variants/<name>.so built with -DCGP_TIMING: SRC=policy_mfma.hip tools/build_variant.sh cgptiming -DCGP_TIMING;  python tools/cgp_phases.py cgptiming [N])."""
import sys, os, shutil, ctypes as C
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
shutil.copy(os.path.join(_root, 'tools', '_variants', sys.argv[1] + '.so'), os.path.join(_root, 'me-trpo_amd', 'libmetrpo.so'))
import torch, metrpo_amd
from metrpo_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)
for N in ([int(sys.argv[2])] if len(sys.argv) > 2 else [50000, 500000]):
    eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
    eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
    eng.set_option('CG_PERSIST', 1)
    g = torch.Generator(device='cuda').manual_seed(0)
    obs = torch.randn(N, 10, device='cuda', generator=g); act = torch.randn(N, 2, device='cuda', generator=g); adv = torch.randn(N, device='cuda', generator=g)
    mean = eng.policy_actions(obs, torch.zeros(N, 2, device='cuda'))[1]
    b = eng.make_batch(obs, act * 0.1 + mean, adv, mean, torch.zeros(2, device='cuda'))
    for _ in range(3): eng.trpo_update(b)
    torch.cuda.synchronize()
    assert eng.cg_persist_launches() == 3
    buf = (C.c_ulonglong * 256)()
    assert lib.metrpo_debug_cgp_phases(buf) == 0
    names = ['tangent loads, image -> LDS', 'tile loop + row store', 'barrier 1 (wait for all rows)', 'column sums', 'barrier 2 arrival', 'CG step / wait for it', 'acquire', ]
    for wg, nm in ((0, 'first workgroup'), (1, 'last workgroup')):
        print('N=%d, %s: 10 ns ticks per phase, iterations 0..9' % (N, nm))
        for i, nme in enumerate(names):
            print('    %-30s %s' % (nme, ' '.join('%6d' % (buf[(wg * 16 + it) * 8 + i + 1] - buf[(wg * 16 + it) * 8 + i]) for it in range(10))))
        print('    %-30s %s' % ('whole iteration', ' '.join('%6d' % (buf[(wg * 16 + it) * 8 + 7] - buf[(wg * 16 + it) * 8]) for it in range(10))))
