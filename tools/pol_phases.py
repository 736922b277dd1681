#!/usr/bin/env python3
"""Phase boundaries (s_memtime) of the cached-activation FVP kernel, workgroup 0 (needs tools/_variants/<name>.so built with -DPOL_TIMING:
SRC=policy_mfma.hip tools/build_variant.sh ptiming -DPOL_TIMING;  python tools/pol_phases.py ptiming [N])."""
import sys, os, shutil, ctypes as C
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
shutil.copy(os.path.join(_root, 'tools', '_variants', sys.argv[1] + '.so'), os.path.join(_root, 'me-trpo_amd', 'libmetrpo.so'))
import torch, metrpo_amd
from metrpo_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)
for N in ([int(sys.argv[2])] if len(sys.argv) > 2 else [32768, 500000]):
    eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
    eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
    g = torch.Generator(device='cuda').manual_seed(0)
    obs = torch.randn(N, 10, device='cuda', generator=g); act = torch.randn(N, 2, device='cuda', generator=g); adv = torch.randn(N, device='cuda', generator=g)
    mean = eng.policy_actions(obs, torch.zeros(N, 2, device='cuda'))[1]
    b = eng.make_batch(obs, act * 0.1 + mean, adv, mean, torch.zeros(2, device='cuda'))
    for _ in range(3): eng.trpo_update(b)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert lib.metrpo_debug_pol_phases(buf) == 0
    names = ['image -> LDS, biases', 'first fetch (wait)', 'tile loop', 'wait for the other waves', 'wave rows -> LDS', 'block sum + row store']
    print('N=%d: s_memtime ticks from the wave\'s first mark (waves 0..7 of workgroup 0); tick = 10 ns at 100 MHz' % N)
    for i, nme in enumerate(names):
        print('    %-26s %s' % (nme, ' '.join('%6d' % (buf[8 * w + i + 1] - buf[8 * w + i]) for w in range(8))))
    print('    %-26s %s' % ('start skew vs wave 0', ' '.join('%6d' % (buf[8 * w] - buf[0]) for w in range(8))))
