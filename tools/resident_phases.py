#!/usr/bin/env python3
"""Per-phase cycle breakdown of the resident rollout kernel (rollout_resident.hip) at the params-file shape; needs the instrumented
variant: SRC=rollout_resident.hip tools/build_variant.sh restiming -DRES_TIMING [-DRES_NO_SENT for Ant], then
python tools/resident_phases.py restiming [env hidden steps-per-round [rounds ...]]   (first argument `shipped`: the library as built, wall time only)."""
import sys, os, shutil, ctypes as C
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
if len(sys.argv) > 1 and sys.argv[1] != 'shipped':      # pre-built experiment variant (tools/_variants/<name>.so) replaces the library in this scratch copy
    shutil.copy(os.path.join(_root, 'tools', '_variants', sys.argv[1] + '.so'), os.path.join(_root, 'me-trpo_amd', 'libmetrpo.so'))
import torch, metrpo_amd
from metrpo_amd import synthetic, _lib
env, K, B, H, R = 'swimmer', 5, 100, 200, 3
hid = 512
if len(sys.argv) > 2:      # python tools/resident_phases.py restiming ant 1024 500   (env, hidden width, steps per round)
    env, hid, H = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
eng = metrpo_amd.Engine(env, K, (hid, hid), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (hid, hid), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
lib = C.CDLL(_lib.LIB_PATH)
for R in ((3, 1) if len(sys.argv) <= 5 else tuple(int(a) for a in sys.argv[5:])):
    T = R * H
    out = eng.alloc_trajectory(B, T, H)
    for i in range(3):
        eng.rollout(B, T, H, 'step_rand', pool, seed=i, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        eng.rollout(B, T, H, 'step_rand', pool, seed=i, out=out)
    e1.record(); torch.cuda.synchronize()
    print('R = %d rounds of %d steps, B = %d: %s, %.3f ms per rollout = %.2f us per step' % (R, H, B, eng.last_rollout_kernel(), e0.elapsed_time(e1) / 10, e0.elapsed_time(e1) * 100 / H))
    buf = (C.c_ulonglong * 128)()
    if not hasattr(lib, 'metrpo_debug_resident_phases') or lib.metrpo_debug_resident_phases(buf) != 0:
        continue
    for role, names in ((0, ['producers 0-3: wait for X | finishers 4-7: wait for partials', 'producers: MFMA bursts + hand-over | finishers: sum, layer 2, push', 'producers: tiles per step whose requested input was late' if hid == 512 else 'wide form: slot wait + hand-over + finishing', 'wide form: loop bookkeeping']), (1, ['policy + action + X push', 'obs row + head choice', 'wait for P + sum', 'residual, reward, reset'])):
        print('  %s workgroup, cycles per step per wave:' % ('compute' if role == 0 else 'first post'))
        for i, nme in enumerate(names):
            print('    %-34s %s' % (nme, ' '.join('%7.1f' % (buf[role * 64 + w * 8 + i] / H) for w in range(8))))
    wall = (C.c_ulonglong * 1024)()
    if hasattr(lib, 'metrpo_debug_resident_wall') and lib.metrpo_debug_resident_wall(wall) == 0:
        import numpy as np
        w = np.array(wall, dtype=np.int64).reshape(4, 256)[:, 20:min(H, 256)] * 10          # ns
        wm = (C.c_ulonglong * 512)()
        if hasattr(lib, 'metrpo_debug_resident_wmax') and lib.metrpo_debug_resident_wmax(wm) == 0:
            m = np.array(wm, dtype=np.int64).reshape(2, 256)[:, 20:min(H, 256)] * 10
            print('  all workgroups serving tile 0, ns after the X push (median): X seen by the LAST of them %d | P pushed by the LAST of them %d (by workgroup 0: %d)'
                  % (np.median(m[0] - w[0]), np.median(m[1] - w[0]), np.median(w[2] - w[0])))
        print('  tile (round 0, env tile 0), ns (median over steps 20..): X push -> seen by workgroup 0 %d | its layers %d | P push -> all 16 x K slices complete at the post wave %d | post wave until next X push %d'
              % (np.median(w[1] - w[0]), np.median(w[2] - w[1]), np.median(w[3] - w[2]), np.median(w[0][1:] - w[3][:-1])))
