#!/usr/bin/env python3
"""Phase breakdown of the merged post(t - 1) + pre(t) launch of the step-wise rollout (rollout_gemm.hip: k_big_pre_mfma<ENV, true>), workgroup 0 / wave 0.
Needs the instrumented variant: SRC=rollout_gemm.hip tools/build_variant.sh pptiming -DPP_TIMING, then python tools/pp_phases.py pptiming [env K hidden B]."""
import sys, os, shutil, ctypes as C
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
shutil.copy(os.path.join(_root, 'tools', '_variants', sys.argv[1] + '.so'), os.path.join(_root, 'me-trpo_amd', 'libmetrpo.so'))
import torch, metrpo_amd
from metrpo_amd import synthetic, _lib
env, K, hid, B = (sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ('ant', 10, 512, 2500)
T = 60
eng = metrpo_amd.Engine(env, K, (hid, hid), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (hid, hid), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
out = eng.alloc_trajectory(B, T, T)
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 8)()
eng.rollout(B, T, T, 'step_rand', pool, seed=0, out=out); torch.cuda.synchronize()
lib.metrpo_debug_pp_phases(buf); a = list(buf)
eng.rollout(B, T, T, 'step_rand', pool, seed=1, out=out); torch.cuda.synchronize()
lib.metrpo_debug_pp_phases(buf); b = list(buf)
n = T - 1
names = ['close step t - 1 (loads, selection, reward, reset, state stores)', 'obs row stores, policy image into LDS, barrier', 'policy chain', 'noise, action, normalised input, stores (drained)']
print('%s K=%d 2x%d B=%d: %s' % (env, K, hid, B, eng.last_rollout_kernel()))
for i, nm in enumerate(names):
    print('  %-72s %7.0f cycles = %.2f us' % (nm, (b[i] - a[i]) / n, (b[i] - a[i]) / n / 2400.0))
print('  sum %.2f us' % (sum(b[i] - a[i] for i in range(8)) / n / 2400.0))
