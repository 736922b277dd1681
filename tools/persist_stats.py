"""Developer tool: per-workgroup statistics of the persistent stream-K rollout (option PERSIST_STATS) at a bench config's per-GPU share."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import metrpo_amd
from metrpo_amd import synthetic
from metrpo_amd._lib import lib
cfgn, T = sys.argv[1], int(sys.argv[2])
cfg = synthetic.CONFIGS[cfgn]; env, K, H = cfg['env'], cfg['K'], cfg['H']; B = cfg['B'] // cfg['gpus']
eng = metrpo_amd.Engine(env, K, cfg['dyn_hidden'], cfg['pol_hidden'])
Ws, bs, norm = synthetic.make_dynamics(env, K, cfg['dyn_hidden'], seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, cfg['pol_hidden'], eng.na))
pool = torch.tensor(synthetic.make_pool(env), device=eng.device)
out = eng.alloc_trajectory(B, T, H)
for rep in range(3):
    eng.rollout(B, T, H, 'step_rand', pool, seed=rep, out=out)
eng.set_option('PERSIST_STATS', sys.argv[3] if len(sys.argv) > 3 else '1')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.rollout(B, T, H, 'step_rand', pool, seed=7, out=out); e1.record(); torch.cuda.synchronize()
print(eng.last_rollout_kernel(), 'rollout ms', e0.elapsed_time(e1), 'us/step', e0.elapsed_time(e1) * 1e3 / T)
buf = (C.c_ulonglong * (8 * 512))()
n = lib.metrpo_debug_persist_stats(eng._ctx, buf, 512, None)
a = np.array(buf[:8 * n], dtype=np.float64).reshape(n, 8)
cl, a = a[a[:, 7] == 1], a[a[:, 7] == 0]
us = a[:, 0] / 100.0
print('compute workgroups', len(a), 'launch us: min %.0f mean %.0f max %.0f' % (us.min(), us.mean(), us.max()))
print('blocked on a flag (longest wave) us: mean %.1f max %.1f  = %.1f %% of the launch' % ((a[:, 1] / 100).mean(), (a[:, 1] / 100).max(), 100 * a[:, 1].mean() / a[:, 0].mean()))
print('tiles per workgroup: min %d max %d; us per tile (launch / tiles): %.2f; prologue us %.1f' % (a[:, 5].min(), a[:, 5].max(), us.mean() / a[:, 5].mean(), (a[:, 6] / 100).mean()))
if len(cl):
    print('closers', len(cl), 'steps closed each: %.0f; us per closing %.2f; busy %.1f %%, waiting %.1f %% of their launch' % (cl[:, 3].mean(), cl[:, 4].sum() / max(cl[:, 3].sum(), 1) / 100, 100 * cl[:, 4].mean() / cl[:, 0].mean(), 100 * cl[:, 1].mean() / cl[:, 0].mean()))
