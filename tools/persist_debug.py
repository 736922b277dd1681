"""Developer tool: persistent stream-K rollout vs the launch-per-step path, mismatch map per (step, 128-env row block)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import helpers as Hh
env, K, dh, B, T, H = sys.argv[1], int(sys.argv[2]), tuple(int(x) for x in sys.argv[3].split(',')), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
if len(sys.argv) > 7:
    e0 = Hh.make_engine('swimmer', 5, (64, 64), (32, 32), seed=1)
    e0[0].rollout(64, 4, 4, 'step_rand', e0[4], seed=1); torch.cuda.synchronize()
    if sys.argv[7] == 'del':
        del e0
eng, dm, theta, pdims, pool = Hh.make_engine(env, K, dh, (32, 32), seed=71)
eng.set_option('STREAMK', '1'); eng.set_rollout_variant(1); eng.set_option('QUIET', '1')
a = eng.rollout(B, T, H, 'step_rand', pool, seed=3); ka = eng.last_rollout_kernel()
a = {k: getattr(a, k).clone() for k in ('obs', 'act', 'mean', 'rew', 'done', 'tpath')}
eng.set_option('NO_PERSIST', '1')
b = eng.rollout(B, T, H, 'step_rand', pool, seed=3); kb = eng.last_rollout_kernel()
torch.cuda.synchronize()
print(ka, kb)
RB = (B + 127) // 128
for k in ('obs', 'mean', 'rew', 'done'):
    x, y = a[k].float().cpu().numpy(), getattr(b, k).float().cpu().numpy()
    x = x.reshape(T, B, -1); y = y.reshape(T, B, -1)
    bad = (x != y).any(-1)
    print(k, 'mismatching envs per (t, rb):')
    for t in range(T):
        print('  t=%2d' % t, [int(bad[t, r * 128:(r + 1) * 128].sum()) for r in range(RB)], 'zeros:', int((x[t] == 0).all(-1).sum()), 'maxdiff %.3g' % np.abs(x[t] - y[t]).max())
x, y = a['obs'].float().cpu().numpy(), b.obs.float().cpu().numpy()
d = np.abs(x[1] - y[1])
print('obs[1] |diff| max per dim:', np.round(d.max(0), 5))
print('obs[1] |diff| first env:', np.round(x[1][0] - y[1][0], 5), ' values', np.round(y[1][0], 4))
print('act diff at t=1', float(np.abs(a['act'][1].cpu().numpy() - b.act[1].cpu().numpy()).max()))
