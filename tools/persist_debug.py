"""Developer tool: persistent stream-K rollout vs the launch-per-step path, mismatch map per (step, 128-env row block)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import helpers as Hh
env, K, dh, B, T, H = sys.argv[1], int(sys.argv[2]), tuple(int(x) for x in sys.argv[3].split(',')), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
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
