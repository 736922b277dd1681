import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import metrpo_amd
from metrpo_amd import synthetic
eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
Ws, bs, norm = synthetic.make_dynamics('swimmer', 5, (64, 64), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool('swimmer'), device='cuda')
for B in (4608, 5000, 5400, 5800, 6000, 6200, 6336):
    res = []
    for var in (0, 2):
        eng.set_rollout_variant(var)
        out = eng.alloc_trajectory(B, 100, 100)
        for _ in range(3): eng.rollout(B, 100, 100, 'step_rand', pool, seed=1, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10): eng.rollout(B, 100, 100, 'step_rand', pool, seed=2 + i, out=out)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    print('B=%d tiles=%d  migrating %.3f ms   two-per-CU %.3f ms' % (B, (B + 15) // 16, res[0], res[1]))
