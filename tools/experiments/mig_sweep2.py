import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import metrpo_amd
from metrpo_amd import synthetic
for env in ('swimmer', 'hopper', 'snake'):
    eng = metrpo_amd.Engine(env, 5, (64, 64), (32, 32))
    Ws, bs, norm = synthetic.make_dynamics(env, 5, (64, 64), seed=0)
    eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
    eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
    pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
    for B in (5000, 6000, 6500, 7000, 8192, 9000, 10000, 11000, 12000, 13000, 14000, 16384):
        res = []
        for mode in ('2', '1', ''):
            os.environ['METRPO_COOP_MODE'] = mode
            out = eng.alloc_trajectory(B, 50, 50)
            for _ in range(2): eng.rollout(B, 50, 50, 'step_rand', pool, seed=1, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5): eng.rollout(B, 50, 50, 'step_rand', pool, seed=2 + i, out=out)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 5)
        print('%-13s B=%5d  two-per-CU %.3f ms   one-per-CU %.3f ms   launch rule %.3f ms' % (env, B, res[0], res[1], res[2]), flush=True)
