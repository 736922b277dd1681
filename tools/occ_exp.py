import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metrpo_amd
from metrpo_amd import synthetic
env, K, H = 'swimmer', 4, 100
eng = metrpo_amd.Engine(env, K, (64, 64), (32, 32))
Ws, bs, norm = synthetic.make_dynamics(env, K, (64, 64), seed=0)
eng.set_dynamics_layers(Ws, bs, norm['in_mean'], norm['in_std'], norm['diff_mean'], norm['diff_std'])
eng.set_policy(metrpo_amd.xavier_policy_theta(eng.ns, (32, 32), eng.na))
pool = torch.as_tensor(synthetic.make_pool(env), device='cuda')
for B in (4096, 8192):
    out = eng.alloc_trajectory(B, H, H)
    for i in range(2): eng.rollout(B, H, H, 'step_rand', pool, seed=i, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5): eng.rollout(B, H, H, 'step_rand', pool, seed=i, out=out)
    e1.record(); torch.cuda.synchronize()
    print("extra_lds=%s K=4 v1 B=%d: %.3f ms" % (os.environ.get('METRPO_EXTRA_LDS', '0'), B, e0.elapsed_time(e1) / 5), flush=True)
