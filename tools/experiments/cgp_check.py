import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np, metrpo_amd
def run(N, env=None):
    eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
    eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
    g = torch.Generator(device='cuda').manual_seed(0)
    obs = torch.randn(N, 10, device='cuda', generator=g); act = torch.randn(N, 2, device='cuda', generator=g); adv = torch.randn(N, device='cuda', generator=g)
    mean = eng.policy_actions(obs, torch.zeros(N, 2, device='cuda'))[1]
    b = eng.make_batch(obs, act * 0.1 + mean, adv, mean, torch.zeros(2, device='cuda'))
    th0 = eng.get_policy().clone()
    out = {}
    for mode in ('persistent', 'launches'):
        if mode == 'launches': os.environ['METRPO_NO_PERSISTENT_CG'] = '1'
        else: os.environ.pop('METRPO_NO_PERSISTENT_CG', None)
        eng.set_policy(th0)
        r = eng.trpo_update(b, want_vectors=True)
        pers = eng.last_cg_persistent()
        for _ in range(3):
            eng.set_policy(th0); eng.trpo_update(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            eng.set_policy(th0); eng.trpo_update(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
        out[mode] = (r, pers, dt)
        print(N, mode, 'persistent=%s' % pers, '%.3f ms' % dt, 'beta %.6g loss %.6g kl %.6g nb %d' % (r['beta'], r['loss'], r['kl'], r['n_backtrack']), flush=True)
    a, b_ = out['persistent'][0], out['launches'][0]
    d = (a['d'] - b_['d']).abs().max().item() / b_['d'].abs().max().item()
    print('   rel diff of direction %.3g, beta rel %.3g' % (d, abs(a['beta'] - b_['beta']) / b_['beta']))
for N in (500000, 60000, 5000):
    run(N)
