"""Times the policy update of the 100-50-25 policy on the fused kernels (csrc/policy_fused3.hip) and on the GEMM path they replace.
usage: python tools/f3_time.py [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from test_gpu_engine import _update_problem

def main():
    Ns = [int(a) for a in sys.argv[1:]] or [50000, 1000000]
    for N in Ns:
        eng, th, pdims, obs, act, adv, om, ols = _update_problem('humanoid', N, pol_hidden=(100, 50, 25))
        batch = eng.make_batch(obs, act, adv, om, ols)
        v = np.random.RandomState(1).randn(eng.P)
        for path in ((True,) if os.environ.get('F3_ONLY') else (True, 'gemm')):
            eng.set_update_path(path)
            res = {}
            for name, fn in (('grad', lambda: eng.loss_grad(batch)), ('fvp', lambda: eng.fvp(batch, v)), ('losskl', lambda: eng.loss_kl(batch))):
                for _ in range(3): fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): fn()
                torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 10 * 1e3
            theta0 = eng.get_policy().clone()
            ts = []
            for _ in range(6):
                eng.set_policy(theta0); torch.cuda.synchronize(); t0 = time.perf_counter()
                eng.trpo_update(batch, max_kl=0.01); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            print('N=%d path=%s grad %.3f ms  fvp(uncached) %.3f ms  losskl %.3f ms  trpo_update %.3f ms (min of 6: %.3f)' % (N, path, res['grad'], res['fvp'], res['losskl'], np.median(ts), min(ts)), flush=True)

if __name__ == '__main__':
    main()
