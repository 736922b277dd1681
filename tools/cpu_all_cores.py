import sys, os, time
sys.path.insert(0, '/root/repo')
from oracle import cpu_baseline
from threadpoolctl import threadpool_limits, threadpool_info
import multiprocessing
print('cpus', multiprocessing.cpu_count())
for lim in (1, None):
    if lim:
        with threadpool_limits(limits=lim):
            cb = cpu_baseline.run_iteration('swimmer', 5, (64, 64), (32, 32), B=5000, H=50, seed=0)
    else:
        cb = cpu_baseline.run_iteration('swimmer', 5, (64, 64), (32, 32), B=5000, H=50, seed=0)
    print('threads', lim, 'env-steps/s %.0f' % (cb['units'] / cb['seconds']), {k: round(v, 3) for k, v in cb['breakdown'].items()})
