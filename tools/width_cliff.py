"""Rollout time of 2-hidden-layer ensembles whose widths the fused 2 x 64 kernels do not hold (round 5: the tile GEMMs from 16 units up, before: the thread-per-env
kernel for every width below 128), swimmer K = 5, B = 5000, H = 100: new path against the generic kernel."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
import helpers as Hh
for hid in ((96, 96), (100, 100), (48, 48), (128, 128)):
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', 5, hid, (32, 32), seed=3)
    eng.set_option('QUIET', '1')
    out = eng.alloc_trajectory(5000, 100, 100)
    poolt = torch.tensor(pool, device=eng.device)
    for force in (False, True):
        for _ in range(2): eng.rollout(5000, 100, 100, 'step_rand', poolt, seed=1, out=out, force_generic=force)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): eng.rollout(5000, 100, 100, 'step_rand', poolt, seed=1, out=out, force_generic=force)
        torch.cuda.synchronize(); print(hid, 'generic' if force else eng.last_rollout_kernel(), round((time.perf_counter() - t0) / 3 * 1e3, 2), 'ms', flush=True)
