"""Round 6: rollout time per hidden width at C1's batch (swimmer K = 5, B = 5000, H = 100, 2 x 32 policy): which kernel family a width gets and what it costs
(INTEGRATION.md section 9).  usage: python tools/width_table.py [w w ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import helpers as Hh
widths = [int(a) for a in sys.argv[1:]] or [32, 48, 64, 80, 96, 128, 192, 256, 384, 512, 1024]
K, B, T = 5, 5000, 100
for w in widths:
    eng, dm, theta, pdims, pool = Hh.make_engine('swimmer', K, (w, w), (32, 32), seed=3)
    eng.set_option('QUIET', '1')
    out = eng.alloc_trajectory(B, T, T)
    poolt = torch.tensor(pool, device=eng.device)
    for _ in range(2): eng.rollout(B, T, T, 'step_rand', poolt, seed=1, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): eng.rollout(B, T, T, 'step_rand', poolt, seed=1, out=out)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    flop = K * B * T * 2.0 * (11 * w + w * w + w * 10)
    print('hidden %4d x %-4d %-20s %8.2f ms  %6.1f TFLOP/s (algorithmic)  %5.1f x the 64 x 64 FLOPs' % (w, w, eng.last_rollout_kernel(), ms, flop / ms / 1e9, (11 * w + w * w + w * 10) / (21.0 * 64 + 64 * 64)), flush=True)
    del eng
