#!/usr/bin/env python3
"""Policy-update kernel durations (rocprofv3 --kernel-trace) at several sample counts N: a linear fit over the tile count separates the
fixed per-launch cost (weight image, block reduction, partial rows) from the per-tile cost.  Run on the GPU box:  python tools/update_sizes.py"""
import sys, os, subprocess, sqlite3, glob, tempfile, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, root)
    import torch, metrpo_amd
    N = int(sys.argv[2])
    eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
    eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
    g = torch.Generator(device='cuda').manual_seed(0)
    obs = torch.randn(N, 10, device='cuda', generator=g); act = torch.randn(N, 2, device='cuda', generator=g); adv = torch.randn(N, device='cuda', generator=g)
    theta0 = eng.get_policy().clone()
    for _ in range(4):
        eng.set_policy(theta0)
        mean = eng.policy_actions(obs, torch.zeros(N, 2, device='cuda'))[1]
        b = eng.make_batch(obs, act * 0.1 + mean, adv, mean, torch.zeros(2, device='cuda'))
        eng.trpo_update(b)
    torch.cuda.synchronize()
    sys.exit(0)
for N in (32768, 65536, 131072, 262144, 500000, 1000000):
    d = tempfile.mkdtemp(dir='/tmp')
    r = subprocess.run(['rocprofv3', '--kernel-trace', '-d', d, '-o', 't', '--', sys.executable, os.path.abspath(__file__), '--child', str(N)],
                       capture_output=True, text=True, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
    db = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
    if not db:
        print(N, 'no trace', r.stderr[-400:]); continue
    rows = sqlite3.connect(db[0]).execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
    pick = [(n, c, a, m) for n, c, a, m in rows if 'k_policy_mfma' in n or 'k_finalize' in n]
    print('N=%8d  ' % N + '  '.join('%s x%d avg %.1f min %.1f' % (n.split('(')[0].replace('void k_policy_mfma', 'pol')[:22], c, a / 1e3, m / 1e3) for n, c, a, m in pick), flush=True)
    shutil.rmtree(d, ignore_errors=True)
