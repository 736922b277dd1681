"""One-shot direct all-reduce (comm.hip / xchg_device.h): G processes on ONE GPU map each other's receive regions
through hipIpcGetMemHandle and run the sharded TRPO update with the exchange in the tail of the reduction kernels.  The reference has
no counterpart (single process, utils.py:229-232); the contract is theta(G ranks x N/G samples) == theta(1 rank x N samples) within
SURVEY 8d's tolerances, and bit-identical vectors on all ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest
import tolerances as TOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu(t):
    return t.detach().cpu().numpy()


def _launch(world, out_file, path, port, extra_env=None, n_updates=1):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', '_multi_rank_oneshot.py'), out_file, path, str(n_updates)]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)


@pytest.mark.parametrize('path,world', [('mfma', 4), ('gemm', 4), ('generic', 2), ('mfma', 8), ('f3', 4)])
def test_ranks_on_one_gpu_equal_one_rank(path, world, tmp_path):
    from test_gpu_engine import _update_problem
    out_file = str(tmp_path / 'ranks.npz')
    res = _launch(world, out_file, path, 29531 + world)
    assert res.returncode == 0, (res.stdout[-1500:] + res.stderr[-3000:])
    many = np.load(out_file)
    eng, th, pdims, obs, act, adv, om, ols = (_update_problem('humanoid', 6000, seed=29, pol_hidden=(100, 50, 25)) if path == 'f3' else _update_problem(N=6000, seed=29))
    eng.set_update_path({'mfma': True, 'generic': False, 'gemm': 'gemm', 'f3': True}[path])
    one = eng.trpo_update(eng.make_batch(obs, act, adv, om, ols), want_vectors=True)
    # float32 partial sums over different sample groupings: 1e-7-relative differences in g, amplified by 10 CG iterations in d
    # (g rel-L2 1e-5, d rel-L2 1e-3)
    np.testing.assert_allclose(many['g'], cpu(one['g']), rtol=0, atol=TOL.MULTI_RANK_GRAD * np.abs(cpu(one['g'])).max())
    rel = np.linalg.norm(many['d'] - cpu(one['d'])) / np.linalg.norm(cpu(one['d']))
    assert rel < 1e-3
    assert abs(float(many['loss_before']) - one['loss_before']) < 1e-6
    assert abs(float(many['beta']) - one['beta']) < 1e-3 * one['beta'] and int(many['n_backtrack']) == one['n_backtrack']
    assert bool(many['accepted']) and one['accepted']
    step = np.abs(cpu(eng.get_policy()) - th).max()
    np.testing.assert_allclose(many['theta'], cpu(eng.get_policy()), rtol=0, atol=TOL.MULTI_RANK_THETA * step + 1e-7)


def test_missing_rank_times_out_instead_of_hanging():
    """A rank whose peers never send must come back with METRPO_EHIP after the time limit -- not spin forever on the GPU.  One process,
    two contexts: ctx A exports its region, ctx B's blob stands in for the absent peer (a region nobody writes to ... but hipIpcOpenMemHandle
    refuses handles of the own process, so the attach itself must fail cleanly; the time-out is exercised by the 2-process variant below)."""
    import torch
    import metrpo_amd
    from test_gpu_engine import _update_problem
    eng = _update_problem(N=500, seed=3)[0]
    blob = eng.comm_ipc_export()
    with pytest.raises(metrpo_amd._lib.MetrpoError, match='two ranks in one process'):
        eng.comm_ipc_attach(blob + blob, 2, 0)
    assert eng.comm_transport() is None
    # world size 1 attach is a no-op transport: updates stay on the single-rank path
    eng.comm_ipc_attach(blob, 1, 0)
    assert eng.comm_transport() is None
    eng.comm_ipc_detach()
    t = torch.ones(3, dtype=torch.float64, device='cuda')
    with pytest.raises(metrpo_amd._lib.MetrpoError, match='no communicator'):
        eng.allreduce_sum_(t)


def test_absent_peer_reports_timeout(tmp_path):
    """Two processes attach; rank 1 then leaves without ever exchanging.  Rank 0's exchange must give up after METRPO_XCHG_TIMEOUT_MS and
    surface METRPO_EHIP through comm_check / trpo_update."""
    script = tmp_path / 'absent.py'
    script.write_text('''
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import metrpo_amd
from test_gpu_engine import _update_problem
dist.init_process_group("gloo"); rank = dist.get_rank(); torch.cuda.set_device(0)
eng = _update_problem(N=500, seed=3)[0]
comm = metrpo_amd.Comm()
assert comm.attach_engine(eng, transport="one-shot") == "one-shot"
eng.comm_set_timeout_ms(500)
dist.barrier()
if rank == 0:
    t0 = time.time()
    eng.allreduce_sum_(torch.ones(4, dtype=torch.float64, device="cuda"))
    try:
        eng.comm_check(); print("NO-ERROR")
    except metrpo_amd._lib.MetrpoError as e:
        print("TIMED-OUT %%.2f %%s" %% (time.time() - t0, e))
dist.barrier()
''' % (ROOT, ROOT))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29547', str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR='127.0.0.1'), cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    assert 'TIMED-OUT' in res.stdout and 'did not arrive' in res.stdout, res.stdout[-2000:]


def test_update_works_after_a_timed_out_exchange_and_detach(tmp_path):
    """What Comm.attach_engine('auto') relies on when its test exchange times out: after metrpo_comm_ipc_detach the context must be usable again --
    the sticky time-out cell of the one-shot exchange may not make every later metrpo_trpo_update fail (round-3 advisor finding)."""
    script = tmp_path / 'recover.py'
    script.write_text('''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import metrpo_amd
from test_gpu_engine import _update_problem
dist.init_process_group("gloo"); rank = dist.get_rank(); torch.cuda.set_device(0)
eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=500, seed=3)
comm = metrpo_amd.Comm()
assert comm.attach_engine(eng, transport="one-shot") == "one-shot"
eng.comm_set_timeout_ms(300)
dist.barrier()
if rank == 0:
    eng.allreduce_sum_(torch.ones(4, dtype=torch.float64, device="cuda"))          # rank 1 never joins: times out
    try:
        eng.comm_check(); print("NO-ERROR")
    except metrpo_amd._lib.MetrpoError:
        print("TIMED-OUT")
dist.barrier()
eng.comm_ipc_detach()
if rank == 0:
    batch = eng.make_batch(obs, act, adv, om, ols)
    res = eng.trpo_update(batch)                                                    # single rank now: must not report the old time-out
    eng.comm_check()
    print("UPDATE-OK", res["accepted"])
dist.barrier()
''' % (ROOT, ROOT))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29549', str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR='127.0.0.1'), cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    assert 'TIMED-OUT' in res.stdout and 'UPDATE-OK' in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_bench_self_launches_for_n_gpus():
    """`python bench.py --gpus 4` the way the driver invokes `--gpus 1` (plain python, no launcher, no WORLD_SIZE): bench.py re-executes
    itself under torch.distributed.run; on this 1-GPU box the 4 ranks share cuda:0 (gloo for the bookkeeping, the one-shot transport
    for the path's all-reduces) and rank 0 prints ONE JSON line with n_gpus = 4."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '3', '--warmup', '1'],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 4 and rec['steps'] == 3 and rec['scaling'] == 'weak' and rec['value'] > 0
    assert 'one-shot' in rec['allreduce_us']['transport'], rec['allreduce_us']
    assert rec['preflight'] is not None and rec['preflight']['ok'] is True and rec['preflight']['theta_identical'] is True, rec['preflight']
    assert 'oversubscribed' in rec and 'cpu_baseline' not in rec
    assert abs(rec['value'] - 4 * 5 * 5000 * 100 / (rec['ms_per_step'] * 1e-3)) / rec['value'] < 1e-9


def test_preflight_tool_two_ranks_on_one_device_prints_the_latency_table():
    """`python tools/multi_gpu_preflight.py --gpus 2` (the first command on a multi-GPU box) on this 1-GPU box: the two ranks share cuda:0
    over gloo + the one-shot transport; exit code 0, every stage's line, and the us-per-all-reduce table for P = 2 / 1476 / 2288 / 12492."""
    import re
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'multi_gpu_preflight.py'), '--gpus', '2'],
                         capture_output=True, text=True, timeout=600, env=dict(env, MASTER_ADDR='127.0.0.1'), cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert '[preflight] OK' in res.stdout and 'theta bit-identical on all 2 ranks' in res.stdout
    line = [l for l in res.stdout.splitlines() if 'us per all-reduce' in l]
    assert len(line) == 1, res.stdout[-2000:]
    table = dict((int(a), float(b)) for a, b in re.findall(r'P = (\d+): ([0-9.]+)', line[0]))
    assert sorted(table) == [2, 1476, 2288, 12492] and all(0.5 < v < 5000 for v in table.values()), table


def test_preflight_collectives_run_on_an_nccl_group():
    """the preflight's bookkeeping collectives must live on the process group's device type.  One rank, backend nccl (RCCL accepts a
    single-rank communicator on a 1-GPU box): preflight() returns ok with the report filled in."""
    script = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tools'))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29571', world_size=1, rank=0)
from multi_gpu_preflight import preflight
ok, rep = preflight(1, 0, 0, latency_table=True)
assert ok and rep['theta_identical'] and sorted(rep['allreduce_us']) == [2, 1476, 2288, 12492], rep
print('NCCL-PREFLIGHT-OK', rep['transport'])
dist.destroy_process_group()
""" % (ROOT, ROOT)
    res = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0 and 'NCCL-PREFLIGHT-OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
