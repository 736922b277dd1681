#!/usr/bin/env python3
"""First contact with a multi-GPU box, in one command (the path shards the imagined envs over ranks and exchanges only small
float64 sums: advantage statistics, baseline normal equations, loss + gradient, every Fisher-vector product, each line-search pair):

    python tools/multi_gpu_preflight.py --gpus N         (re-executes itself under torch.distributed.run, one rank per GPU)

Checks, in this order, and exits non-zero with the reason at the first failure:
  1. devices: N <= visible GPUs (else ranks share devices and say so), peer access (hipDeviceCanAccessPeer) for every pair;
  2. transport: Comm.attach_engine('auto') -- the one-shot direct all-reduce over peer-mapped receive regions (comm.hip) with its test
     exchange ACROSS devices, else RCCL from C, else the torch.distributed callback -- and which one every rank agreed on;
  3. the exchange itself: sums of rank-dependent ramps equal the closed form, bit for bit on every rank;
  4. latency table of SURVEY 8e: us per all-reduce for P in {1 476, 2 288, 12 492} float64 (+ the 2-element line-search pair);
  5. one sharded TRPO update (C1's policy, N = 50 000 samples per rank): theta bit-identical on all ranks afterwards.
`bench.py --gpus N` runs steps 1-3 and 5 before it times anything (it imports preflight() from here)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fail(rank, msg, rep=None):
    print('[preflight rank %d] FAIL: %s' % (rank, msg), file=sys.stderr, flush=True)
    if rep is not None:
        rep['reason'] = msg
    return False


def preflight(world, rank, device_index, latency_table=True, out=print):
    """Collective over torch.distributed (already initialised).  Returns (ok, report dict); prints from rank 0 through `out`.
    Every stage's verdict is agreed on by ALL ranks (a MIN all-reduce of the local ok flag) before anybody moves on or returns: a rank-local failure
    ends the check on every rank together instead of leaving the others inside the next collective.  The bookkeeping collectives run on tensors of
    the process group's own device type (NCCL: cuda; gloo: cpu)."""
    import torch
    import torch.distributed as dist
    import metrpo_amd
    rep = {'world': world}
    say = (lambda *a: out(*a)) if rank == 0 else (lambda *a: None)
    n_dev = torch.cuda.device_count()
    side = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    rep['devices'] = n_dev
    rep['oversubscribed'] = world > n_dev

    def agree(ok_local, msg):
        """True iff the stage passed on every rank; a failing rank prints its reason, the others name the stage."""
        flag = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=side)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return True
        _fail(rank, msg if not ok_local else 'another rank failed at: ' + msg, rep)
        return False

    # 1. peer access between every pair of devices the ranks use
    devs = sorted(set(r % n_dev for r in range(world)))
    bad = [(i, j) for i in devs for j in devs if i != j and not torch.cuda.can_device_access_peer(i, j)]
    rep['peer_access'] = not bad
    say('[preflight] %d ranks on %d visible device(s)%s; peer access between all %d device pairs: %s'
        % (world, n_dev, ' (ranks SHARE devices: functional check only)' if world > n_dev else '', len(devs) * (len(devs) - 1),
           'yes' if not bad else 'NO for %s' % bad))
    # 2. transport
    eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
    eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
    if world > n_dev:
        eng.set_exclusive(False)                             # ranks share devices: no kernel whose workgroups wait on each other inside one launch
    comm = metrpo_amd.Comm()
    if not agree(comm.world == world, 'torch.distributed world size %d != %d' % (comm.world, world)):
        return False, rep
    transport = comm.attach_engine(eng, transport=os.environ.get('METRPO_COMM', 'auto')) or 'torch.distributed callback'
    names = [None] * world
    dist.all_gather_object(names, transport)
    rep['transport'] = transport
    if not agree(len(set(names)) == 1, 'ranks disagree on the transport: %s' % names):
        return False, rep
    say('[preflight] transport agreed on by all ranks: %s%s' % (transport, '' if transport == 'one-shot' else
                                                                 ' (one-shot unavailable: %s)' % getattr(comm, 'one_shot_error', 'not attempted')))
    # 3. exchange correctness: rank r contributes (r + 1) * [1 .. n]
    sums_ok, bad_n = True, None
    for n in (2, eng.P + 1, 12493):
        ramp = torch.arange(1, n + 1, dtype=torch.float64, device=eng.device)
        buf = ramp * (rank + 1)
        comm.allreduce_sum_(buf)
        torch.cuda.synchronize()
        if not torch.equal(buf, ramp * (world * (world + 1) // 2)) and sums_ok:
            sums_ok, bad_n = False, n                        # keep going: the other ranks are inside the same sequence of exchanges
    if transport == 'one-shot':
        try:
            eng.comm_check()
        except Exception as e:
            sums_ok, bad_n = False, 'comm_check: %s' % e
    if not agree(sums_ok, 'all-reduce of %s float64 returned a wrong sum' % bad_n):
        return False, rep
    say('[preflight] all-reduce sums exact for 2 / %d / 12493 float64 on every rank' % (eng.P + 1))
    # 4. latency table
    if latency_table:
        rep['allreduce_us'] = {}
        for n in (2, 1476, 2288, 12492):
            buf = torch.zeros(n, dtype=torch.float64, device=eng.device)
            for _ in range(20):
                comm.allreduce_sum_(buf)
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(200):
                comm.allreduce_sum_(buf)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 200 * 1e6
            t = torch.tensor([us], dtype=torch.float64, device=side)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rep['allreduce_us'][n] = float(t.item())
        say('[preflight] us per all-reduce (max over ranks, 200 back-to-back, G = %d): ' % world +
            ', '.join('P = %d: %.1f' % (n, u) for n, u in rep['allreduce_us'].items()))
    # 5. one sharded update: identical theta everywhere
    N = 50000
    g = torch.Generator(device=eng.device).manual_seed(1234 + rank)
    obs = torch.randn(N, 10, device=eng.device, generator=g) * 0.5
    act, mean = eng.policy_actions(obs, torch.randn(N, 2, device=eng.device, generator=g))        # on-policy samples: the old distribution IS the policy's
    adv = torch.randn(N, device=eng.device, generator=g)
    batch = eng.make_batch(obs, act, adv, mean, eng.get_policy()[-2:], n_global=N * world)
    kw = {} if transport != 'torch.distributed callback' else {'allreduce': lambda t: dist.all_reduce(t)}
    res = eng.trpo_update(batch, **kw)
    theta = eng.get_policy().double().to(side)
    all_theta = [torch.empty_like(theta) for _ in range(world)]
    dist.all_gather(all_theta, theta)
    same = all(torch.equal(all_theta[0], t) for t in all_theta)
    rep['theta_identical'] = same
    if not agree(same, 'theta differs between ranks after one sharded update (max |diff| %.3e)'
                 % max(float((all_theta[0] - t).abs().max()) for t in all_theta)):
        return False, rep
    say('[preflight] sharded TRPO update: accepted=%s, kl=%.3e, theta bit-identical on all %d ranks' % (bool(res['accepted']), res['kl'], world))
    if transport == 'one-shot':
        eng.comm_ipc_detach()
    say('[preflight] OK')
    return True, rep


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=2)
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ:
        import socket
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]
        os.environ.setdefault('OMP_NUM_THREADS', '4')
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
                                  '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:])
    import torch
    world, rank, local = int(os.environ['WORLD_SIZE']), int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        print('[preflight] FAIL: no GPU visible'); sys.exit(2)
    torch.cuda.set_device(local % n_dev)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    torch.distributed.init_process_group('nccl' if world <= n_dev else 'gloo')      # RCCL refuses two ranks on one device
    ok, rep = preflight(world, rank, local % n_dev)
    flag = torch.tensor([1 if ok else 0])
    if torch.distributed.get_backend() == 'nccl':
        flag = flag.cuda()
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.barrier()
    sys.exit(0 if int(flag.item()) == 1 else 3)
