#!/usr/bin/env python3
"""Cost of the sharded update's exchanges, measurable on ONE GPU:

  python tools/comm_overhead.py                  world 1: fused update with / without a world-size-1 RCCL communicator
  python tools/comm_overhead.py --ranks G        re-executes itself under torch.distributed.run with G processes on cuda:0 (gloo side
                                                 channel) and reports, for the one-shot transport (comm.hip):
      * us per stand-alone exchange (k_xchg_allreduce) for the path's vector lengths 2, 1+P, and Humanoid's 12 493 / 13 110;
      * ms per C1-sized update with the N = 500 000 samples split over the G ranks (the GPU does the same total work as the
        single-rank update, so the difference is what the exchanges and the co-scheduling of G processes cost);
      * the same update with the exchanges routed through the torch.distributed (gloo) host callback, for scale.
    Ranks sharing one GPU exchange through local HBM, so the figures are the protocol's floor (launch + packet write + poll), without
    the xGMI hop (~1-2 us per direction on a directly connected pair)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('--ranks', type=int, default=1)
args = ap.parse_args()
if args.ranks > 1 and 'WORLD_SIZE' not in os.environ:
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]
    os.environ.setdefault('OMP_NUM_THREADS', '4')
    os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.ranks),
                              '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:])

import torch, metrpo_amd
world = int(os.environ.get('WORLD_SIZE', '1'))
if world > 1:
    torch.distributed.init_process_group('gloo')
rank = torch.distributed.get_rank() if world > 1 else 0
torch.cuda.set_device(0)
eng = metrpo_amd.Engine('swimmer', 5, (64, 64), (32, 32))
eng.set_policy(metrpo_amd.xavier_policy_theta(10, (32, 32), 2))
N = 500000 // world
g = torch.Generator(device='cuda').manual_seed(rank)
obs = torch.randn(N, 10, device='cuda', generator=g); act = torch.randn(N, 2, device='cuda', generator=g) * 0.5
adv = torch.randn(N, device='cuda', generator=g)
om = torch.zeros(N, 2, device='cuda'); ols = torch.zeros(2, device='cuda')
b = eng.make_batch(obs, act, adv, om, ols, n_global=N * world)
theta0 = eng.get_policy().clone()


def say(*a):
    if rank == 0:
        print(*a, flush=True)


def run(label, **kw):
    for _ in range(3):
        eng.set_policy(theta0); eng.trpo_update(b, **kw)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.set_policy(theta0); eng.trpo_update(b, **kw)
    torch.cuda.synchronize()
    say('%-64s %.3f ms per update' % (label, (time.perf_counter() - t0) / 20 * 1e3))


if world == 1:
    run('fused, single rank, N = %d' % N)
    eng.comm_init(metrpo_amd.Engine.comm_unique_id(), 1, 0)
    run('RCCL communicator (world 1)')
    eng.comm_destroy()
    sys.exit(0)

comm = metrpo_amd.Comm()
run('%d ranks x N = %d, torch.distributed gloo via host callback' % (world, N), allreduce=lambda t: torch.distributed.all_reduce(t))
assert comm.attach_engine(eng, transport='one-shot') == 'one-shot'
run('%d ranks x N = %d, one-shot exchange in the reduction tails' % (world, N))
for n in (2, eng.P + 1, 12493, 13110):
    buf = torch.zeros(n, dtype=torch.float64, device='cuda')
    for _ in range(20):
        eng.allreduce_sum_(buf)
    torch.cuda.synchronize(); torch.distributed.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        eng.allreduce_sum_(buf)
    e1.record(); torch.cuda.synchronize()
    say('stand-alone one-shot exchange, %6d float64, %d ranks on one GPU: %.2f us' % (n, world, e0.elapsed_time(e1) * 1e3 / 200))
eng.comm_check()
torch.distributed.barrier()
eng.comm_ipc_detach()
