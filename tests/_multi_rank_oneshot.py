"""Helper of tests/test_gpu_comm.py: G processes (torch.distributed.run, gloo) that all sit on cuda:0 attach the one-shot direct
all-reduce of comm.hip (peer-mapped receive regions, hipIpcGetMemHandle works between processes on ONE device) and run

  * stand-alone exchanges of several lengths (2, 3, [F*F+F], a vector longer than one slot), each against a gloo all-reduce,
  * the fused TRPO update (metrpo_trpo_update) on 1/G of a fixed batch -- the exchange rides in the tail of k_finalize (fused MFMA
    and generic kernels) or runs as the stand-alone kernel between the GEMM path's reductions and the CG step.

Rank 0 writes theta_new + diagnostics; every rank asserts that all ranks ended with bit-identical vectors."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
from test_gpu_engine import _update_problem          # noqa: E402


def same_on_all_ranks(t, world):
    got = [None] * world
    dist.all_gather_object(got, t.detach().cpu().numpy().tobytes())
    return all(g == got[0] for g in got)


def main(out_path, path, n_updates):
    import metrpo_amd
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    eng, th, pdims, obs, act, adv, om, ols = (_update_problem('humanoid', 6000, seed=29, pol_hidden=(100, 50, 25)) if path == 'f3'      # the fused three-hidden-layer kernels
                                              else _update_problem(N=6000, seed=29))
    eng.set_update_path({'mfma': True, 'generic': False, 'gemm': 'gemm', 'f3': True}[path])
    assert path != 'f3' or eng.update_path(6000) == 'mfma'
    comm = metrpo_amd.Comm()
    assert comm.attach_engine(eng, transport='one-shot') == 'one-shot' and eng.comm_transport() == 'one-shot'

    # ---- stand-alone exchanges vs gloo
    gen = torch.Generator().manual_seed(100 + rank)
    for n in (2, 3, 600, 13110, 40000):
        mine = torch.randn(n, dtype=torch.float64, generator=gen)
        ref = mine.clone(); dist.all_reduce(ref)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        in_rank_order = torch.zeros_like(mine)
        for p in parts:
            in_rank_order += p                                            # the order the kernel adds the slots in
        got = comm.allreduce_sum_(mine.cuda())
        assert torch.equal(got.cpu(), in_rank_order), "n=%d: not the rank-ordered float64 sum" % n
        assert torch.allclose(got.cpu(), ref, rtol=1e-14, atol=1e-14)
        assert same_on_all_ranks(got, world)
    eng.comm_check()

    # ---- fused update on this rank's share
    N = len(obs)
    lo, hi = rank * N // world, (rank + 1) * N // world
    batch = eng.make_batch(obs[lo:hi], act[lo:hi], adv[lo:hi], om[lo:hi], ols[lo:hi], n_global=N)
    outs = []
    for _ in range(n_updates):
        out = eng.trpo_update(batch, want_vectors=True)
        outs.append(out)
    torch.cuda.synchronize()
    eng.comm_check()
    out = outs[0]
    theta = eng.get_policy().double()
    for t in (theta, out['g'], out['d']):
        assert same_on_all_ranks(t, world), "ranks ended with different vectors"
    # ---- the same update in two halves (metrpo_trpo_update_begin / _end): three speculative line-search trials decided on the device, the
    #      ones after the accepted trial leave at once but keep the exchange sequence in step -- same theta, bit for bit, and the protocol lives on
    eng.set_policy(th)
    assert eng.trpo_update(batch, spec_trials=3) is None
    late = eng.trpo_update_end()
    assert late['n_backtrack'] == out['n_backtrack'] and late['accepted'] == out['accepted'] and late['loss'] == out['loss'] and late['kl'] == out['kl']
    if n_updates == 1:
        assert torch.equal(eng.get_policy().double(), theta), "two-halves update differs from the one-call update"
    assert same_on_all_ranks(eng.get_policy(), world)
    probe = comm.allreduce_sum_(torch.full((5,), float(rank + 1), dtype=torch.float64, device='cuda'))
    assert torch.equal(probe.cpu(), torch.full((5,), world * (world + 1) / 2.0, dtype=torch.float64))
    eng.comm_check()
    if rank == 0:
        np.savez(out_path, theta=theta.cpu().numpy(), g=out['g'].cpu().numpy(), d=out['d'].cpu().numpy(), beta=out['beta'],
                 n_backtrack=out['n_backtrack'], accepted=out['accepted'], loss=out['loss'], kl=out['kl'],
                 loss_before=out['loss_before'])
    dist.barrier()
    eng.comm_ipc_detach()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
