"""Helper of tests/test_gpu_api.py::test_two_ranks_equal_one_rank_fused_update: run under torch.distributed.run with 2 processes
(gloo), both on cuda:0.  Each rank holds HALF of a fixed sample batch, runs the fused C TRPO driver with the all-reduce hook and
rank 0 writes theta_new + diagnostics."""
import os, sys, json
import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
from test_gpu_engine import _update_problem          # noqa: E402


def main(out_path, path):
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    eng, th, pdims, obs, act, adv, om, ols = _update_problem(N=6000, seed=29)
    eng.set_update_path({'mfma': True, 'generic': False, 'gemm': 'gemm'}[path])
    N = len(obs)
    lo, hi = rank * N // world, (rank + 1) * N // world
    batch = eng.make_batch(obs[lo:hi], act[lo:hi], adv[lo:hi], om[lo:hi], ols[lo:hi], n_global=N)
    calls = []

    def ar(t):
        calls.append(t.numel())
        dist.all_reduce(t)
    out = eng.trpo_update(batch, allreduce=ar, want_vectors=True)
    torch.cuda.synchronize()
    theta = eng.get_policy().double().cpu().numpy()
    gathered = [None] * world
    dist.all_gather_object(gathered, theta.tobytes())
    assert all(g == gathered[0] for g in gathered), "ranks ended with different theta"
    if rank == 0:
        np.savez(out_path, theta=theta, g=out['g'].cpu().numpy(), d=out['d'].cpu().numpy(), beta=out['beta'], n_backtrack=out['n_backtrack'],
                 accepted=out['accepted'], loss=out['loss'], kl=out['kl'], calls=np.array(calls))
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
