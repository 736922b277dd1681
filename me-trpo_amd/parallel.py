"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm,
"gloo" for the CPU tests).  The imagined-env batch axis B is sharded across ranks; the only
exchanges on the path are sum all-reduces of small float64 vectors:
advantage statistics [3], baseline normal equations [F*F+F], loss+gradient [1+P], every
Fisher-vector product [P] and each line-search (loss, kl) pair [2]."""
import os

import torch


class Comm(object):
    """World of 1 unless torch.distributed is initialised (or init_from_env() is called)."""

    def __init__(self, group=None, always_reduce=False):
        self.dist = torch.distributed if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
        self.group = group
        self.always_reduce = always_reduce and self.dist is not None      # exercise the collective even at world size 1 (tests)
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1

    @staticmethod
    def init_from_env(backend=None):
        """torchrun-style bootstrap: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT."""
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if world > 1 and not torch.distributed.is_initialized():
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            if backend == 'nccl':
                torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            torch.distributed.init_process_group(backend=backend)
        return Comm()

    def attach_engine(self, engine):
        """Bootstrap an RCCL communicator INSIDE the engine's ctx (rank 0's unique id is broadcast through torch.distributed
        once): from then on the float64 sum all-reduces of the path go through libmetrpo.so itself -- metrpo_trpo_update issues
        them from C with no host callback in the CG loop, and allreduce_sum_ below uses the same communicator."""
        if self.dist is None or self.dist.get_backend(self.group) != 'nccl':
            return False                                     # gloo (CPU-side tests): keep the torch.distributed callback path
        uid = torch.zeros(128, dtype=torch.uint8, device=engine.device)
        if self.rank == 0:
            uid.copy_(torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8))
        self.dist.broadcast(uid, src=0, group=self.group)
        engine.comm_init(bytes(uid.cpu().numpy().tobytes()), self.world, self.rank)
        self.engine = engine
        return True

    def allreduce_sum_(self, t):
        """In-place sum over ranks; stream-ordered with the caller's current stream."""
        if self.world > 1 or self.always_reduce:
            eng = getattr(self, 'engine', None)
            if eng is not None and t.dtype == torch.float64 and t.is_cuda and t.is_contiguous():
                return eng.allreduce_sum_(t)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_min_(self, t):
        if self.world > 1 or self.always_reduce:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t

    def barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.group)

    def max_float(self, x, device='cpu'):
        if self.world == 1:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def shard(self, total):
        """Contiguous shard [lo, hi) of `total` items for this rank (sizes differ by at most 1)."""
        base, rem = divmod(total, self.world)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)
