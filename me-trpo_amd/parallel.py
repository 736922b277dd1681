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

    def attach_engine(self, engine, transport=None):
        """Move the float64 sum all-reduces of the path into libmetrpo.so (no Python and no host callback inside the CG loop).
        transport (default: env METRPO_COMM, else 'auto'):
          'one-shot'  peer-mapped receive regions, every rank writes its vector into every peer's slot and adds the slots in rank
                      order (the exchanges are latency-bound, xGMI is point-to-point); works over any backend because
                      only 128-byte IPC handles travel through torch.distributed;
          'rccl'      an RCCL communicator owned by the ctx (ncclAllReduce issued from C); needs the nccl backend;
          'auto'      one-shot, verified by a test exchange; RCCL if that is unavailable; else the torch.distributed callback.
        Returns the transport in use ('one-shot', 'rccl') or False."""
        if self.dist is None or self.world < 2:
            return False
        transport = transport or os.environ.get('METRPO_COMM', 'auto')
        self.engine = None
        if transport in ('auto', 'one-shot'):
            if self._attach_one_shot(engine):
                self.engine = engine
                return 'one-shot'
            if transport == 'one-shot':
                raise RuntimeError('one-shot all-reduce transport unavailable')
        if transport in ('auto', 'rccl') and self.dist.get_backend(self.group) == 'nccl':
            uid = torch.zeros(128, dtype=torch.uint8, device=engine.device)
            if self.rank == 0:
                uid.copy_(torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8))
            self.dist.broadcast(uid, src=0, group=self.group)
            engine.comm_init(bytes(uid.cpu().numpy().tobytes()), self.world, self.rank)
            self.engine = engine
            return 'rccl'
        return False                                         # gloo without IPC: keep the torch.distributed callback path

    def _all_ok(self, ok, device):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def _attach_one_shot(self, engine):
        """Collective.  Every step's outcome is agreed on by all ranks, so either everybody ends up attached or nobody does."""
        side = engine.device if self.dist.get_backend(self.group) == 'nccl' else torch.device('cpu')
        if self.world > 8:
            return False
        try:
            blob, ok = engine.comm_ipc_export(), True
        except Exception as e:                               # noqa: BLE001 -- any failure just means "use the next transport"
            blob, ok, self.one_shot_error = bytes(engine.IPC_BLOB_BYTES), False, repr(e)
        mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(side)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine, group=self.group)
        if not self._all_ok(ok, side):
            return False
        blobs = b''.join(bytes(p.cpu().numpy().tobytes()) for p in parts)
        try:
            engine.comm_ipc_attach(blobs, self.world, self.rank)
        except Exception as e:                               # noqa: BLE001
            ok, self.one_shot_error = False, repr(e)
        if not self._all_ok(ok, side):
            if ok:
                engine.comm_ipc_detach()
            return False
        # test exchange with a short time limit: rank r contributes (r + 1) * [1, 2, ..., n]
        n = engine.P + 1
        try:
            engine.comm_set_timeout_ms(3000)
            ramp = torch.arange(1, n + 1, dtype=torch.float64, device=engine.device)
            probe = ramp * (self.rank + 1)
            self.dist.barrier(group=self.group)
            engine.allreduce_sum_(probe)
            engine.allreduce_sum_(probe)                     # both slot parities
            engine.comm_check()
            ok = bool(torch.equal(probe, ramp * (self.world * (self.world + 1) // 2) * self.world))
            engine.comm_set_timeout_ms(int(os.environ.get('METRPO_XCHG_TIMEOUT_MS', '20000')))
        except Exception as e:                               # noqa: BLE001
            ok, self.one_shot_error = False, repr(e)
        if not self._all_ok(ok, side):
            engine.comm_ipc_detach()
            return False
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
