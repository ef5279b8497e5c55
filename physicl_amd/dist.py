"""Multi-GPU layer: one process per GPU, particles sharded by global index, and ONE collective --
a sum all-reduce of the small int64 counter vector (alive, hits, sign counts, plane crossings).

The reference has no distributed code at all; photons do not interact
(README.md:11), so a shard never needs another shard's particles: no halo, no migration.  The only
global quantities are the counters that exit conditions and measure steps read
(``len(sim.objects)``, physicl/__init__.py:414; rows of light.py:374-431).  Backend "nccl" is RCCL
over xGMI on ROCm; "gloo" runs the same code on CPU tensors (tests).  torch is imported only when
world_size > 1.
"""
import os

import numpy as np


def shard_range(n_global, rank, world):
    """Contiguous block [lo, hi) of global particle ids owned by ``rank``."""
    n_global, rank, world = int(n_global), int(rank), int(world)
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return (n_global * rank) // world, (n_global * (rank + 1)) // world


class CounterComm:
    """Sum/max all-reduce of tiny host vectors across the ranks of one node."""

    def __init__(self, rank=0, world=1, backend="nccl", local_rank=0, _init=True):
        self.rank, self.world, self.backend, self.local_rank = int(rank), int(world), backend, int(local_rank)
        self._dist = None
        self._torch = None
        self._dev = None
        if self.world > 1 and _init:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            if backend == "nccl":
                if not torch.cuda.is_available():
                    raise RuntimeError("backend 'nccl' (RCCL) needs a GPU; use backend='gloo' for CPU runs")
                torch.cuda.set_device(self.local_rank)
                self._dev = torch.device("cuda", self.local_rank)
            else:
                self._dev = torch.device("cpu")
            if not dist.is_initialized():
                kw = {"device_id": self._dev} if backend == "nccl" else {}
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)

    @classmethod
    def from_env(cls, backend="nccl"):
        """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run."""
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), backend,
                   int(os.environ.get("LOCAL_RANK", "0")))

    def shard(self, n_global):
        return shard_range(n_global, self.rank, self.world)

    def allreduce_sum(self, values):
        """values: int64 array-like (the counter vector).  Returns the element-wise sum over ranks."""
        a = np.ascontiguousarray(values, dtype=np.int64)
        if self.world == 1:
            return a.copy()
        t = self._torch.from_numpy(a.copy()).to(self._dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allreduce_max(self, x):
        if self.world == 1:
            return float(x)
        t = self._torch.tensor([float(x)], dtype=self._torch.float64, device=self._dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.cpu()[0])

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()

    def device_synchronize(self):
        if self.world > 1 and self.backend == "nccl":
            self._torch.cuda.synchronize()

    def close(self):
        if self.world > 1 and self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
