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

    def __init__(self, rank=0, world=1, backend="nccl", local_rank=0, _init=True, device_index=None):
        self.rank, self.world, self.backend, self.local_rank = int(rank), int(world), backend, int(local_rank)
        dev_index = self.local_rank if device_index is None else int(device_index)
        self._dist = None
        self._torch = None
        self._dev = None
        self._group = None
        if self.world > 1 and _init:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            if backend == "nccl":
                if not torch.cuda.is_available():
                    raise RuntimeError("backend 'nccl' (RCCL) needs a GPU; use backend='gloo' for CPU runs")
                torch.cuda.set_device(dev_index)
                self._dev = torch.device("cuda", dev_index)
            else:
                self._dev = torch.device("cpu")
            if not dist.is_initialized():
                self._init_group(backend)

    def _init_group(self, backend):
        """Control plane = a gloo process group (always comes up).  For backend "nccl" the counter all-reduce
        runs on an RCCL group created on top of it and proven with one tiny all-reduce; if RCCL cannot be
        brought up the counters stay on gloo (40 bytes per step: no effect on throughput) and ``self.backend``
        says so."""
        import datetime
        torch, dist = self._torch, self._dist
        dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world,
                                timeout=datetime.timedelta(seconds=300))
        self._group = None                      # None = the default (gloo) group
        if backend != "nccl":
            return
        ok = 0
        try:
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            probe = torch.ones(1, dtype=torch.int64, device=self._dev)
            dist.all_reduce(probe, group=g)
            ok = int(int(probe.cpu()[0]) == self.world)
        except Exception as e:                      # noqa: BLE001 -- any RCCL bring-up failure
            import sys
            print("physicl_amd.dist: RCCL unavailable (%s: %s); reducing the counters over gloo"
                  % (type(e).__name__, str(e).splitlines()[0][:200]), file=sys.stderr)
        # every rank must take the same decision
        agree = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree[0]) == 1:
            self._group = g
        else:
            if os.environ.get("PCL_NO_GLOO_FALLBACK"):
                raise RuntimeError("RCCL process group could not be created")
            self.backend, self._dev = "gloo", torch.device("cpu")

    @classmethod
    def from_env(cls, backend="nccl", device_index=None):
        """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run."""
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), backend,
                   int(os.environ.get("LOCAL_RANK", "0")), device_index=device_index)

    def shard(self, n_global):
        return shard_range(n_global, self.rank, self.world)

    def allreduce_sum(self, values):
        """values: int64 array-like (the counter vector).  Returns the element-wise sum over ranks."""
        a = np.ascontiguousarray(values, dtype=np.int64)
        if self.world == 1:
            return a.copy()
        t = self._torch.from_numpy(a.copy()).to(self._dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        return t.cpu().numpy()

    def allgather_concat(self, values):
        """Concatenation, in rank order, of every rank's 1-D array (variable lengths) -- the energy lists of
        ScatterMeasureStep(measure_E=True): shards are contiguous index blocks, so rank order is particle order.
        Goes over the gloo control plane (these lists are diagnostics, not a per-step hot path)."""
        a = np.ascontiguousarray(values)
        if self.world == 1:
            return a.copy()
        parts = [None] * self.world
        self._dist.all_gather_object(parts, a)                       # default (gloo) group
        return np.concatenate([np.asarray(p, dtype=a.dtype) for p in parts]) if parts else a

    def allreduce_max(self, x):
        if self.world == 1:
            return float(x)
        t = self._torch.tensor([float(x)], dtype=self._torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)          # control plane (gloo)
        return float(t[0])

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()

    def device_synchronize(self):
        if self.world > 1 and self.backend == "nccl":
            self._torch.cuda.synchronize()

    def close(self):
        if self.world > 1 and self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
