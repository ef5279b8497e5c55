"""Multi-GPU layer: one process per GPU, particles sharded by global index, and ONE collective --
a sum all-reduce of the small int64 counter vector (alive, hits, sign counts, plane crossings).

The reference has no distributed code at all; photons do not interact
(README.md:11), so a shard never needs another shard's particles: no halo, no migration.  The only
global quantities are the counters that exit conditions and measure steps read
(``len(sim.objects)``, physicl/__init__.py:414; rows of light.py:374-431).  Backend "nccl" is RCCL
over xGMI on ROCm; "gloo" runs the same code on CPU tensors (tests, rehearsals on one GPU).  torch is
imported only when world_size > 1.

Backend "nccl" is strict: if the RCCL group cannot be created, or its start-up all-reduce does not
see every rank, construction raises on EVERY rank -- a throughput number must never hide a
collective that silently ran somewhere else.  Ask for ``backend="gloo"`` explicitly to rehearse.
"""
import os

import numpy as np


def shard_range(n_global, rank, world):
    """Contiguous block [lo, hi) of global particle ids owned by ``rank``."""
    n_global, rank, world = int(n_global), int(rank), int(world)
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return (n_global * rank) // world, (n_global * (rank + 1)) // world


def _pci_of(torch, index):
    """PCI address of torch's device ``index`` ("0000:05:00.0"), or "?"."""
    try:
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:                           # noqa: BLE001 -- diagnostics only
        return "?"


class CollectiveError(RuntimeError):
    """The requested collective backend could not be brought up on every rank."""


class CounterComm:
    """Sum/max all-reduce of tiny host vectors across the ranks of one node."""

    def __init__(self, rank=0, world=1, backend="nccl", local_rank=0, _init=True, device_index=None):
        if backend not in ("nccl", "gloo"):
            raise ValueError("backend must be 'nccl' (RCCL) or 'gloo', not %r" % (backend,))
        self.rank, self.world, self.backend, self.local_rank = int(rank), int(world), backend, int(local_rank)
        self.device_index = self.local_rank if device_index is None else int(device_index)
        self.ranks_seen = 1            # what the start-up all-reduce of ones returned on the data-path group
        self.rccl_version = None
        self._dist = None
        self._torch = None
        self._dev = None
        self._group = None
        self._buf = {}                 # element count -> (pinned host tensor, device tensor)
        if self.world > 1 and _init:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            if backend == "nccl":
                if not torch.cuda.is_available():
                    raise CollectiveError("backend 'nccl' (RCCL) needs a GPU; use backend='gloo' for CPU runs")
                torch.cuda.set_device(self.device_index)
                self._dev = torch.device("cuda", self.device_index)
            else:
                self._dev = torch.device("cpu")
            if not dist.is_initialized():
                self._init_group(backend)

    def _init_group(self, backend):
        """Control plane = a gloo process group (rendezvous, barriers, the max of the timings).  For backend
        "nccl" the counter all-reduce runs on an RCCL group created on top of it and proven with a one-element
        all-reduce; every rank then learns over gloo whether ALL ranks succeeded, and all raise together if not."""
        import datetime
        torch, dist = self._torch, self._dist
        dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world,
                                timeout=datetime.timedelta(seconds=300))
        self._group = None                      # None = the default (gloo) group
        if backend != "nccl":
            probe = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(probe)
            self.ranks_seen = int(probe[0])
            return
        ok, why, g = 0, "", None
        # the first RCCL communicator of the process: let the library say why it fails, should it fail (read at
        # communicator creation; a caller's own NCCL_DEBUG wins)
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        try:
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            probe = torch.ones(1, dtype=torch.int64, device=self._dev)
            dist.all_reduce(probe, group=g)
            self.ranks_seen = int(probe.cpu()[0])
            ok = int(self.ranks_seen == self.world)
            if not ok:
                why = "the RCCL all-reduce of ones returned %d, expected %d" % (self.ranks_seen, self.world)
            try:
                v = torch.cuda.nccl.version()
                self.rccl_version = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
            except Exception:                   # noqa: BLE001 -- a version string is not worth failing for
                self.rccl_version = None
        except Exception as e:                  # noqa: BLE001 -- any RCCL bring-up failure
            why = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:300] if str(e) else "")
        agree = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)          # every rank takes the same decision
        if int(agree[0]) != 1:
            # one failure report that carries EVERY rank's view (over the gloo control plane, which works): which device
            # and PCI function each rank drove, whether its own bring-up succeeded, and the first line of its error --
            # the first real multi-GPU run must not need a second run to be understood
            mine = {"rank": self.rank, "device": self.device_index, "pci": _pci_of(torch, self.device_index), "ok": bool(ok),
                    "ranks_seen": self.ranks_seen, "error": why}
            try:
                views = [None] * self.world
                dist.all_gather_object(views, mine)
            except Exception:                   # noqa: BLE001 -- the report is best effort
                views = [mine]
            dist.destroy_process_group()
            lines = ["  rank %(rank)d: device %(device)s pci %(pci)s ok=%(ok)s ranks_seen=%(ranks_seen)s %(error)s" % v
                     for v in views if v]
            raise CollectiveError("RCCL process group could not be brought up on every rank (rank %d of %d, device %d%s); "
                                  "pass backend='gloo' explicitly to rehearse without RCCL\n%s\n  (NCCL_DEBUG=%s; RCCL's own "
                                  "messages are on stderr above)"
                                  % (self.rank, self.world, self.device_index, ": " + why if why else "", "\n".join(lines),
                                     os.environ.get("NCCL_DEBUG")))
        self._group = g

    @classmethod
    def from_env(cls, backend="nccl", device_index=None):
        """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run (or by
        ``bench.py --gpus N`` when it starts its own ranks)."""
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), backend,
                   int(os.environ.get("LOCAL_RANK", "0")), device_index=device_index)

    def info(self):
        """What the bench line reports under "collective"."""
        return {"backend": self.backend if self.world > 1 else None, "ranks_seen": self.ranks_seen,
                "rccl_version": self.rccl_version, "world": self.world}

    def shard(self, n_global):
        return shard_range(n_global, self.rank, self.world)

    def _buffers(self, n):
        """One pinned host tensor + one device tensor per vector length, reused by every all-reduce."""
        b = self._buf.get(n)
        if b is None:
            torch = self._torch
            host = torch.empty(n, dtype=torch.int64)
            if self._dev.type == "cuda":
                host = host.pin_memory()
            b = self._buf[n] = (host, torch.empty(n, dtype=torch.int64, device=self._dev) if self._dev.type == "cuda" else host)
        return b

    def allreduce_sum(self, values):
        """values: int64 array-like (the counter vector).  Returns the element-wise sum over ranks."""
        a = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
        if self.world == 1:
            return a.copy()
        host, dev = self._buffers(a.size)
        host.numpy()[:] = a
        if dev is not host:
            dev.copy_(host, non_blocking=True)
        self._dist.all_reduce(dev, op=self._dist.ReduceOp.SUM, group=self._group)
        if dev is not host:
            host.copy_(dev)                      # synchronises with the collective's stream
        return host.numpy().copy()

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

    def allgather_object(self, obj):
        """Every rank's small Python object, in rank order (control plane)."""
        if self.world == 1:
            return [obj]
        parts = [None] * self.world
        self._dist.all_gather_object(parts, obj)
        return parts

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
