"""The counters' collective WITHOUT torch: ``pcl_comm_*`` of the C ABI (RCCL loaded by the library at run time).

``physicl_amd.dist.CounterComm`` goes through ``torch.distributed``; a host that is not Python -- or a Python host that
does not want torch in its ranks -- runs one process per GPU and sums the counter rows with the library's own entry
points (include/physicl_hip.h, "the counters' collective"): rank 0 makes a unique id, every rank creates the
communicator from it, one int64 sum all-reduce per launch on the context's stream.  ``NativeCounterComm`` is that, shaped
like ``CounterComm`` so that ``Simulation(comm=...)`` takes it.  The id travels through ``exchange``: a function
``exchange(id_bytes_or_None) -> id_bytes`` (rank 0 is handed its fresh id and must publish it, the others are handed
None and must return what rank 0 published), or a file path (``file_exchange``: rank 0 writes the file atomically, the
others wait for it) -- whatever the host's launcher has.

No fallback: if librccl cannot be loaded or the bring-up fails, the constructor / ``attach`` raises on that rank
(``HipError``); the launcher takes the job down (physicl_amd/launch.py does).
"""
import os
import time
from ctypes import byref, c_int, c_int64, c_void_p, create_string_buffer

import numpy as np

from .dist import shard_range

ID_BYTES = 128


def file_exchange(path, rank, timeout_s=120.0):
    """``exchange`` through a file: rank 0 writes ``path`` (temporary name + rename: never seen half-written), the other
    ranks wait for it to appear."""
    def exchange(mine):
        if rank == 0:
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "wb") as f:
                f.write(mine)
            os.replace(tmp, path)
            return mine
        t_end = time.time() + timeout_s
        while time.time() < t_end:
            try:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == ID_BYTES:
                    return data
            except OSError:
                pass
            time.sleep(0.01)
        raise TimeoutError("rank %d: no communicator id at %s after %.0f s" % (rank, path, timeout_s))
    return exchange


class NativeCounterComm:
    """Sum all-reduce of the int64 counter vector over the ranks of one node, RCCL through ``pcl_comm_*``."""
    backend = "rccl-native"

    def __init__(self, rank, world, exchange, local_rank=None, device=None):
        self.rank, self.world = int(rank), int(world)
        if not (0 <= self.rank < self.world):
            raise ValueError("rank %d outside world of %d" % (self.rank, self.world))
        self.local_rank = self.rank if local_rank is None else int(local_rank)
        self._exchange = file_exchange(exchange, self.rank) if isinstance(exchange, (str, os.PathLike)) else exchange
        self._comm, self._dev, self._lib = None, None, None
        self.ranks_seen, self.rccl_version = 1, None
        if device is not None:
            self.attach(device)

    # -- bring-up ----------------------------------------------------------------------------------------------------
    def attach(self, device):
        """Create the communicator on ``device``'s context (a ``_hip.Device``); Simulation calls this when it opens its
        device.  Collective: returns when every rank has arrived and a one-element all-reduce has seen all of them."""
        from . import _hip
        if self._comm is not None:
            return
        self._dev, self._lib = device, _hip.load()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        ident = None
        if self.rank == 0:
            buf = create_string_buffer(ID_BYTES)
            _hip.check(self._lib.pcl_comm_unique_id(buf))
            ident = buf.raw
        ident = self._exchange(ident)
        if not isinstance(ident, (bytes, bytearray)) or len(ident) != ID_BYTES:
            raise ValueError("exchange() must return the %d bytes rank 0 published" % ID_BYTES)
        comm = c_void_p()
        _hip.check(self._lib.pcl_comm_create(device.ctx, bytes(ident), self.rank, self.world, byref(comm)))
        self._comm = comm
        r, w, v, n = c_int(), c_int(), c_int(), c_int64()
        _hip.check(self._lib.pcl_comm_info(comm, byref(r), byref(w), byref(v), byref(n)))
        self.ranks_seen = w.value                                     # (pcl_comm_create fails unless its probe saw them all)
        self.rccl_version = "%d.%d.%d" % (v.value // 10000, v.value // 100 % 100, v.value % 100) if v.value else None

    # -- what Simulation and bench.py use ------------------------------------------------------------------------------
    def shard(self, n_global):
        return shard_range(n_global, self.rank, self.world)

    def info(self):
        return {"backend": self.backend, "ranks_seen": self.ranks_seen, "rccl_version": self.rccl_version, "world": self.world}

    def allreduce_sum(self, values):
        a = np.ascontiguousarray(values, dtype=np.int64).reshape(-1).copy()
        if self._comm is None:
            raise RuntimeError("NativeCounterComm: attach(device) first (Simulation does it when it opens its device)")
        from . import _hip
        _hip.check(self._lib.pcl_comm_allreduce_sum_i64(self._comm, a.ctypes.data, int(a.size)))
        return a

    def allreduce_max(self, x):
        """max over ranks of a non-negative time in seconds (microsecond resolution): one slot per rank, summed."""
        v = np.zeros(self.world, dtype=np.int64)
        v[self.rank] = int(round(float(x) * 1e6))
        return float(self.allreduce_sum(v).max()) * 1e-6

    def barrier(self):
        self.allreduce_sum([1])

    def device_synchronize(self):
        if self._dev is not None:
            self._dev.sync()

    def allgather_concat(self, values):
        raise NotImplementedError("NativeCounterComm carries the counters only; use physicl_amd.dist.CounterComm for measure_E lists")

    def close(self):
        if self._comm is not None:
            self._lib.pcl_comm_destroy(self._comm)
            self._comm = None
