"""CPU: the pieces of the process-per-GPU collective (pcl_comm_*, physicl_amd/comm.py) that need no GPU -- the library
finds librccl and reports its failure as an error (never a silent "world of one"), and the unique id reaches every rank
through the exchange function.  The all-reduce itself runs on the GPU box (tests/test_gpu_comm.py)."""
import os
import subprocess
import sys
from ctypes import create_string_buffer

import numpy as np
import pytest

from physicl_amd import _hip
from physicl_amd.comm import ID_BYTES, NativeCounterComm, file_exchange

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_without_a_gpu_the_id_request_fails_loudly_and_names_rccl():
    lib = _hip.load()
    buf = create_string_buffer(ID_BYTES)
    rc = lib.pcl_comm_unique_id(buf)
    if rc == 0:                                   # (a GPU box: the id is there)
        assert any(buf.raw)
        return
    msg = lib.pcl_last_error().decode()
    assert rc == -3 and "RCCL" in msg and "no HIP device" in msg, msg      # PCL_ERR_STATE, said before librccl is even loaded
    with pytest.raises(_hip.HipError):
        _hip.check(rc)


def test_arguments_are_checked_before_anything_is_loaded():
    lib = _hip.load()
    assert lib.pcl_comm_unique_id(None) == -2
    assert lib.pcl_comm_allreduce_sum_i64(None, None, 0) == -2 and lib.pcl_comm_destroy(None) == 0
    with pytest.raises(ValueError):
        NativeCounterComm(2, 2, exchange=lambda x: x)
    c = NativeCounterComm(1, 4, exchange=lambda x: x)
    assert c.shard(10) == (2, 5) and c.info()["backend"] == "rccl-native" and c.world == 4
    with pytest.raises(RuntimeError):
        c.allreduce_sum([1, 2])                   # no communicator yet: never a local "sum"


def test_the_id_reaches_every_rank_through_a_file(tmp_path):
    """Three processes, rank 0 publishes 128 bytes, the others wait for the file: what a launcher without torch would do
    between pcl_comm_unique_id and pcl_comm_create."""
    path = str(tmp_path / "comm.id")
    code = r"""
import sys
sys.path.insert(0, %r)
from physicl_amd.comm import file_exchange
rank = int(sys.argv[1])
mine = bytes(range(128)) if rank == 0 else None
got = file_exchange(%r, rank, timeout_s=30)(mine)
print(got.hex())
""" % (ROOT, path)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, text=True) for r in (2, 1, 0)]
    outs = [p.communicate(timeout=60)[0].strip() for p in procs]
    assert all(p.returncode == 0 for p in procs) and outs[0] == outs[1] == outs[2] == bytes(range(128)).hex()
    # a file that is still short is not taken for the id
    short = str(tmp_path / "short.id")
    open(short, "wb").write(b"x" * 17)
    with pytest.raises(TimeoutError):
        file_exchange(short, 1, timeout_s=0.2)(None)


def test_allreduce_max_is_built_from_the_sum():
    class Fake(NativeCounterComm):
        def __init__(self, rank, world, others):
            super().__init__(rank, world, exchange=lambda x: x)
            self.others = others

        def allreduce_sum(self, values):
            return np.asarray(values, dtype=np.int64) + self.others
    c = Fake(1, 3, np.array([2_500_000, 0, 1_000_000]))
    assert abs(c.allreduce_max(1.75) - 2.5) < 1e-9


def test_knobs_are_named_pcl_and_settable_without_a_gpu():
    """pcl_set_knob only records a value (read at the next call of the path it belongs to): no device needed."""
    lib = _hip.load()
    assert lib.pcl_set_knob(b"LD_PRELOAD", b"x") == -2 and lib.pcl_set_knob(None, b"1") == -2
    _hip.set_knob("PCL_AHEAD_K", 5)
    _hip.set_knob("PCL_AHEAD_K", None)
    with pytest.raises(_hip.HipError):
        _hip.set_knob("NOT_A_KNOB", "1")
