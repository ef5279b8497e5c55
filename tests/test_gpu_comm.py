"""GPU: pcl_comm_* -- RCCL loaded by the library itself, one int64 sum all-reduce on the context's stream (the collective
a host with one process per GPU and no torch uses).  The pool's boxes have one GPU: the communicator is
brought up with a world of one here (identity), two ranks on ONE device must be refused by RCCL as an error (no hang, no
silent fallback), and a Simulation driven through it gives the rows of the plain run."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_of_one_is_the_identity_and_says_which_rccl_it_is():
    from physicl_amd import _hip as hip
    from physicl_amd.comm import NativeCounterComm
    with hip.Device(0) as d:
        c = NativeCounterComm(0, 1, exchange=lambda ident: ident, device=d)
        info = c.info()
        assert info["ranks_seen"] == 1 and info["world"] == 1 and info["rccl_version"] and info["backend"] == "rccl-native"
        v = np.array([5, -3, 2 ** 40, 0, 7], dtype=np.int64)
        assert np.array_equal(c.allreduce_sum(v), v) and np.array_equal(c.allreduce_sum(np.arange(2048)), np.arange(2048))
        with pytest.raises(hip.HipError):
            c.allreduce_sum(np.arange(2049))                      # more than a launch's rows: refused, not truncated
        assert c.allreduce_max(0.25) == 0.25
        c.barrier()
        c.close()


def test_simulation_through_the_native_collective_gives_the_plain_rows():
    import physicl_amd as phys
    import physicl_amd.light as light
    import physicl_amd.newton as newton
    from physicl_amd.comm import NativeCounterComm
    rows = {}
    for how in ("plain", "native"):
        comm = NativeCounterComm(0, 1, exchange=lambda ident: ident) if how == "native" else None
        sim = phys.Simulation(exit=lambda s: len(s.ts) >= 12 or len(s.objects) == 0, seed=3, comm=comm)
        sim.add_objs(light.generate_photons_bulk(50_000, min=1.0, max=2.0, seed=3))
        sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
        sim.add_step(1, newton.NewtonianKinematicsStep())
        sim.add_step(2, light.ScatterDeleteStep(np.double(0.001), np.double(0.0007)))
        m = light.ScatterMeasureStep(None, True, [[2.0e5, np.nan, np.nan]])
        sim.add_step(3, m)
        sim.start()
        sim.join()
        assert sim.error is None
        rows[how] = [np.asarray(r).tolist() for r in m.data]
        if comm is not None:
            assert comm.info()["ranks_seen"] == 1 and comm.rccl_version
            comm.close()
        sim.close(download=False)
    assert rows["plain"] == rows["native"] and len(rows["plain"]) == 12


def test_two_ranks_on_one_device_are_refused_by_rccl_not_summed_locally(tmp_path):
    """What the one-GPU box can show of the >= 2-rank path: both ranks load librccl, rank 0's id reaches rank 1 through a
    file, both call pcl_comm_create -- and RCCL refuses a communicator with two ranks on the same GPU.  Each rank must
    come back with an error (HipError), never with a communicator."""
    code = r"""
import sys
sys.path.insert(0, %r)
from physicl_amd import _hip as hip
from physicl_amd.comm import NativeCounterComm
rank = int(sys.argv[1])
d = hip.Device(0)
try:
    NativeCounterComm(rank, 2, exchange=%r, device=d)
    print("CREATED")
except hip.HipError as e:
    print("REFUSED", str(e).splitlines()[0][:200])
""" % (ROOT, str(tmp_path / "id"))
    env = dict(os.environ, NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
             for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT")
    assert all("REFUSED" in o and "CREATED" not in o for o in outs), outs
