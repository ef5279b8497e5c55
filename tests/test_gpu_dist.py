"""GPU, 2 processes sharing device 0, gloo for the counter all-reduce: a sharded Simulation (index
shards, global-id-keyed RNG, all-reduced counters) produces exactly the rows of the single-process run.
(RCCL itself needs >= 2 GPUs; the driver's multi-GPU bench exercises it.  Everything else of the
N > 1 path -- sharding, id bases, the reduce of the counter vector, exit on the GLOBAL alive count --
runs here on real kernels.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import physicl as phys, physicl.light, physicl.newton
from physicl_amd.dist import CounterComm
comm = CounterComm.from_env(backend="gloo")
kind, spl = %(kind)r, %(spl)d
trace = kind.endswith("_trace")
kind = kind[:-6] if trace else kind
sim = phys.Simulation(cl_on=True, device=0, comm=comm if comm.world > 1 else None, seed=21, rng="philox", steps_per_launch=spl,
                      exit=(lambda s: len(s.objects) < 2000) if kind == "delete" else
                           (lambda s: len(s.objects) == 0) if kind == "delete_empty" else (lambda s: s.t >= 0.0055))
N = 150001
if kind in ("batch", "delete", "delete_empty", "mixed"):
    sim.add_objs(phys.light.generate_photons_bulk(N, min=phys.light.E_from_wavelength(700e-9),
                                                  max=phys.light.E_from_wavelength(200e-9), seed=21))
else:
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(3e-19 * (1 + 1e-4 * i)), uid=i)
                  for i in range(3001)])
sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
sim.add_step(1, phys.newton.NewtonianKinematicsStep())
if kind.startswith("delete"):
    sim.add_step(2, phys.light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
else:
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001), wavelength_dep_scattering=False))
m1 = phys.light.ScatterMeasureStep(None, True, [[6e5, np.nan, np.nan], [np.nan, 0.0, np.nan]], measure_E=(kind == "objects"))
m2 = phys.light.ScatterSignMeasureStep(None, True)
if kind == "mixed":          # BASELINE configs[4]'s loop with a measure step behind each light step
    sim.add_step(3, m2)
    sim.add_step(4, phys.newton.NewtonianKinematicsStep())
    sim.add_step(5, phys.light.ScatterDeleteStep(np.double(0.0002), np.double(0.001)))
    sim.add_step(6, m1)
else:
    sim.add_step(3, m1)
    sim.add_step(4, m2)
tp = None
if trace:        # a tracked subset that straddles the boundary between the two shards (ids 75000 | 75001)
    tp = phys.light.TracePathMeasureStep(None, trace_ids=list(range(200)) + list(range(74990, 75012)) + [N - 1], trace_dv=True)
    sim.add_step(9, tp)
sim.run()
flat = lambda r: [x if isinstance(x, list) else float(x) for x in r]       # measure_E rows carry energy lists
print(json.dumps({"rank": comm.rank, "m1": [flat(r) for r in m1.data], "m2": [[float(x) for x in r] for r in m2.data],
                  "alive": len(sim.objects), "hits": int(sim.hits), "local": int(sim._dev.count), "sched": dict(sim.schedule),
                  "trace": None if tp is None else [[r[0], int(r[1])] + [np.asarray(x, dtype=float).reshape(-1).tolist() for x in r[2:]]
                                                    for r in tp.data[1:]]}))
comm.close()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, kind, spl=1):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT, "kind": kind, "spl": spl}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    return sorted(outs, key=lambda d: d["rank"])


@pytest.mark.parametrize("kind", ["batch", "objects", "delete"])
def test_two_shards_reproduce_the_single_process_rows(kind):
    one = run_world(1, kind)[0]
    two = run_world(2, kind)
    for rank in two:
        assert rank["m1"] == one["m1"] and rank["m2"] == one["m2"]       # every rank records the GLOBAL rows
        assert rank["alive"] == one["alive"] and rank["hits"] == one["hits"]
    assert two[0]["local"] + two[1]["local"] == one["local"]
    assert len(one["m1"]) >= 5 and one["m1"][0][1] > 0
    if kind == "objects":          # measure_E: the crossing photons' energies, gathered across the shards in object order
        lists = [x for r in one["m1"] for x in r if isinstance(x, list) and x]
        assert lists and all(len(set(l)) == len(l) for l in lists)


@pytest.mark.parametrize("kind", ["batch", "delete_empty", "mixed"])
def test_two_shards_with_several_passes_per_launch_reproduce_the_single_process_rows(kind):
    """Simulation(steps_per_launch=4) on two shards == one pass per launch on one process: the K x counters of a
    launch are all-reduced at once, and the cut at the pass that empties the GLOBAL store is taken by both ranks."""
    one = run_world(1, kind, 1)[0]
    two = run_world(2, kind, 4)
    for rank in two:
        assert rank["m1"] == one["m1"] and rank["m2"] == one["m2"]
        assert rank["alive"] == one["alive"] and rank["hits"] == one["hits"]
    assert two[0]["local"] + two[1]["local"] == one["local"]
    assert len(one["m1"]) >= 5


@pytest.mark.parametrize("kind", ["batch_trace", "delete_empty_trace", "mixed_trace"])
def test_two_shards_trace_the_tracked_subset_of_the_single_process_run(kind):
    """TracePathMeasureStep on a sharded run: every rank works the tracked ids it holds out on the device, ahead of its K-pass
    launches, and the table is put together with one int64 sum all-reduce at the end -- the single-process table, on every
    rank (NaN == NaN: json carries them)."""
    one = run_world(1, kind, 1)[0]
    two = run_world(2, kind, 4)
    same = lambda a, b: json.dumps(a) == json.dumps(b)               # (NaN-safe comparison of the nested lists)
    assert len(one["trace"]) >= 100 and any(len(r) > 3 for r in one["trace"])
    for rank in two:
        assert same(rank["trace"], one["trace"])
        assert rank["m2"] == one["m2"]
        assert any(k.endswith("_multi") for k in rank["sched"])      # the K-passes-per-launch schedule stayed
    if kind != "batch_trace":                                        # removed photons: their lists end early
        assert len({len(r) for r in one["trace"]}) > 1 or any("nan" in json.dumps(r).lower() for r in one["trace"])


def test_rccl_group_comes_up_with_one_rank():
    """The RCCL leg of CounterComm (new_group("nccl") on top of the gloo control plane, the probe, an all-reduce of
    a K x 5 counter block on a device tensor) -- with the single rank a one-GPU box allows."""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "rccl_single_rank.py")],
                                  env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stderr=subprocess.STDOUT, timeout=300)
    text = out.decode()
    assert "backend after init: nccl" in text and "allreduce ok: [0 1 2 3 4] 12720" in text, text[-2000:]
