"""CPU, world_size 2, gloo: the N>1 path -- index sharding + the counter all-reduce -- gives the
same global counters and the same per-photon results as one unsharded run.  Per-shard compute in
this CPU test is done by the oracle (test infrastructure); on GPUs it is the HIP step (the same
property is tested on one GPU in test_gpu_parity.py::test_results_do_not_depend_on_sharding)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from physicl_amd.dist import CounterComm, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import physicl_oracle as orc
from physicl_amd.dist import CounterComm
C, H = 299792458.0, 6.62607015e-34
comm = CounterComm.from_env(backend="gloo")
N, seed = 30001, 17
lo, hi = comm.shard(N)
ids = np.arange(lo, hi, dtype=np.int64)
n = hi - lo
st = {"r": [np.zeros(n)] * 3, "v": [np.full(n, C), np.zeros(n), np.zeros(n)], "dr": [np.zeros(n)] * 3,
      "dv": [np.zeros(n)] * 3, "E": orc.philox_energy(seed, ids, 2.8e-19, 9.9e-19), "id": ids}
rows, locals_ = [], []
for step in range(4):
    orc.step_newton(st, 1e-3)
    hit = orc.step_scatter_isotropic(st, orc.philox_draws(seed, step, st["id"]), 1e-3, 1e-3, C)
    orc.step_scatter_delete(st, orc.philox_draws(seed, 100 + step, st["id"])[2], 5e-4, 1e-3)
    local = [len(st["id"]), int(hit.sum())] + list(orc.sign_counts(st["v"])) + \
            [orc.plane_crossings(st["r"], st["dr"], [6e5, np.nan, np.nan])]
    rows.append(comm.allreduce_sum(local).tolist())
    locals_.append(local)
# K steps per launch: the K x counters of a launch go through ONE all-reduce (bench.py run_multi, Simulation._run_multi)
batched = comm.allreduce_sum(np.array(locals_, dtype=np.int64).reshape(-1)).reshape(4, -1).tolist()
gathered = comm.allgather_concat(np.arange(comm.rank + 2, dtype=np.float64) + 10 * comm.rank).tolist()
tmax = comm.allreduce_max(float(comm.rank))
comm.barrier()
print(json.dumps({"rank": comm.rank, "rows": rows, "batched": batched, "gathered": gathered, "ids": st["id"].tolist(), "tmax": tmax}))
comm.close()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    return sorted(outs, key=lambda d: d["rank"])


def test_shard_ranges_partition_the_ids():
    for n in (0, 1, 7, 8, 100_000_001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_single_rank_comm_is_identity():
    c = CounterComm(0, 1)
    assert c.allreduce_sum([1, 2, 3]).tolist() == [1, 2, 3] and c.allreduce_max(2.5) == 2.5
    c.barrier(), c.device_synchronize(), c.close()
    assert c.shard(10) == (0, 10)


def test_world2_gloo_counters_equal_unsharded_run():
    one = run_world(1)[0]
    two = run_world(2)
    assert two[0]["rows"] == two[1]["rows"] == one["rows"]          # every rank sees the global counters
    assert two[0]["batched"] == two[1]["batched"] == one["rows"]     # ... also when K steps' rows are reduced at once
    assert two[0]["ids"] + two[1]["ids"] == one["ids"]               # survivors: concatenation of the shards
    assert two[0]["tmax"] == two[1]["tmax"] == 1.0
    assert two[0]["gathered"] == two[1]["gathered"] == [0.0, 1.0, 10.0, 11.0, 12.0] and one["gathered"] == [0.0, 1.0]
    assert one["rows"][-1][0] == len(one["ids"]) > 0
