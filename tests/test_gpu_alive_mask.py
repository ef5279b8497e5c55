"""GPU: the delete loop body behind an alive mask (k_delete_alive, pcl_step_fused_delete) against the oracle.

The reference removes the flagged photons from ``sim.objects`` every pass (physicl/light.py:258-260); all that is
observable is that the survivors keep their order.  The device keeps the removed photons' slots, one kernel per body,
until fewer than half of the slots are alive, and lets r lag behind by up to 8 moves.  Everything observable must be
what the oracle's step-by-step chain gives: survivor ids (bit-exact), positions and velocities (bit-exact: IEEE mul / add
only), every measure row, the flags of each body -- whenever and however the store is looked at in between.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import physicl_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_LIT = 299792458.0


def cols(a):
    return [np.ascontiguousarray(a[:, k]) for k in range(3)]


def oracle_body(st, dt, A, n, seed, step, planes, dtype=np.float64):
    """One loop body on the oracle state: Newton, ScatterDelete, then the counters of the measure steps."""
    orc.step_newton(st, dt, dtype=dtype)
    before = len(st["id"])
    flags, keep = orc.step_scatter_delete(st, orc.philox_draws(seed, step, st["id"], dtype=dtype)[2], A, n, dtype=dtype)
    row = [len(keep), before - len(keep)] + [int(x) for x in orc.sign_counts(st["v"])]
    for loc in planes:                               # light.py:385-399 in the store's precision: prev = r - dr, rounded
        ax = [k for k in range(3) if not np.isnan(loc[k])][0]
        L, x = dtype(loc[ax]), np.asarray(st["r"][ax], dtype=dtype)
        prev = (x - np.asarray(st["dr"][ax], dtype=dtype)).astype(dtype)
        row.append(int(np.count_nonzero(((prev <= L) & (L <= x)) | ((prev >= L) & (L >= x)))))
    return flags, row


def device_row(o):
    return [o["N"], o["removed"]] + [int(x) for x in o["sign"]] + [int(x) for x in o["planes"]]


def assert_state(d, st, hip, what):
    s = d.download_state()
    assert np.array_equal(s["id"], st["id"]), what
    for f in ("r", "v", "dr"):
        for k in range(3):
            assert np.array_equal(s[f][k], st[f][k]), (what, f, k)
    assert np.array_equal(s["E"], st["E"]), what


@pytest.mark.parametrize("look", [True, False], ids=["flags_every_body", "rows_only"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("N,pdel", [(300_000, 0.3), (200_001, 0.04), (70_000, 0.55)])
def test_delete_bodies_on_the_alive_mask_vs_oracle(N, pdel, dtype, look):
    """Chains of bodies at three removal rates: 0.3 compacts every third body, 0.04 lets more than 8 moves pile up
    (k_apply_pending) before the first compaction, 0.55 compacts every second body.  Rows and flags are compared after
    every body (looking at the flags does not disturb the store), the whole state at a few points and at the end.
    ``rows_only``: nothing but the rows is looked at between the bodies -- the pattern the library answers from bodies
    worked out ahead (k_delete_ahead: these stores are below 2^20 slots), committed when the state is downloaded."""
    from physicl_amd import _hip as hip
    np_t = np.float64 if dtype == "f64" else np.float32
    rs = np.random.RandomState(N)
    direction = rs.normal(size=(N, 3))
    direction /= np.linalg.norm(direction, axis=1)[:, None]
    init = {"r": (rs.normal(size=(N, 3)) * 1e5).astype(np_t), "v": (direction * C_LIT).astype(np_t), "E": rs.uniform(1, 2, N).astype(np_t)}
    dt, n_k, seed = 1e-3, 1e-3, 17
    A_k = pdel / (n_k * C_LIT * dt)
    planes = [[2e5, np.nan, np.nan], [np.nan, -1e5, np.nan], [np.nan, np.nan, 0.0]]
    st = {"r": cols(init["r"]), "v": cols(init["v"]), "dr": [np.zeros(N, np_t)] * 3, "dv": [np.zeros(N, np_t)] * 3, "E": init["E"].copy(),
          "id": np.arange(N, dtype=np.int64) + 5}
    bodies = 14 if pdel < 0.1 else 9
    with hip.Device(0) as d:
        d.store_alloc(N, dtype)
        d.upload_state(dict(init, id_base=5))
        extents = []
        for step in range(bodies):
            before = d.count
            flags, row = oracle_body(st, dt, A_k, n_k, seed, step, planes, dtype=np_t)
            o = d.step_fused_delete(dt, A_k, n_k, hip.RNG_PHILOX, seed, step, planes, lazy=True)
            assert device_row(o) == row, (step, device_row(o), row)
            extents.append(d.slots)
            assert d.count == row[0] and d.slots >= d.count
            if look or step == bodies - 2:
                assert np.array_equal(d.last_delete_flags(before), flags), step
                assert d.slots == extents[-1]                                 # looking at the flags moved nothing
            if step in (3, bodies - 1):
                assert_state(d, st, hip, (step, "download"))
                assert d.slots == d.count                                     # ... a download compacts
        launches, served, missed = d.ahead_stats()
        if not look:
            # bodies came without a launch (the state download after body 3 cuts the first launch's rows short, which also
            # makes the library pause before it works ahead again)
            assert launches >= 1 and served > launches, (launches, served, missed)
        elif pdel < 0.1:      # nothing was compacted by the path itself for more than 8 bodies: the pending moves were flushed
            assert extents[4:13] == [extents[4]] * 9 and extents[4] > d.count
        elif pdel < 0.5:
            assert extents[0] == N and extents[1] == N and extents[2] < N    # the third body starts below 50 % and compacts
        else:
            assert extents[0] == N and extents[1] < N                         # the second body already does


@pytest.mark.parametrize("pattern", ["every_body", "runs_of_three"])
def test_changing_time_steps_and_the_run_length_list_of_pending_moves_vs_oracle(pattern):
    """r lags behind the bodies of an alive-mask store; the moves it is owed are kept as runs of equal dt (a loop repeats one
    dt: one entry however long the run).  A time step that changes with every body opens a run per body -- the list of eight
    fills and a body's kernel writes r back --, runs of three merge; positions, flags and rows must be the oracle's either way
    (low removal rate: the path never compacts by itself, so everything rides on the pending moves)."""
    from physicl_amd import _hip as hip
    N, n_k, seed = 150_000, 1e-3, 23
    rs = np.random.RandomState(7)
    direction = rs.normal(size=(N, 3))
    direction /= np.linalg.norm(direction, axis=1)[:, None]
    init = {"r": rs.normal(size=(N, 3)) * 1e5, "v": direction * C_LIT, "E": rs.uniform(1, 2, N)}
    planes = [[np.nan, 1e5, np.nan], [-2e5, np.nan, np.nan]]
    st = {"r": cols(init["r"]), "v": cols(init["v"]), "dr": [np.zeros(N)] * 3, "dv": [np.zeros(N)] * 3, "E": init["E"].copy(),
          "id": np.arange(N, dtype=np.int64)}
    dts = [1e-3 * (1.0 + 0.125 * ((k if pattern == "every_body" else k // 3) % 5)) for k in range(24)]
    with hip.Device(0) as d:
        d.store_alloc(N)
        d.upload_state(dict(init, id_base=0))
        for step, dt in enumerate(dts):
            before = d.count
            A_k = 0.03 / (n_k * C_LIT * dt)
            flags, row = oracle_body(st, dt, A_k, n_k, seed, step, planes)
            o = d.step_fused_delete(dt, A_k, n_k, hip.RNG_PHILOX, seed, step, planes, lazy=True)
            assert device_row(o) == row, (step, device_row(o), row)
            if step % 7 == 6:
                assert np.array_equal(d.last_delete_flags(before), flags), step
            if step in (10, 23):
                assert_state(d, st, hip, (pattern, step))


def test_alive_mask_chain_without_looking_equals_chain_with_looking():
    """Same bodies on two stores: one is downloaded after every body (dense every time), the other never until the end.
    Rows and final state must agree bit for bit; so must a third store on the round-2 pipeline (PCL_ALIVE=0)."""
    code = r"""
import json, os, sys
import numpy as np
from physicl_amd import _hip as hip
N, look = 400_000, sys.argv[1] == "look"
d = hip.Device(0); d.store_alloc(N); d.fill_photons(N, 1000, 299792458.0, 1.0, 2.0, 9)
rows = []
for step in range(12):
    o = d.step_fused_delete(1e-3, 1e-3, 0.8e-3, hip.RNG_PHILOX, 9, step, [[1.0e6, np.nan, np.nan]], lazy=True)
    rows.append([o["N"], o["removed"]] + [int(x) for x in o["sign"]] + [int(x) for x in o["planes"]])
    if look:
        d.download(hip.R0, 3)
s = d.download_state()
import hashlib
h = hashlib.sha256()
for f in ("r", "v", "dr", "dv"):
    for k in range(3):
        h.update(np.ascontiguousarray(s[f][k]).tobytes())
h.update(s["E"].tobytes()); h.update(s["id"].tobytes())
print("RESULT", json.dumps({"rows": rows, "sha": h.hexdigest(), "n": int(len(s["id"]))}))
"""
    outs = []
    for arg, env in (("look", {}), ("blind", {}), ("blind", {"PCL_ALIVE": "0"}), ("blind", {"PCL_ALIVE_RATIO": "0.9", "PCL_ALIVE_MIN_SLOTS": "0"})):
        out = subprocess.check_output([sys.executable, "-c", code, arg], cwd=ROOT, env=dict(os.environ, **env), timeout=600).decode()
        outs.append(out[out.index("RESULT") + 7:].strip())
    assert outs[0] == outs[1] == outs[2] == outs[3]
    assert '"n": 0' not in outs[0]


def test_alive_mask_interleaved_with_scatter_steps_and_multi_launches():
    """A delete body leaves the store behind its mask; a scatter step, a K-step launch or a K-body delete launch that
    follows must see the dense store (ids explicit from then on), and a delete body after THEM starts fresh."""
    from physicl_amd import _hip as hip
    N = 150_000
    sc = dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=6.62607015e-34, rng_mode=hip.RNG_PHILOX, seed=4)
    results = []
    for env_alive in ("1", "0"):
        code_rows = []
        # (same process for both: PCL_ALIVE is read once, so the comparison run uses explicit downloads to force density)
        with hip.Device(0) as d:
            d.store_alloc(N)
            d.fill_photons(N, 0, C_LIT, 1.0, 1.0, 4)
            for step in range(0, 12, 4):
                o = d.step_fused_delete(1e-3, 1e-3, 0.5e-3, hip.RNG_PHILOX, 4, step, [], lazy=True)
                if env_alive == "0":
                    d.download(hip.R0, 1)
                code_rows.append((o["N"], o["removed"], tuple(o["sign"])))
                assert not d.is_uniform()
                o = d.step_fused(1e-3, dict(sc, step=step + 1), (), lazy=True)
                code_rows.append((o["N"], o["hits"], tuple(o["sign"])))
                o = d.step_fused_delete(1e-3, 1e-3, 0.5e-3, hip.RNG_PHILOX, 4, step + 2, [], lazy=True)
                if env_alive == "0":
                    d.download(hip.R0, 1)
                code_rows.append((o["N"], o["removed"], tuple(o["sign"])))
                for o in d.step_fused_delete_multi(1e-3, 2, 1e-3, 0.3e-3, 4, step + 3, []):
                    code_rows.append((o["N"], o["removed"], tuple(o["sign"])))
            s = d.download_state()
        results.append((code_rows, s))
    (ra, sa), (rb, sb) = results
    assert ra == rb and len(sa["id"]) == len(sb["id"]) > 0
    assert np.array_equal(sa["id"], sb["id"])
    for f in ("r", "v", "dr", "dv"):
        for k in range(3):
            assert np.array_equal(sa[f][k], sb[f][k]), (f, k)


def _ahead_run(hip, N, dtype, script):
    """Run ``script`` (a list of actions) on a fresh store; returns (rows, final state, ahead statistics)."""
    rows = []
    with hip.Device(0) as d:
        d.store_alloc(N, dtype)
        d.fill_photons(N, 77, C_LIT, 1.0, 2.0, 21)
        step = 0
        for act in script:
            if act[0] == "delete":
                _, dt, pdel, planes = act
                o = d.step_fused_delete(dt, pdel / (1e-3 * C_LIT * dt), 1e-3, hip.RNG_PHILOX, 21, step, planes, lazy=True)
                rows.append(device_row(o) if planes is not None else [o["N"], o["removed"]])
            elif act[0] == "skip":                      # the caller's launch counter jumps (another kernel ran elsewhere)
                step += act[1] - 1
                continue
            elif act[0] == "iso":
                o = d.step_fused(1e-3, dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=6.62607015e-34, rng_mode=hip.RNG_PHILOX, seed=21, step=step), (), lazy=True)
                rows.append([o["N"], o["hits"]])
            elif act[0] == "flags":
                rows.append(d.last_delete_flags(act[1] if act[1] else rows[-1][0] + rows[-1][1]).tolist())
            elif act[0] == "peek":
                rows.append([d.count, float(d.download(hip.R0, 1)[0])])
            step += 1
        s = d.download_state()
        stats = d.ahead_stats()
    return rows, s, stats


PL1 = [[3.0e5, np.nan, np.nan]]
PL3 = [[3.0e5, np.nan, np.nan], [np.nan, 0.0, np.nan], [np.nan, np.nan, -2.0e5]]
AHEAD_SCRIPTS = {
    "until_empty": [("delete", 1e-3, 0.3, PL1)] * 60,
    "no_measure": [("delete", 1e-3, 0.25, None)] * 40,
    "three_planes_then_flags": [("delete", 1e-3, 0.2, PL3)] * 11 + [("flags", 0)] + [("delete", 1e-3, 0.2, PL3)] * 9,
    "dt_changes": [("delete", 1e-3, 0.2, PL1)] * 7 + [("delete", 2e-3, 0.2, PL1)] * 9 + [("delete", 1e-3, 0.1, PL3)] * 5,
    "launch_counter_jumps": [("delete", 1e-3, 0.2, PL1)] * 6 + [("skip", 3)] + [("delete", 1e-3, 0.2, PL1)] * 8,
    "scatter_in_between": [("delete", 1e-3, 0.15, PL1)] * 5 + [("iso",)] + [("delete", 1e-3, 0.15, PL1)] * 6 + [("iso",)] * 2 + [("delete", 1e-3, 0.15, PL1)] * 4,
    "peeks": [("delete", 1e-3, 0.1, PL1)] * 4 + [("peek",)] + [("delete", 1e-3, 0.1, PL1)] * 20 + [("peek",)] + [("delete", 1e-3, 0.1, PL1)] * 3,
}


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", sorted(AHEAD_SCRIPTS))
def test_bodies_worked_out_ahead_change_nothing_but_the_number_of_launches(name, dtype):
    """pcl_step_fused_delete on a small store answers a run's loop from K bodies worked out in one launch (k_delete_ahead)
    and makes the state real when anything else is asked (k_ahead_commit).  Every script -- a run until the store is empty,
    calls that stop matching the prediction (another dt, other planes, a jump of the launch counter), scatter steps, flag
    and position reads in between -- must give the rows, flags and final state of the same script with PCL_AHEAD=0, bit
    for bit; the oracle pins the plain path in the tests above."""
    from physicl_amd import _hip as hip
    N = 260_000 if name != "until_empty" else 90_000
    try:
        hip.set_knob("PCL_AHEAD", "0")
        rows0, s0, st0 = _ahead_run(hip, N, dtype, AHEAD_SCRIPTS[name])
        hip.set_knob("PCL_AHEAD", None)
        hip.set_knob("PCL_AHEAD_K", "5" if name == "peeks" else None)
        rows1, s1, st1 = _ahead_run(hip, N, dtype, AHEAD_SCRIPTS[name])
        # ... and in the form big stores take (above PCL_AHEAD_MAX_SLOTS): the bodies up to the one that compacts, r left behind
        # at the commit, the compaction run from the committed masks
        hip.set_knob("PCL_AHEAD_MAX_SLOTS", "0")
        hip.set_knob("PCL_ALIVE_MIN_SLOTS", "0")
        rows2, s2, st2 = _ahead_run(hip, N, dtype, AHEAD_SCRIPTS[name])
        # ... and with the kernel that gives every slot its own lane also where the one that lists the alive photons would run
        hip.set_knob("PCL_AHEAD_MAX_SLOTS", None)
        hip.set_knob("PCL_ALIVE_MIN_SLOTS", None)
        hip.set_knob("PCL_AHEAD_LIVE", "0")
        rows3, s3, st3 = _ahead_run(hip, N, dtype, AHEAD_SCRIPTS[name])
    finally:
        for k in ("PCL_AHEAD", "PCL_AHEAD_K", "PCL_AHEAD_MAX_SLOTS", "PCL_ALIVE_MIN_SLOTS", "PCL_AHEAD_LIVE"):
            hip.set_knob(k, None)
    assert st0 == (0, 0, 0) and st1[0] >= 1 and st1[1] > st1[0] and st2[0] >= 1 and st2[1] > st2[0] and st3 == st1, (st0, st1, st2, st3)
    assert rows0 == rows1 == rows2 == rows3
    for sx in (s1, s2, s3):
        assert np.array_equal(s0["id"], sx["id"]) and np.array_equal(s0["E"], sx["E"])
        for f in ("r", "v", "dr", "dv"):
            for k in range(3):
                assert np.array_equal(s0[f][k], sx[f][k]), (f, k)
    if name == "until_empty":
        assert rows1[-1][0] == 0 and len(s1["id"]) == 0


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("K,planes", [(5, PL1), (40, PL1), (64, None), (17, PL3)])
def test_a_k_body_call_takes_the_single_calls_path_with_k_as_a_promise(K, planes, dtype):
    """pcl_step_fused_delete_multi on an all-photon store: the bodies one by one through pcl_step_fused_delete's code, the
    number still to come handed to it as a promise -- one k_delete_ahead(_live) launch per as many bodies as a launch holds
    (never a launch per body, never a pause), the rest answered from its rows.  Rows and final state equal K single calls
    with the bodies-ahead path off, and the K-step flag kernel (PCL_MULTI_AHEAD=0) -- bit for bit; two such calls in a row
    continue each other; the kernel's work tally (pcl_store_ahead_work) covers every alive slot once per launch."""
    from physicl_amd import _hip as hip
    N, dt, pdel, seed = 200_003, 1e-3, 0.2, 33
    A = pdel / (1e-3 * C_LIT * dt)

    def run(how):
        with hip.Device(0) as d:
            d.store_alloc(N, dtype)
            d.fill_photons(N, 5, C_LIT, 1.0, 2.0, seed)
            rows = []
            if how == "single":
                for k in range(2 * K):
                    o = d.step_fused_delete(dt, A, 1e-3, hip.RNG_PHILOX, seed, 3 + k, planes, lazy=True)   # (odd first step)
                    rows.append(device_row(o) if planes is not None else [o["N"], o["removed"]])
            else:
                for call in range(2):
                    for o in d.step_fused_delete_multi(dt, K, A, 1e-3, seed, 3 + call * K, planes):
                        rows.append(device_row(o) if planes is not None else [o["N"], o["removed"]])
            stats, work = d.ahead_stats(), d.ahead_work()
            return rows, d.download_state(), stats, work

    try:
        hip.set_knob("PCL_AHEAD", "0")
        rows0, s0, st0, _ = run("single")
        hip.set_knob("PCL_AHEAD", None)
        rows1, s1, st1, w1 = run("multi")
        hip.set_knob("PCL_MULTI_AHEAD", "0")
        rows2, s2, st2, _ = run("multi")
    finally:
        hip.set_knob("PCL_AHEAD", None)
        hip.set_knob("PCL_MULTI_AHEAD", None)
    assert rows0 == rows1 == rows2
    for sx in (s1, s2):
        assert np.array_equal(s0["id"], sx["id"]) and np.array_equal(s0["E"], sx["E"])
        for f in ("r", "v", "dr", "dv"):
            for k in range(3):
                assert np.array_equal(s0[f][k], sx[f][k]), (f, k)
    bodies = len([r for r in rows1 if r[0] + r[1] > 0])                     # bodies that found photons
    launches, served, missed = st1
    assert st0 == (0, 0, 0) and st2 == (0, 0, 0)
    # (a body that finds the store sparse enough to compact runs the plain way: at most a few of a run)
    assert bodies - 3 <= served <= bodies and 1 <= launches <= -(-bodies // min(K, 24)) + 2, (st1, bodies)
    if planes is None or len(planes) <= 1:                                  # (more planes: k_delete_ahead, which keeps no tally)
        g2, g1, r2, r1 = w1
        # the first launch loads every slot, and its first step is odd: a first pass of one body; later launches, later passes
        assert g1 >= -(-N // 128) and g2 + g1 > g1 - 1 and r2 >= 1
    else:
        assert w1 == (0, 0, 0, 0)


LIVE_PLANES = {"no_measure": None, "no_planes": [], "one_plane": [[2e5, np.nan, np.nan]], "three_planes": PL3}


@pytest.mark.parametrize("ids", ["implicit_ids", "explicit_ids"])
@pytest.mark.parametrize("first_step", [0, 1], ids=["even_first_step", "odd_first_step"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("planes_key", sorted(LIVE_PLANES))
def test_delete_ahead_live_kernel_vs_oracle(planes_key, dtype, first_step, ids):
    """k_delete_ahead_live -- the kernel that carries every delete leg of the bench -- directly against the oracle's
    step-by-step chain (physicl/light.py:239-260, physicl/newton.py:15-16), in every shape ahead_launch_t picks it for:
    no measure step at all (n_planes = -1), a measure step without planes, one plane; fp64 and fp32; a first body on an
    even and on an odd launch index (an odd one starts in the middle of a Philox decision block); ids implicit (a fresh
    store) and explicit (after an earlier compaction).  Nothing but the rows is looked at between the bodies, so the
    bodies are answered from the launch's rows; flags once near the end, survivor ids / r / v / dr at the end -- bit for
    bit.  ``pcl_store_ahead_work`` says which kernel ran: the live kernel tallies its groups and rounds, k_delete_ahead
    (three planes: the shape the live kernel does not take) tallies nothing."""
    from physicl_amd import _hip as hip
    planes = LIVE_PLANES[planes_key]
    np_t = np.float64 if dtype == "f64" else np.float32
    N, dt, n_k, seed = 300_007, 1e-3, 1e-3, 29
    rs = np.random.RandomState(11)
    direction = rs.normal(size=(N, 3))
    direction /= np.linalg.norm(direction, axis=1)[:, None]
    init = {"r": (rs.normal(size=(N, 3)) * 1e5).astype(np_t), "v": (direction * C_LIT).astype(np_t), "E": rs.uniform(1, 2, N).astype(np_t)}
    st = {"r": cols(init["r"]), "v": cols(init["v"]), "dr": [np.zeros(N, np_t)] * 3, "dv": [np.zeros(N, np_t)] * 3, "E": init["E"].copy(),
          "id": np.arange(N, dtype=np.int64) + 1000}
    oplanes = planes or []

    def body(d, step, pdel):
        A_k = pdel / (n_k * C_LIT * dt)
        before = d.count
        flags, row = oracle_body(st, dt, A_k, n_k, seed, step, oplanes, dtype=np_t)
        o = d.step_fused_delete(dt, A_k, n_k, hip.RNG_PHILOX, seed, step, planes, lazy=True)
        got = device_row(o) if planes is not None else [o["N"], o["removed"]]
        want = row if planes is not None else row[:2]
        assert got == want, (step, got, want)
        return before, flags

    try:
        with hip.Device(0) as d:
            d.store_alloc(N, dtype)
            d.upload_state(dict(init, id_base=1000))
            step = first_step
            if ids == "explicit_ids":
                hip.set_knob("PCL_AHEAD", "0")                     # two plain bodies, then a download: the store is compacted, ids explicit
                for _ in range(2):
                    body(d, step, 0.45)
                    step += 1
                assert_state(d, st, hip, "after the compaction")
                assert d.slots == d.count and not d.is_uniform()
                hip.set_knob("PCL_AHEAD", None)
            w0, (l0, s0, _) = d.ahead_work(), d.ahead_stats()
            bodies = 11
            for k in range(bodies):
                before, flags = body(d, step, 0.2)
                if k == bodies - 2:
                    assert np.array_equal(d.last_delete_flags(before), flags), step
                step += 1
            w1, (l1, s1, _) = d.ahead_work(), d.ahead_stats()
            assert_state(d, st, hip, "end")
            work = [b - a for a, b in zip(w0, w1)]
            assert l1 - l0 >= 1 and s1 - s0 > l1 - l0, (l0, s0, l1, s1)       # bodies were answered from rows worked out ahead
            if planes_key == "three_planes":
                assert work == [0, 0, 0, 0]                                    # k_delete_ahead ran (it keeps no tally)
            else:
                g2, g1, r2, r1 = work                                          # the live kernel ran: every slot loaded once per launch
                assert g2 + g1 >= (l1 - l0) and g2 + g1 >= -(-before // 128), work
                if ids == "implicit_ids":      # (after the download of the explicit case the library runs one plain body before it works ahead again)
                    assert (g1 > 0) if first_step % 2 == 1 else (g2 > 0), (work, first_step)     # odd first step: a first pass of one body
    finally:
        hip.set_knob("PCL_AHEAD", None)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_delete_ahead_live_with_changing_time_steps_vs_oracle(dtype):
    """The changing-dt chain of the test above with ONE plane, so that the live kernel is the one that works the bodies out:
    runs of three equal time steps (every change of dt ends a launch's prediction: the rows handed out so far are committed,
    r owes the run's moves, and a new launch starts from there).  Rows every body, flags now and then, the state twice."""
    from physicl_amd import _hip as hip
    np_t = np.float64 if dtype == "f64" else np.float32
    N, n_k, seed = 150_000, 1e-3, 31
    rs = np.random.RandomState(8)
    direction = rs.normal(size=(N, 3))
    direction /= np.linalg.norm(direction, axis=1)[:, None]
    init = {"r": (rs.normal(size=(N, 3)) * 1e5).astype(np_t), "v": (direction * C_LIT).astype(np_t), "E": rs.uniform(1, 2, N).astype(np_t)}
    planes = [[np.nan, 1e5, np.nan]]
    st = {"r": cols(init["r"]), "v": cols(init["v"]), "dr": [np.zeros(N, np_t)] * 3, "dv": [np.zeros(N, np_t)] * 3, "E": init["E"].copy(),
          "id": np.arange(N, dtype=np.int64)}
    dts = [1e-3 * (1.0 + 0.125 * ((k // 3) % 5)) for k in range(24)]
    with hip.Device(0) as d:
        d.store_alloc(N, dtype)
        d.upload_state(dict(init, id_base=0))
        for step, dt in enumerate(dts):
            before = d.count
            A_k = 0.05 / (n_k * C_LIT * dt)
            flags, row = oracle_body(st, dt, A_k, n_k, seed, step, planes, dtype=np_t)
            o = d.step_fused_delete(dt, A_k, n_k, hip.RNG_PHILOX, seed, step, planes, lazy=True)
            assert device_row(o) == row, (step, device_row(o), row)
            if step % 7 == 6:
                assert np.array_equal(d.last_delete_flags(before), flags), step
            if step in (10, 23):
                assert_state(d, st, hip, step)
        launches, served, missed = d.ahead_stats()
        assert launches >= 1 and sum(d.ahead_work()) > 0, (launches, served, missed, d.ahead_work())
