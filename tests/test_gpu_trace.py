"""GPU: TracePathMeasureStep for a tracked subset, worked out on the device ahead of the K-pass launch
(pcl_store_trace_ahead; physicl/light.py:433-483 is the step it serves).

Bars:
* Level 2: the rows are bit-identical to downloading r (and dv) of the tracked ids after every pass of the same loop run
  one launch per light step -- isotropic, delete and both mixed loops, every pcoll variant, fp64 and fp32, on dense stores,
  stores with explicit ids (after a compaction), with plain Objects, behind an alive mask with moves pending, on a shard
  (id_base != 0); NaN rows from the pass that removed a photon on; the store is not changed by the call, and the last row
  equals the store after the K-pass launch.
* against the CPU oracle's chain on the tracked ids over 32 steps: same decisions (NaN pattern / hit pattern), positions
  within K * dt * 4 ulp(c) (the oracle's libm sin/cos differ from the device's by <= 4 ulp of c in a scattered velocity).
* Simulation: the table of a device-traced run == the table the host plugin builds on the same run (explicit objects),
  for scatter, delete and mixed loops; a PhotonBatch run traces its first 1000 photons on the K-passes-per-launch
  schedule; two contexts (devices=[0, 0]) give the one-context table.
"""
import numpy as np
import pytest

from oracle import physicl_oracle as orc

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
H_LIT = 6.62607015e-34
V_ABS_TOL = 4 * np.spacing(C_LIT)
EXPR_EX = "0.000000001 * exp(r0[gid] - 5)"
EXPR_RAD = "2.5 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - 6.0)/(3.5))"
CASES = {
    # tag: (use_E, expr, A, n, dt)
    "base": (False, None, 1e-3, 1e-3, 1e-3),
    "lambda": (True, None, 1e-15, 1e-19, 5e-3),
    "varn": (True, EXPR_EX, 1e-15, 1e-19, 1e-9),
    "varn_radial": (False, EXPR_RAD, 0.4, 1.0, 1e-9),           # (kernel A scales the expression: pcoll 0.03 .. 1.6 over the box)
}
DELETE = (1e-3, 0.4e-3)          # (A, n) of the delete phase: 12 % removed per pass at dt = 1e-3


@pytest.fixture(scope="module")
def hip():
    from physicl_amd import _hip
    return _hip


@pytest.fixture()
def make_store(hip):
    devs = []

    def make(capacity, dtype="f64"):
        d = hip.Device(0)
        d.store_alloc(capacity, dtype)
        devs.append(d)
        return d
    yield make
    for d in devs:
        d.close()


def scatter_dict(hip, tag, seed, step):
    use_e, expr, A, n, dt = CASES[tag]
    flags = (hip.SCATTER_WAVELENGTH if use_e else 0) | (hip.SCATTER_VARIABLE_N if expr else 0)
    return dict(A=A, n=n, flags=flags, c=C_LIT, h=H_LIT, n_expr=expr, rng_mode=hip.RNG_PHILOX, seed=seed, step=step), dt


def initial(N, dtype, seed, id_base=0, kinds=False):
    rs = np.random.RandomState(seed)
    npdt = np.float64 if dtype == "f64" else np.float32
    st = {"r": rs.uniform(-8, 8, (N, 3)).astype(npdt), "v": np.tile([C_LIT, 0.0, 0.0], (N, 1)).astype(npdt),
          "E": rs.uniform(2.8e-19, 9.9e-19, N).astype(npdt), "id_base": id_base}
    if kinds:
        st["kind"] = (rs.uniform(size=N) < 0.8).astype(np.uint8)
    return st


def tracked(N, id_base=0):
    """ids {0 .. 999}, a window across the first tile boundary, the last 64 -- and three that are not in the store."""
    ids = np.concatenate([np.arange(min(1000, N)), np.arange(2040, 2056), np.arange(max(N - 64, 0), N), [N, N + 5, 10 * N + 3]])
    return np.unique(ids[(ids >= 0)]) + id_base


def eq_nan(a, b):
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def chain_by_single_launches(d, hip, ids, tag, dt, K, phases, record_phase, seed, step0):
    """The same loop one launch per light step, r / dv of the tracked ids downloaded behind phase ``record_phase``."""
    rows = np.full((K, len(ids), 4), np.nan)
    step = step0
    for k in range(K):
        for j, ph in enumerate(phases):
            if ph == "iso":
                sc, _ = scatter_dict(hip, tag, seed, step)
                d.step_fused(dt, sc, [], lazy=True)
            else:
                d.step_fused_delete(dt, DELETE[0], DELETE[1], hip.RNG_PHILOX, seed, step, None, lazy=True)
            step += 1
            if j == record_phase:
                have = d.download_ids()
                pos = np.searchsorted(have, ids)
                pos[pos >= len(have)] = 0
                there = (have[pos] == ids) if len(have) else np.zeros(len(ids), bool)
                r = np.stack([d.download(f) for f in (hip.R0, hip.R1, hip.R2)], 1)
                dv = np.stack([d.download(f) for f in (hip.DV0, hip.DV1, hip.DV2)], 1)
                rows[k, there, :3] = r[pos[there]]
                rows[k, there, 3] = np.any(dv[pos[there]] != 0, axis=1)
    return rows


LOOPS = {"iso": (("iso",), 0), "delete": (("delete",), 0), "iso_delete": (("iso", "delete"), 1), "iso_delete_rec0": (("iso", "delete"), 0),
         "delete_iso": (("delete", "iso"), 1)}


def launch(d, hip, tag, dt, K, phases, seed, step0):
    sc, _ = scatter_dict(hip, tag, seed, step0)
    if phases == ("iso",) and d.is_uniform():
        return d.step_fused_multi(dt, K, sc)
    if phases == ("delete",):
        return d.step_fused_delete_multi(dt, K, DELETE[0], DELETE[1], seed, step0)
    return d.step_mixed_multi(dt, K, phases, sc, DELETE, (), seed, step0)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("loop", sorted(LOOPS))
@pytest.mark.parametrize("tag", sorted(CASES))
def test_trace_ahead_is_the_download_after_every_pass(make_store, hip, tag, loop, dtype):
    phases, rec = LOOPS[loop]
    N, K = 5003, 9
    _, _, _, _, dt = CASES[tag]
    if "delete" in phases:
        dt = 1e-3                                   # |v dt| = 3e5 m: the delete phase removes 12 % per pass
    init = initial(N, dtype, 11)
    ids = tracked(N)
    seed, step0 = 0xFEEDF00D, 7
    a = make_store(N, dtype)
    a.upload_state(init)
    before = a.download_state()
    sc, _ = scatter_dict(hip, tag, seed, step0)
    rows = a.trace_ahead(ids, dt, K, phases, rec, sc, DELETE, seed, step0)
    after = a.download_state()
    for f in ("r", "v", "dr", "dv"):                # the call reads the store, nothing else
        for k in range(3):
            assert np.array_equal(before[f][k], after[f][k])
    b = make_store(N, dtype)
    b.upload_state(init)
    ref = chain_by_single_launches(b, hip, ids, tag, dt, K, phases, rec, seed, step0)
    assert eq_nan(rows, ref)
    assert np.all(np.isnan(rows[:, -3:, :]))        # ids that are not in the store
    if "delete" in phases:
        assert np.isnan(rows[-1, :1000, 0]).sum() > 100 and np.isnan(rows[0, :1000, 0]).sum() < 400
    else:
        assert not np.isnan(rows[:, :-3, :]).any()
        assert 0 < rows[:, :-3, 3].sum()             # somebody scattered, so some dv is not zero
    # ... and the K-pass launch the rows were worked out for leaves the tracked photons where the last row says
    launch(a, hip, tag, dt, K, phases, seed, step0)
    if rec == len(phases) - 1:
        have = a.download_ids()
        r = np.stack([a.download(f) for f in (hip.R0, hip.R1, hip.R2)], 1)
        there = ~np.isnan(rows[-1, :, 0])
        pos = np.searchsorted(have, ids[there])
        assert np.array_equal(have[pos], ids[there])
        assert np.array_equal(r[pos], rows[-1, there, :3])
        assert set(ids[~there]).isdisjoint(have.tolist())


@pytest.mark.parametrize("state", ["compacted", "kinds", "alive_mask", "shard"])
@pytest.mark.parametrize("loop", ["iso", "delete", "iso_delete"])
def test_trace_ahead_on_every_kind_of_store(make_store, hip, state, loop, pcl_knobs):
    phases, rec = LOOPS[loop]
    N, K, tag, dt = 70_001, 6, "base", 1e-3
    id_base = 3_000_000_000 if state == "shard" else 0
    init = initial(N, "f64", 5, id_base, kinds=(state == "kinds"))
    seed = 31
    stores = []
    for _ in range(2):
        d = make_store(N)
        d.upload_state(init)
        step = 1
        if state == "compacted":                    # explicit ids: one eager delete step
            d.step_newton(dt)
            d.step_scatter_delete(1e-3, 1e-3, hip.RNG_PHILOX, seed, step)
            step += 1
        if state == "alive_mask":                   # three lazy delete bodies on the alive mask: holes, r owed three moves
            for _k in range(3):
                d.step_fused_delete(dt, 1e-3, 0.2e-3, hip.RNG_PHILOX, seed, step, None, lazy=True)
                step += 1
        stores.append(d)
    a, b = stores
    ids = tracked(N, id_base)
    sc, _ = scatter_dict(hip, tag, seed, step)
    if state == "alive_mask":
        assert a.slots > a.count                     # the store really is behind a mask
    rows = a.trace_ahead(ids, dt, K, phases, rec, sc, DELETE, seed, step)
    ref = chain_by_single_launches(b, hip, ids, tag, dt, K, phases, rec, seed, step)
    assert eq_nan(rows, ref)
    assert 0 < (~np.isnan(rows[:, :, 0])).sum()
    if state == "kinds" and "delete" in phases:      # plain Objects are never removed
        plain = init["kind"][ids[:-3] - id_base] == 0
        assert not np.isnan(rows[:, :-3, 0][:, plain]).any()
    launch(a, hip, tag, dt, K, phases, seed, step)
    have = a.download_ids()
    there = ~np.isnan(rows[-1, :, 0])
    assert set(ids[there]).issubset(have.tolist()) and set(ids[~there]).isdisjoint(have.tolist())


@pytest.mark.parametrize("loop", ["iso", "delete", "iso_delete"])
@pytest.mark.parametrize("tag", ["base", "varn"])
def test_trace_ahead_vs_the_oracle_chain_over_32_steps(make_store, hip, tag, loop):
    phases, rec = LOOPS[loop]
    N, K = 300_000, 32 // len(phases)
    use_e, expr, A, n, dt = CASES[tag]
    if "delete" in phases:
        dt = 1e-3
    init = initial(N, "f64", 77)
    ids = tracked(N)[:-3]
    seed, step0 = 90210, 4
    d = make_store(N)
    d.upload_state(init)
    sc, _ = scatter_dict(hip, tag, seed, step0)
    dl = (1e-3, 0.1e-3)
    rows = d.trace_ahead(ids, dt, K, phases, rec, sc, dl, seed, step0)
    st = {"r": [np.ascontiguousarray(init["r"][ids, k]) for k in range(3)],
          "v": [np.ascontiguousarray(init["v"][ids, k]) for k in range(3)],
          "dr": [np.zeros(len(ids))] * 3, "dv": [np.zeros(len(ids))] * 3, "E": init["E"][ids].copy(), "id": ids.copy()}
    step = step0
    tol_v = V_ABS_TOL
    for k in range(K):
        for j, ph in enumerate(phases):
            orc.step_newton(st, dt)
            if ph == "iso":
                orc.step_scatter_isotropic(st, orc.philox_draws(seed, step, st["id"]), A, n, C_LIT, h=H_LIT, use_E=use_e, n_expr=expr)
            else:
                orc.step_scatter_delete(st, orc.philox_draws(seed, step, st["id"])[2], dl[0], dl[1])
            step += 1
            if j == rec:
                there = ~np.isnan(rows[k, :, 0])
                assert np.array_equal(ids[there], st["id"])                      # the same photons are gone
                got, want = rows[k, there, :3], np.stack(st["r"], 1)
                # positions: every earlier scattered velocity may be 4 ulp(c) off, times dt, per step since
                assert np.max(np.abs(got - want)) <= (k + 1) * len(phases) * dt * tol_v + 4 * np.spacing(np.max(np.abs(want)))
                if ph == "iso":
                    moved = np.any(np.stack(st["dv"], 1) != 0, axis=1)
                    assert np.array_equal(rows[k, there, 3] != 0, moved)         # the same photons were scattered
    if "delete" in phases:
        assert 0 < np.isnan(rows[-1, :, 0]).sum() < len(ids)


def test_trace_ahead_refuses_what_it_cannot_do(make_store, hip):
    d = make_store(100)
    d.upload_state(initial(100, "f64", 1))
    sc, dt = scatter_dict(hip, "base", 1, 0)
    with pytest.raises(hip.HipError):
        d.trace_ahead([5, 3], dt, 2, ("iso",), 0, sc, None, 1, 0)                # not ascending
    with pytest.raises(hip.HipError):
        d.trace_ahead([1, 2], dt, 2, ("iso",), 1, sc, None, 1, 0)                # record_phase outside the pass
    with pytest.raises(hip.HipError):
        d.trace_ahead([1, 2], dt, 40, ("iso", "delete"), 0, sc, DELETE, 1, 0)    # more phases than a launch holds
    assert d.trace_ahead([], dt, 2, ("iso",), 0, sc, None, 1, 0).shape == (2, 0, 4)


# ---------------------------------------------------------------------------------------------------------------------
# Simulation level
# ---------------------------------------------------------------------------------------------------------------------
def build_sim(kind, n, spl, fuse=True, devices=None, batch=False, trace_kw=None, seed=3, passes=12, tp_first=False):
    import physicl as phys
    import physicl.light
    import physicl.newton
    kw = dict(cl_on=True, seed=seed, rng="philox", steps_per_launch=spl, fuse=fuse, exit=lambda s: s.t >= (passes - 0.5) * 1e-3)
    if devices is not None:
        kw["devices"] = devices
    sim = phys.Simulation(**kw)
    if batch:
        sim.add_objs(phys.light.generate_photons_bulk(n, min=phys.light.E_from_wavelength(700e-9), max=phys.light.E_from_wavelength(200e-9), seed=seed))
    else:
        sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(3e-19 * (1 + 1e-4 * i)), uid=i)
                      for i in range(n)])
    tp = phys.light.TracePathMeasureStep(None, **(trace_kw or {}))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    if kind == "delete":
        sim.add_step(2, phys.light.ScatterDeleteStep(np.double(0.0004), np.double(0.001)))
    else:
        sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    signs = phys.light.ScatterSignMeasureStep(None, True)
    if kind == "mixed":
        sim.add_step(3, signs)
        if tp_first:
            sim.add_step(7, tp)                   # behind the scatter phase, before the second move
        sim.add_step(4, phys.newton.NewtonianKinematicsStep())
        sim.add_step(5, phys.light.ScatterDeleteStep(np.double(0.0002), np.double(0.001)))
        if not tp_first:
            sim.add_step(6, tp)
    else:
        sim.add_step(3, tp)
        sim.add_step(4, signs)
    return sim, tp, signs


def table(tp):
    """The trace table with every entry as a float list (positions are 3-vectors, the padding scalar NaNs)."""
    out = []
    for row in tp.data[1:]:
        out.append([row[0]] + [np.asarray(x, dtype=np.float64).reshape(-1).tolist() for x in row[1:]])
    return [np.asarray(x, dtype=np.float64).reshape(-1).tolist() if not isinstance(x, str) else x for x in tp.data[0]], out


def tables_equal(a, b):
    ta, ra = table(a)
    tb, rb = table(b)
    assert ta == tb and len(ra) == len(rb)
    for x, y in zip(ra, rb):
        assert x[0] == y[0] and len(x) == len(y)
        for p, q in zip(x[1:], y[1:]):
            assert len(p) == len(q) and all((u == v) or (u != u and v != v) for u, v in zip(p, q)), (p, q)


@pytest.mark.parametrize("trace_dv", [False, True])
@pytest.mark.parametrize("kind", ["iso", "delete", "mixed", "mixed_tp_first"])
def test_simulation_traces_on_the_device_what_the_host_plugin_traces(kind, trace_dv, pcl_knobs):
    """Explicit objects, every object traced (the reference's meaning): K passes per launch with the tracked subset worked
    out ahead == one launch per light step == the host plugin walking the downloaded state (fuse=False: nothing fused,
    TracePathMeasureStep runs as a plugin behind the separate steps)."""
    base, first = kind.split("_tp_")[0] if "_tp_" in kind else kind, kind.endswith("first")
    runs = []
    for spl, fuse in ((8, True), (1, True), (1, False)):
        sim, tp, signs = build_sim(base, 700, spl, fuse, trace_kw=dict(trace_dv=trace_dv), tp_first=first)
        sim.run()
        runs.append((sim, tp, signs))
    assert runs[0][0].schedule["fused_multi"] + runs[0][0].schedule["fused_delete_multi"] + runs[0][0].schedule["mixed_multi"] >= 1
    assert runs[0][0].launch_note is None
    assert not runs[2][0].schedule                                   # nothing fused there: the host plugin
    for sim, tp, signs in runs[1:]:
        tables_equal(runs[0][1], tp)
        assert [list(r) for r in signs.data] == [list(r) for r in runs[0][2].data]
    t, rows = table(runs[0][1])
    # (a photon removed before the step first runs is never seen: no row, as in the reference)
    assert len(t) == 13 and (len(rows) == 700 if (base == "iso" or first) else 500 < len(rows) < 700)
    if base != "iso":                                                # removed photons: shorter lists, NaN padding
        assert any(np.isnan(r[-1]).all() for r in rows)
    if trace_dv and base != "delete":
        assert sum(r[1][0] for r in rows) > 0                        # freq column counts the scatterings
    for s, _, _ in runs:
        s.close(download=False)


def test_photon_batch_is_traced_on_the_k_pass_schedule():
    """A PhotonBatch (no Python objects): the first 1000 photons by default, trace_ids across a tile boundary and at the end of
    the store, K passes per launch; equal to the one-launch-per-step run and to two contexts on the device."""
    n = 300_000
    ids = list(range(0, 1000)) + list(range(2040, 2056)) + list(range(n - 64, n))
    ref = None
    for spl, devices in ((8, None), (1, None), (8, [0, 0])):
        sim, tp, signs = build_sim("iso", n, spl, batch=True, devices=devices, trace_kw=dict(trace_ids=ids), passes=20)
        sim.run()
        assert sim._batch is not None                                # nobody materialised a photon
        if spl > 1:
            assert sim.schedule["fused_multi"] >= 2 and not sim.schedule["fused"]
        t, rows = table(tp)
        assert len(rows) == len(ids) and rows[0][0] == "<class 'physicl.light.PhotonObject'>"     # the reference's label
        if ref is None:
            ref = (tp, [list(r) for r in signs.data])
            x = sim.download("r")
            for j, i in enumerate(ids):                              # the last traced position is where the photon is
                assert rows[j][-1] == x[i].tolist()
        else:
            tables_equal(ref[0], tp)
            assert [list(r) for r in signs.data] == ref[1]
        sim.close(download=False)
    # the default: the first 1000
    sim, tp, _ = build_sim("mixed", 50_000, None, batch=True, passes=6)
    sim.run()
    assert 800 < len(tp.data) - 1 <= 1000 and sim.schedule["mixed_multi"] >= 1      # (the first pass's delete phase removed some unseen)
    sim.close(download=False)


def test_leaving_the_device_traced_schedule_keeps_the_trace_ids():
    """Eight device-traced passes, then the loop stops being fused (fuse=False: TracePathMeasureStep runs as a host plugin
    behind separate steps) and goes on: still one row per object, every position in it."""
    sim, tp, _ = build_sim("iso", 300, 4, passes=8)
    sim.run()
    assert all(len(tp.pos_dict[i]["pos"]) == 8 for i in range(300))
    ref, tp_ref, _ = build_sim("iso", 300, 1, passes=12)
    ref.run()
    sim.fuse = False
    sim.exit = lambda s: s.t >= 11.5e-3
    sim.running = True                     # (Simulation.run would reset the clock, as the reference's does: the loop by hand)
    while not sim.exit(sim):
        sim._run_pass()
    tp.terminate(sim)
    assert len(tp.data) == 301
    assert all(len(tp.pos_dict[i]["pos"]) == 12 for i in range(300))
    tables_equal(tp_ref, tp)
    sim.close(download=False)
    ref.close(download=False)


def test_objects_taken_to_the_host_in_the_middle_of_a_traced_delete_run():
    """Six device-traced passes of a delete loop, then a script looks at ``sim.objects`` (the list loses its removed photons and
    is uploaded again, renumbered), six more passes: every row is filed under the OBJECT it belongs to -- the same table as the
    host plugin builds over the very same sequence (fuse=False)."""
    tables = []
    for fuse, spl in ((True, 3), (False, 1)):
        sim, tp, _ = build_sim("delete", 600, spl, fuse, passes=6)
        sim.run()
        n_mid = len(sim.objects)
        first = sim.objects[0]                                   # residency -> host: the list now holds the survivors only
        assert n_mid < 600 and first is sim.objects[0]
        sim.exit = lambda s: s.t >= 11.5e-3
        sim.running = True                                       # (Simulation.run would reset the clock: the loop by hand)
        while not sim.exit(sim):
            sim._run_pass()
        tp.terminate(sim)
        tables.append(tp)
        assert len(sim.ts) == 12 and len(sim.objects) < n_mid
        sim.close(download=False)
    tables_equal(tables[0], tables[1])
    t, rows = table(tables[0])
    lens = sorted({len([x for x in r[1:] if len(x) == 3]) for r in rows})
    assert lens[0] < 6 < lens[-1] == 12                          # photons removed early, photons that lived through both halves


# ---------------------------------------------------------------------------------------------------------------------
# The reference's own TracePathMeasureStep (tests/golden/make_golden.py g5_trace: light.py:433-483 run unmodified behind the
# OpenCL light steps): the same seeded flow here, on the K-passes-per-launch schedule, one launch per step, and as a host plugin
# ---------------------------------------------------------------------------------------------------------------------
def _fixture_rows(z, pre, trace_dv):
    rows, at = [], 0
    for i in range(len(z[pre + "info"])):
        n = int(z[pre + "pos_len"][i])
        rows.append((str(z[pre + "info"][i]), int(z[pre + "freq"][i]) if trace_dv else None, int(z[pre + "lead_scalars"][i]),
                     z[pre + "pos"][at:at + n], int(z[pre + "trail_scalars"][i])))
        at += n
    return rows


def _table_rows(tp, trace_dv):
    """A table row of ours taken apart the way the generator took the reference's apart."""
    out = []
    for row in tp.data[1:]:
        body = list(row[2:] if trace_dv else row[1:])
        a = 0
        while a < len(body) and np.ndim(body[a]) == 0:
            assert np.isnan(body[a])
            a += 1
        b = len(body)
        while b > a and np.ndim(body[b - 1]) == 0:
            assert np.isnan(body[b - 1])
            b -= 1
        out.append((str(row[0]), int(row[1]) if trace_dv else None, a, np.array([np.asarray(x, dtype=np.float64).reshape(3) for x in body[a:b]]).reshape(-1, 3),
                    len(body) - b))
    return out


@pytest.mark.parametrize("mode", ["k_passes", "one_launch_per_step", "host_plugin"])
@pytest.mark.parametrize("case", ["iso", "del"])
def test_trace_table_is_the_reference_s_own(golden, case, mode):
    import physicl as phys
    import physicl.light
    import physicl.newton
    z = golden("g5_trace")
    N, dt, seed = int(z[case + "_N"]), float(z[case + "_dt"]), int(z[case + "_seed"])
    trace_dv = case == "iso"
    kw = dict(cl_on=True, fuse=mode != "host_plugin", steps_per_launch=4 if mode == "k_passes" else 1)
    if case == "iso":
        K = int(z["iso_K"])
        kw["exit"] = lambda s: s.t >= (K - 0.5) * dt
    sim = phys.Simulation(**kw)                                          # rng: numpy's stream, as the reference draws it
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i) for i in range(N)])
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    if case == "iso":
        sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(z["iso_A_user"]), n=np.double(z["iso_n_user"])))
        tp = phys.light.TracePathMeasureStep(None, trace_dv=True)
    else:
        sim.add_step(2, phys.light.ScatterDeleteStep(np.double(z["del_n_user"]), np.double(z["del_A_user"])))
        tp = phys.light.TracePathMeasureStep(None, id_info_fn=lambda o: "photon %d" % o.uid)
    sim.add_step(3, tp)
    np.random.seed(seed)
    sim.start()
    sim.join()
    assert sim.error is None
    assert tp.data[0][0] == str(z[case + "_label"])
    assert np.array_equal(np.asarray(tp.data[0][1:], dtype=np.float64), z[case + "_t_row"])
    want, got = _fixture_rows(z, case + "_", trace_dv), _table_rows(tp, trace_dv)
    assert len(got) == len(want)
    c_ulp = 4 * np.spacing(299792458.0)
    for (wi, wf, wa, wp, wb), (gi, gf, ga, gp, gb) in zip(want, got):
        assert (gi, gf, ga, gb) == (wi, wf, wa, wb) and gp.shape == wp.shape
        # positions: Euler sums of v dt with v within 4 ulp(c) of the reference's after a scattering; exact while nothing scattered
        assert np.max(np.abs(gp - wp), initial=0.0) <= (len(wp) * c_ulp * dt if case == "iso" else 0.0)
    sim.close(download=False)


@pytest.mark.parametrize("mode", ["fused", "host_plugin"])
def test_trace_table_of_photons_that_join_mid_run_is_the_reference_s_own(golden, mode):
    """g5_trace (c): a user Step adds photons in passes 2 and 4.  The reference pads a late joiner's row with 3 NaN scalars per
    missed pass in front AND as many again behind (``a = cols - len(pos)`` does not know about ``b``, light.py:474-479): the same
    table here."""
    import physicl as phys
    import physicl.light
    import physicl.newton
    z = golden("g5_trace")
    N, K, dt = int(z["join_N"]), int(z["join_K"]), float(z["join_dt"])

    class Joiner(phys.Step):
        def __init__(self):
            self.k = 0

        def run(self, sim):
            for j in range({2: 1, 4: 2}.get(self.k, 0)):
                sim.add_obj(phys.light.PhotonObject(v=np.array([0, phys.light.c, 0], dtype=np.double), E=np.double(1.0), uid=900 + 10 * self.k + j))
            self.k += 1

    sim = phys.Simulation(cl_on=True, fuse=mode == "fused", exit=lambda s: s.t >= (K - 0.5) * dt)
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i) for i in range(N)])
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    sim.add_step(3, Joiner())
    tp = phys.light.TracePathMeasureStep(None, id_info_fn=lambda o: "uid %d" % o.uid)
    sim.add_step(4, tp)
    np.random.seed(int(z["join_seed"]))
    sim.start()
    sim.join()
    assert sim.error is None
    assert np.array_equal(np.asarray(tp.data[0][1:], dtype=np.float64), z["join_t_row"])
    want, got = _fixture_rows(z, "join_", False), _table_rows(tp, False)
    assert len(got) == len(want) == N + 3
    c_ulp = 4 * np.spacing(299792458.0)
    for (wi, wf, wa, wp, wb), (gi, gf, ga, gp, gb) in zip(want, got):
        assert (gi, ga, gb) == (wi, wa, wb) and gp.shape == wp.shape
        assert np.max(np.abs(gp - wp), initial=0.0) <= K * c_ulp * dt
    assert [w[2] for w in want[-3:]] == [6, 12, 12] and [w[4] for w in want[-3:]] == [6, 12, 12]
    sim.close(download=False)
