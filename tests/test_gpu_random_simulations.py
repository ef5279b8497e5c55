"""GPU: randomly generated simulations through the plugin API -- the scheduler's choices must not show.

``Simulation`` decides per pass how the step list runs: one launch per Step (``fuse=False``), consecutive native steps
as one kernel (``fuse=True``), up to K whole passes per launch (``steps_per_launch=K``), with the particles resident on
the device until a host plugin or a foreign thread looks at ``sim.objects`` (physicl/__init__.py:512-516 is the loop
being reproduced: every step, in insertion order, every pass).  This file draws step lists at random -- one to three
[Newton, light step, measure steps] groups, a time step that may change from pass to pass, host plugins that read,
edit or remove objects at random places, photons with or without plain Objects among them, or a bulk PhotonBatch --
runs each with four schedules (the fourth: the constructor's default, which decides by itself whether the exit test can be
evaluated ahead of a K-pass launch) and requires IDENTICAL results: ``ts``, every measure row, ``hits``, what the host
plugins saw, and the final r, v, dr, dv, E of every object, bit for bit.
"""
import os
import time

import numpy as np
import pytest

import physicl as phys
import physicl.light
import physicl.newton

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0


class Reader(phys.Step):
    """Looks without touching: object count and one photon's position, every pass."""
    def __init__(self):
        self.seen = []

    def run(self, s):
        n = len(s.objects)
        self.seen.append((n, tuple(np.asarray(s.objects[0].r, dtype=float)) if n else ()))


class Kicker(phys.Step):
    """Edits an object in Python every other pass."""
    def __init__(self):
        self.passes = 0

    def run(self, s):
        self.passes += 1
        if self.passes % 2 == 0 and len(s.objects) > 3:
            o = s.objects[3]
            o.r = o.r + np.array([0.0, 0.25, 0.0])


class Remover(phys.Step):
    """Removes the first object on its third pass (physicl/__init__.py:455-459)."""
    def __init__(self):
        self.passes = 0

    def run(self, s):
        self.passes += 1
        if self.passes == 3 and len(s.objects) > 1:
            s.remove_obj(s.objects[0])


def draw_config(rs):
    cfg = {"dt_kind": rs.choice(["const", "vary"]), "passes": int(rs.randint(4, 11)), "groups": [], "seed": int(rs.randint(1 << 30))}
    for _ in range(rs.randint(1, 4)):
        light = rs.choice(["iso", "iso_lambda", "delete", "none"], p=[0.35, 0.15, 0.35, 0.15])
        measures = [m for m in ("sign", "plane") if rs.random_sample() < 0.6]
        cfg["groups"].append((light, measures))
    source = rs.choice(["photons", "mixed", "batch"], p=[0.4, 0.3, 0.3])
    cfg["source"] = source
    cfg["n"] = int(rs.choice([40, 700, 2600])) if source != "batch" else int(rs.choice([5000, 150_000]))
    plugins = []
    if source != "batch":
        for kind in ("reader", "kicker", "remover"):
            if rs.random_sample() < 0.3:
                plugins.append((kind, int(rs.randint(0, 3 * len(cfg["groups"]) + 1))))
    elif rs.random_sample() < 0.3:
        plugins.append(("count", int(rs.randint(0, 3 * len(cfg["groups"]) + 1))))
    cfg["plugins"] = plugins
    cfg["K"] = int(rs.randint(2, 7))
    # host-drawn randoms in the reference's order (np.random, 3 per photon per scatter step, 1 per delete step) for some of
    # the explicit-object runs: every schedule must consume the global stream identically
    cfg["rng"] = "numpy" if source != "batch" and rs.random_sample() < 0.35 else "philox"
    # (drawn last, so that the configurations of earlier rounds stay what they were) what ``exit`` looks at: the clock and
    # emptiness -- plannable ahead of a K-pass launch -- or the run's own data, in three ways that must all fall back to
    # one launch per light step without showing: a measure step's rows through a closure, ``sim.hits`` through the
    # argument, a threshold on the object count
    cfg["exit_kind"] = str(rs.choice(["clock", "closure_rows", "sim_hits", "count"], p=[0.55, 0.15, 0.15, 0.15]))
    # a time step with units: the clock is then a Measurement, which ``t += dt`` advances IN PLACE (physicl/__init__.py:343
    # deep-copies it into ts for that reason) -- every schedule must leave the same ts behind
    cfg["dt_measurement"] = bool(rs.random_sample() < 0.3)
    # (round 4, drawn after everything else) an exit that ALSO asks the wall clock, the global random stream, or counts its
    # own calls -- never true here, but evaluated K times ahead of a launch it would be another program than the
    # reference's loop, which asks once per pass (physicl/__init__.py:512-516): one launch per light step, silently
    cfg["exit_extra"] = str(rs.choice(["none", "wall_clock", "np_random", "counting"], p=[0.64, 0.12, 0.12, 0.12]))
    return cfg


class Count(phys.Step):
    def __init__(self):
        self.seen = []

    def run(self, s):
        self.seen.append(len(s.objects))


def build_and_run(cfg, steps_per_launch, fuse, **sim_kw):
    T = cfg["passes"]
    np.random.seed(cfg["seed"] % (1 << 31))
    measures = []
    kind = cfg.get("exit_kind", "clock")
    if kind == "closure_rows":             # stops one pass after the first measure step has T - 2 rows
        exit_fn = lambda s: len(s.ts) >= T or len(s.objects) == 0 or (len(measures) > 0 and len(measures[0].data) >= T - 2)   # noqa: E731
    elif kind == "sim_hits":               # stops after a scatter step with fewer than 5 hits
        exit_fn = lambda s: len(s.ts) >= T or len(s.objects) == 0 or (len(s.ts) > 2 and s.hits < 5)   # noqa: E731
    elif kind == "count":                  # stops when fewer than 45 % of the objects are left
        exit_fn = lambda s: len(s.ts) >= T or len(s.objects) * 100 < cfg["n"] * 45   # noqa: E731
    else:
        exit_fn = lambda s: len(s.ts) >= T or len(s.objects) == 0   # noqa: E731
    extra = cfg.get("exit_extra", "none")
    if extra != "none":
        inner, t0, calls = exit_fn, time.time(), [0]
        if extra == "wall_clock":
            exit_fn = lambda s: inner(s) or time.time() - t0 > 36000.0   # noqa: E731
        elif extra == "np_random":             # consumes the global stream the host-drawn randoms come from: once per pass, or the rows differ
            exit_fn = lambda s: inner(s) or np.random.random() > 2.0   # noqa: E731
        else:
            def exit_fn(s):
                calls[0] += 1
                return inner(s) or calls[0] > 10 ** 9
    kw = {} if steps_per_launch == "default" else {"steps_per_launch": steps_per_launch}     # "default": the constructor's own
    sim = phys.Simulation(cl_on=True, rng=cfg["rng"], seed=cfg["seed"], fuse=fuse, exit=exit_fn, **kw, **sim_kw)
    rs = np.random.RandomState(cfg["seed"])
    if cfg["source"] == "batch":
        sim.add_objs(phys.light.generate_photons_bulk(cfg["n"], min=phys.light.E_from_wavelength(700e-9),
                                                      max=phys.light.E_from_wavelength(200e-9), seed=cfg["seed"]))
    else:
        objs = []
        for i in range(cfg["n"]):
            d = np.zeros(3)                                   # PhotonObject insists on |v| == c exactly (light.py:23-24)
            d[rs.randint(3)] = rs.choice([-1.0, 1.0])
            if cfg["source"] == "mixed" and i % 5 == 2:
                objs.append(phys.Object(v=phys.Measurement(rs.normal(size=3), "m**1 s**-1"), uid=i))
            else:
                objs.append(phys.light.PhotonObject(v=d * C_LIT, E=np.double(rs.uniform(2.8e-19, 9.9e-19)), uid=i))
        sim.add_objs(objs)
    wrap = (lambda x: phys.Measurement(np.double(x), "s**1")) if cfg.get("dt_measurement") else np.double
    if cfg["dt_kind"] == "const":
        sim.add_step(0, phys.UpdateTimeStep(lambda s: wrap(0.001)))
    else:
        sim.add_step(0, phys.UpdateTimeStep(lambda s: wrap(0.001) if len(s.ts) % 3 else wrap(0.0005)))
    steps = []
    for light, ms in cfg["groups"]:
        steps.append(phys.newton.NewtonianKinematicsStep())
        if light == "iso":
            steps.append(phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
        elif light == "iso_lambda":
            steps.append(phys.light.ScatterIsotropicStep(A=np.double(1e-19), n=np.double(1e-15), wavelength_dep_scattering=True))
        elif light == "delete":
            steps.append(phys.light.ScatterDeleteStep(np.double(0.001), np.double(0.0004)))
        for m in ms:
            st = phys.light.ScatterSignMeasureStep(None, True) if m == "sign" else \
                phys.light.ScatterMeasureStep(None, True, [[100.0, np.nan, np.nan], [np.nan, -50.0, np.nan]])
            steps.append(st)
            measures.append(st)
    plugs = []
    for kind, pos in sorted(cfg["plugins"], key=lambda p: -p[1]):
        p = {"reader": Reader, "kicker": Kicker, "remover": Remover, "count": Count}[kind]()
        steps.insert(min(pos, len(steps)), p)
        plugs.append(p)
    for k, st in enumerate(steps):
        sim.add_step(k + 1, st)
    sim.start()
    sim.join()
    assert sim.error is None, sim.error
    out = {"ts": [float(t) for t in sim.ts], "hits": int(sim.hits),
           "rows": [[[float(x) for x in np.ravel(np.asarray(c, dtype=float))] for c in m.data] for m in measures],
           "seen": [getattr(p, "seen", None) for p in plugs], "n": len(sim.objects), "schedule": dict(sim.schedule),
           "note": sim.launch_note,
           "next_random": float(np.random.random_sample())}        # where the run left the global stream
    if cfg["source"] == "batch":
        out["state"] = {f: sim.download(f) for f in ("r", "v", "dr", "dv", "E")} if out["n"] else {}
    else:
        objs = list(sim.objects)
        out["uids"] = [o.uid for o in objs]
        out["state"] = {f: np.array([np.asarray(getattr(o, f), dtype=float) for o in objs]) for f in ("r", "v", "dr", "dv")}
        out["state"]["E"] = np.array([float(o.E) if type(o) is phys.light.PhotonObject else np.nan for o in objs])
    sim.close()
    return out


def assert_same(a, b, what):
    for k in ("ts", "hits", "rows", "seen", "n", "next_random"):
        assert a[k] == b[k], (what, k)
    assert a.get("uids") == b.get("uids"), what
    assert sorted(a["state"]) == sorted(b["state"])
    for f in a["state"]:
        assert np.array_equal(a["state"][f], b["state"][f], equal_nan=True), (what, f)


SEEN = {}
AUTO = {}
EXTRA = {}


@pytest.mark.parametrize("seed", range(int(os.environ.get("PCL_RANDOM_SEEDS", "60"))))
def test_random_simulation_does_not_depend_on_the_schedule(seed):
    cfg = draw_config(np.random.RandomState(500 + seed))
    base = build_and_run(cfg, 1, False)
    assert len(base["ts"]) >= 1 and not base["schedule"]
    fused = build_and_run(cfg, 1, True)
    assert_same(base, fused, ("fuse", cfg))
    multi = build_and_run(cfg, cfg["K"], True)
    assert_same(base, multi, ("steps_per_launch", cfg))
    # the constructor's own default (what a script written against the reference gets): automatic, up to 32 passes per launch
    auto = build_and_run(cfg, "default", True)
    assert_same(base, auto, ("default constructor", cfg))
    has_measures = any(ms for _, ms in cfg["groups"])
    plannable = cfg["exit_kind"] == "clock" or (cfg["exit_kind"] == "count" and not any(g[0] == "delete" for g in cfg["groups"])) \
        or (cfg["exit_kind"] == "closure_rows" and not has_measures)
    went_multi = any(k.endswith("_multi") for k in auto["schedule"])
    if went_multi and not plannable and len(base["ts"]) > 3 and cfg["exit_extra"] == "none":
        # a launch may carry the passes planned before the guard tripped; after that, one launch per light step -- and a note
        assert auto["note"] and "one launch per light step" in auto["note"], (cfg, auto["schedule"], auto["note"])
    if cfg["exit_extra"] != "none":            # never planned ahead, and the note says what the function reaches
        plannable = False
        # (a loop that is not eligible for K passes per launch anyway -- host plugins, host-drawn randoms -- never asks the guard)
        assert not went_multi and (auto["note"] is None or "one launch per light step: exit" in auto["note"]), (cfg, auto["schedule"], auto["note"])
        EXTRA[cfg["exit_extra"]] = EXTRA.get(cfg["exit_extra"], 0) + int(auto["note"] is not None)
    if cfg["exit_kind"] == "closure_rows" and has_measures:
        assert not went_multi and (auto["note"] is None or "closes over" in auto["note"]), (cfg, auto["schedule"], auto["note"])
    for k, v in list(fused["schedule"].items()) + list(multi["schedule"].items()) + list(auto["schedule"].items()):
        SEEN[k] = SEEN.get(k, 0) + v
    AUTO[cfg["exit_kind"]] = AUTO.get(cfg["exit_kind"], 0) + int(went_multi)


def test_the_random_simulations_reached_every_schedule():
    """(runs after the cases above) the draw is only worth its time if all five launch formulations were chosen."""
    if len(SEEN) == 0:
        pytest.skip("the parametrised cases did not run in this process")
    assert {"fused", "fused_delete", "fused_multi", "fused_delete_multi", "mixed_multi"} <= set(SEEN), SEEN
    # the default constructor took the K-pass path where the exit test allows it, and every kind of exit was drawn
    assert AUTO.get("clock", 0) > 0 and {"clock", "closure_rows", "sim_hits", "count"} <= set(AUTO), AUTO
    # every kind was drawn and the guard refused some of them (each kind's refusal itself is pinned in tests/test_ahead_cpu.py;
    # here a drawn loop may not be eligible for K passes per launch in the first place)
    assert {"wall_clock", "np_random", "counting"} <= set(EXTRA) and sum(EXTRA.values()) >= 2, EXTRA
