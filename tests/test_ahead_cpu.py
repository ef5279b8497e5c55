"""CPU: the two guards that decide whether ``exit(sim)`` / ``UpdateTimeStep.fn(sim)`` may be evaluated ahead of a
K-pass launch (physicl_amd/ahead.py).  The reference calls both on the live simulation between passes
(physicl/__init__.py:512-516, 337-343); running K passes per launch is only invisible when they look at the clock,
the object count and constants."""
import functools
import math
import time

import numpy as np
import pytest

from physicl_amd.ahead import AheadView, NotAhead, clock_only
from physicl_amd.core import MeasureStep, Step
from physicl_amd.units import Measurement


class FakeSim:
    def __init__(self):
        self.t, self.dt, self.ts = 0.5, 0.1, [0.1, 0.2]
        self.hits, self.steps, self.seed, self.bounds = 7, {}, 3, np.zeros(3)
        self.start_time = 12.0


T_END = 0.25
PARAMS = {"dt": 1e-3, "name": "run", "grid": (1, 2, 3)}
M_GLOBAL = MeasureStep(None)


def exit_global_measure(s):
    return len(M_GLOBAL.data) > 3


COUNTER = 0


def exit_global_counter(s):
    global COUNTER
    COUNTER += 1
    return COUNTER > 10


def SPY(n, s):          # counts its calls (the static guard would refuse it; _plan_passes is called directly below)
    n.append(1)
    return False


def exit_global_const(s):
    return s.t >= T_END or math.isnan(PARAMS["dt"])


def test_view_exposes_the_clock_the_count_and_constants_only():
    sim = FakeSim()
    v = AheadView(sim, 42)
    assert v.t == 0.5 and v.dt == 0.1 and v.ts is sim.ts and v.seed == 3 and v.start_time == 12.0
    assert len(v.objects) == 42 and bool(v.objects) and not AheadView(sim, 0).objects
    for touch in (lambda: v.hits, lambda: v.steps, lambda: v.objects[0], lambda: list(v.objects), lambda: v.objects.index(1),
                  lambda: v.get_state(), lambda: setattr(v, "t", 1.0), lambda: v._alive):
        with pytest.raises(NotAhead):
            touch()
    # neither ``hasattr`` nor a broad ``except`` inside the user's function hides the access
    with pytest.raises(NotAhead):
        hasattr(v, "hits")
    def swallowing(s):
        try:
            return s.hits > 0
        except Exception:                # noqa: BLE001
            return False
    with pytest.raises(NotAhead):
        swallowing(v)


def test_functions_over_plain_data_are_clock_only():
    T, dt, arr, m = 10, np.double(1e-3), np.arange(4.0), Measurement(np.double(2.0), "s**1")
    helper = lambda s: len(s.ts)                                    # noqa: E731
    ok = [lambda s: len(s.ts) >= T or len(s.objects) == 0,
          lambda s: s.t >= 0.2495,
          lambda s: dt,
          lambda s: np.double(PARAMS["dt"]) if len(s.ts) % 3 else arr[0] * 1e-3,
          lambda s: m * 2,
          lambda s: helper(s) > 3 and math.sqrt(2.0) > 1,
          lambda s, limit=5: len(s.ts) >= limit,
          lambda cond: len(cond.objects) == 0,
          exit_global_const]
    for fn in ok:
        assert clock_only(fn, []) == (True, None), fn


def test_functions_that_can_reach_the_runs_own_data_are_not():
    m, other = MeasureStep(None), MeasureStep(None)
    rows, holder, sim = m.data, [], FakeSim()
    holder.append(m)

    class Ctx:
        pass
    ctx = Ctx()
    ctx.m = m
    bad = [lambda s: len(m.data) >= 10,                           # the measure step itself
           lambda s: len(rows) >= 10,                             # its data list, aliased
           lambda s: len(holder[0].data) >= 10,                   # a container holding the step
           lambda s: len(ctx.m.data) >= 10,                       # an object of an unknown class
           lambda s: sim.hits > 3,                                # another handle on a simulation
           lambda s: (lambda: len(m.data))() > 3,                 # through a nested function
           exit_global_measure,                                   # through a global
           functools.partial(lambda k, s: len(s.ts) > k, 3),      # not a plain function
           m.terminate]                                           # a bound method
    for fn in bad:
        ok, why = clock_only(fn, [m, other])
        assert not ok and why, fn
    # a list of the step's rows is only recognised by identity: without the steps it would look like plain data
    assert clock_only(lambda s: len(rows) >= 10, [])[0]
    assert not clock_only(lambda s: len(rows) >= 10, [m])[0]


def test_unbound_closure_cells_and_deep_or_huge_containers_fail_safe():
    def make():
        fn = lambda s: late > 3                                   # noqa: E731,F821
        ok = clock_only(fn, [])
        late = 5                                                  # noqa: F841
        return ok
    assert make()[0] is False
    deep = [[[[[1]]]]]
    assert not clock_only(lambda s: deep, [])[0]
    huge = list(range(5000))
    assert not clock_only(lambda s: huge, [])[0]
    assert isinstance(Step(), Step) and not clock_only(lambda s: s, [])[1]


# ---------------------------------------------------------------------------------------------------------------------
# Simulation._plan_passes: the host part of K passes, evaluated ahead (no device needed: cl_on=False opens none)
# ---------------------------------------------------------------------------------------------------------------------
def _sim(exit_fn, alive=100):
    import physicl_amd as phys
    sim = phys.Simulation(cl_on=False, exit=exit_fn)
    sim.t, sim.dt, sim.ts = 0, 0, []                  # what run() does first (physicl/__init__.py:508-510)
    sim._alive = alive
    return sim, phys


def test_plan_passes_follows_the_reference_loop_on_the_clock():
    sim, phys = _sim(lambda s: len(s.ts) >= 5)
    upd = phys.UpdateTimeStep(lambda s: np.double(0.001) if len(s.ts) % 3 else np.double(0.0005))
    times, dt0 = sim._plan_passes(upd, 8, False)
    # pass 1 has dt = 0.0005, pass 2 changes it: the launch is cut there and the clock is put back
    assert len(times) == 1 and dt0 == 0.0005 and [float(t) for t in sim.ts] == [0.0005] and sim._ahead_ok
    times, dt0 = sim._plan_passes(upd, 8, False)
    assert len(times) == 2 and dt0 == 0.001 and np.allclose([float(t) for t in sim.ts], [0.0005, 0.0015, 0.0025])
    times, dt0 = sim._plan_passes(upd, 8, False)
    assert len(times) == 1 and dt0 == 0.0005
    times, dt0 = sim._plan_passes(upd, 8, False)       # the fifth pass: exit stops the planning after it
    assert len(times) == 1 and len(sim.ts) == 5 and sim.exit(sim)


def test_plan_passes_stops_for_good_when_a_function_looks_at_the_run():
    sim, phys = _sim(lambda s: s.hits > 50 or len(s.ts) >= 9)
    upd = phys.UpdateTimeStep(lambda s: np.double(0.001))
    times, _ = sim._plan_passes(upd, 8, False)
    # the first pass was decided by the real loop, so it runs; the exit test after it cannot be made ahead
    assert len(times) == 1 and not sim._ahead_ok and "sim.hits" in sim.launch_note and len(sim.ts) == 1
    sim2, _ = _sim(lambda s: len(s.ts) >= 9)
    times, _ = sim2._plan_passes(phys.UpdateTimeStep(lambda s: np.double(0.001) * (1 + s.hits)), 8, False)
    assert times == [] and not sim2._ahead_ok and "time-step function" in sim2.launch_note and sim2.ts == []


def test_an_exit_that_depends_on_how_many_objects_are_left_is_not_planned_ahead_of_a_delete_step():
    upd_fn = lambda s: np.double(0.001)                               # noqa: E731
    # a threshold, a non-monotone test (``== 50`` used to pass the probe with one object and raise mid-run),
    # arithmetic on the count, the list handed on: nothing is planned, before anything has advanced
    for fn in (lambda s: len(s.objects) < 30, lambda s: len(s.objects) == 50, lambda s: len(s.objects) * 100 < 4500,
               lambda s: (lambda objs: len(objs) == 0)(s.objects), lambda s: 0 == len(s.objects)):
        sim, phys = _sim(fn)
        times, _ = sim._plan_passes(phys.UpdateTimeStep(upd_fn), 8, True)
        assert times == [] and sim.ts == [] and not sim._ahead_ok and "how many objects" in sim.launch_note, fn
    # the time-step function is held to the same rule
    sim, phys = _sim(lambda s: len(s.ts) >= 6)
    times, _ = sim._plan_passes(phys.UpdateTimeStep(lambda s: np.double(1e-6) * len(s.objects)), 8, True)
    assert times == [] and "time-step function depends on how many" in sim.launch_note
    # ... but only emptiness is fine, in any spelling, and without a delete step the count cannot change at all
    def helper(s):
        return not s.objects
    for fn in (lambda s: len(s.objects) == 0 or len(s.ts) >= 6, lambda s: not s.objects or len(s.ts) >= 6,
               lambda s: len(s.objects) < 1 or len(s.ts) >= 6, lambda s: helper(s) or len(s.ts) >= 6,
               lambda s: not (bool(s.objects) and len(s.ts) < 6)):
        sim, phys = _sim(fn)
        times, _ = sim._plan_passes(phys.UpdateTimeStep(upd_fn), 8, True)
        assert len(times) == 6 and sim._ahead_ok, fn
    sim, phys = _sim(lambda s: len(s.objects) < 30 or len(s.ts) >= 4)
    times, _ = sim._plan_passes(phys.UpdateTimeStep(upd_fn), 8, False)
    assert len(times) == 4 and sim._ahead_ok


def test_functions_of_the_wall_clock_of_a_random_stream_or_with_state_of_their_own_are_not_clock_only():
    """VERDICT r3 item 4 / ADVICE r3: evaluated K times ahead of a launch, ``time.time() - t0 > 5`` overshoots by up to K - 1
    passes, ``np.random.random() > .99`` consumes the global stream in another order than the reference's loop
    (physicl/__init__.py:512-516), and an exit that counts its own calls is simply called at other moments."""
    import datetime
    import os
    import random
    from random import random as rnd
    from time import time as now
    t0 = time.time()
    calls, seen = [0], {}

    def counting(s):
        calls[0] += 1
        return calls[0] > 10

    def appending(s):
        seen.setdefault("n", []).append(s.t)
        return len(seen["n"]) > 10

    def rebinding(s):
        nonlocal t0
        t0 += 1
        return t0 > 1e12

    def importing(s):
        import time as tm
        return tm.time() > 0

    bad = [lambda s: time.time() - t0 > 5, lambda s: np.random.random() > .99, lambda s: random.random() > .99,
           lambda s: rnd() > .99, lambda s: now() - t0 > 5, lambda s: datetime.datetime.now().year > 3000,
           lambda s: os.path.exists("/tmp/stop"), lambda s: np.random.default_rng().random() > 2,
           lambda s: __import__("time").time() > 0, lambda s: eval("1") > 2, lambda s: open("/dev/null") is None,
           counting, appending, rebinding, importing, exit_global_counter]
    for fn in bad:
        ok, why = clock_only(fn, [])
        assert not ok and why, fn
    # what stays plain: the maths of the allow-listed modules, the classes of the package, reading captured containers
    ok = [lambda s: np.sqrt(s.t) > 1 and abs(s.dt) < 2 and max(len(s.ts), 3) > 2, lambda s: PARAMS.get("dt", 0.0) > 1,
          lambda s: Measurement(np.double(1e-3), "s**1"), lambda s: float(np.asarray(s.t)) >= calls[0] + 3.0,
          lambda s: sum(x for x in PARAMS["grid"]) > s.t]
    for fn in ok:
        assert clock_only(fn, []) == (True, None), fn


def test_exit_is_called_once_per_pass_as_in_the_reference_loop():
    """in a loop with a ScatterDeleteStep exit(sim) used to be called up to three times per pass (the view, a
    second view with one object, the row replay).  Planning calls it once per planned boundary."""
    n = []
    sim, phys = _sim(lambda s: SPY(n, s) or len(s.ts) >= 10 or len(s.objects) == 0)
    times, _ = sim._plan_passes(phys.UpdateTimeStep(lambda s: np.double(0.001)), 8, True)
    assert len(times) == 8 and len(n) == 7 and sim._ahead_ok      # 7 boundaries inside the launch; the 8th is the outer loop's


def test_the_static_guard_runs_once_per_function_and_plan():
    m = MeasureStep(None)
    sim, phys = _sim(lambda s: len(m.data) >= 3)
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, m)
    sim._plan_key = "p"
    assert not sim._ahead_agreed(sim.steps[0]) and "closes over 'm'" in sim.launch_note
    sim.exit = lambda s: len(s.ts) >= 3                 # a new function is judged afresh
    assert sim._ahead_agreed(sim.steps[0]) and sim.launch_note is None


def test_a_measurement_clock_is_not_shared_between_ts_and_the_replay():
    """``t += dt`` is IN PLACE when the time step is a Measurement (an ndarray subclass; physicl/__init__.py:343 deep-copies t
    into ts for that reason).  The planner hands out two separate copies per pass -- one for ``ts``, one for the replay, which
    re-installs it as ``sim.t`` -- so advancing the clock for the next launch must not move an entry of ``ts``."""
    import physicl_amd as phys
    sim, _ = _sim(lambda s: len(s.ts) >= 50)
    dt = phys.Measurement(np.double(0.25), "s**1")
    upd = phys.UpdateTimeStep(lambda s: dt)
    times, dt0 = sim._plan_passes(upd, 3, False)
    assert len(times) == 3 and dt0 == 0.25
    sim.t, sim.dt = times[-1]                          # what the row replay leaves behind
    assert all(times[i][0] is not sim.ts[i] for i in range(3))
    times2, _ = sim._plan_passes(upd, 2, False)
    assert [float(np.asarray(t)) for t in sim.ts] == [0.25, 0.5, 0.75, 1.0, 1.25]
    assert [float(np.asarray(t)) for t, _ in times2] == [1.0, 1.25]      # (times[-1] IS the live clock now: it moved on)
    assert len({id(t) for t in sim.ts}) == 5
