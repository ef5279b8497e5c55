"""Where the time of a delete-until-empty run through the plugin API goes (1e7 photons, constructor defaults): device calls,
planning of the passes, and everything else (row replay, exit tests, terminate).  python tools/delsim_split.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys, physicl.light as light, physicl.newton as newton
from physicl_amd import core
n = 10_000_000
for rep in range(4):
    kw = {"steps_per_launch": int(sys.argv[sys.argv.index("--k") + 1])} if "--k" in sys.argv else {}
    sim = phys.Simulation(cl_on=True, seed=7, **kw)
    sim.add_objs(light.generate_photons_bulk(n, min=1.0, max=1.0, seed=7))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(2, light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
    m = light.ScatterMeasureStep(None, True, [[1.0 / (0.001 * 0.001), np.nan, np.nan]])
    sim.add_step(3, m)
    sim._to_device(); sim._dev.sync()
    dev = sim._dev
    spans = []
    orig = dev.step_fused_delete_multi
    def timed(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); spans.append(time.perf_counter() - t0); return r
    dev.step_fused_delete_multi = timed
    plan_t = []
    op = sim._plan_passes
    def tplan(*a, **k):
        t0 = time.perf_counter(); r = op(*a, **k); plan_t.append(time.perf_counter() - t0); return r
    sim._plan_passes = tplan
    if "--thread" in sys.argv:                 # the way a script runs it: start() / join(), time as Simulation.run_time reports it
        t0 = time.perf_counter(); sim.start(); sim.join(); el = time.perf_counter() - t0
        print("  (thread: run_time %.0f us, start..join %.0f us)" % (sim.run_time * 1e6, el * 1e6))
        el = sim.run_time
    else:
        t0 = time.perf_counter(); sim.run(); el = time.perf_counter() - t0
    print("rep %d: run %.0f us; device calls %s us; planning %s us; rest %.0f us; passes %d" % (
        rep, el * 1e6, [round(x * 1e6) for x in spans], [round(x * 1e6) for x in plan_t], (el - sum(spans) - sum(plan_t)) * 1e6, len(sim.ts)))
    sim.close(download=False)
