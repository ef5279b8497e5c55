"""Where the host time of the plugin API's delete loop goes (bench.py's api.delete_default: BASELINE configs[1](ii) at 1e7
photons, default constructor): cProfile of Simulation.run over a few simulations, after two unprofiled ones."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import physicl_amd as phys
import physicl_amd.light as light
import physicl_amd.newton as newton

Nd = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000


def make():
    sim = phys.Simulation(seed=1234)
    sim.add_objs(light.generate_photons_bulk(Nd, min=1.0, max=1.0, seed=1234))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(2, light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
    m = light.ScatterMeasureStep(None, True, [[1.0 / (1e-3 * 1e-3), np.nan, np.nan]])
    sim.add_step(3, m)
    sim._to_device()
    sim._dev.sync()
    return sim


for i in range(3):
    sim = make(); sim.start(); sim.join(); print("threaded run_time %.3f ms, passes %d" % (sim.run_time * 1e3, len(sim.ts)), sim.schedule); sim.close(download=False)
pr = cProfile.Profile()
tot = 0.0
for i in range(5):
    sim = make()
    t0 = time.perf_counter()
    pr.enable(); sim.run(); pr.disable()                      # the thread's body, on this thread
    tot += time.perf_counter() - t0
    print("run() %.3f ms (profiled), sim.run_time %.3f" % ((time.perf_counter() - t0) * 1e3, sim.run_time * 1e3))
    sim.close(download=False)
rows = sorted(pr.getstats(), key=lambda e: -e.totaltime)
print("per run, microseconds: cumulative | own | calls | function")
for e in rows[:40]:
    c = e.code
    name = c if isinstance(c, str) else "%s:%d(%s)" % (os.path.basename(c.co_filename), c.co_firstlineno, c.co_name)
    print("%9.1f %9.1f %6.1f  %s" % (e.totaltime / 5 * 1e6, e.inlinetime / 5 * 1e6, e.callcount / 5.0, name))
