"""cProfile of delete-until-empty through the plugin API at bulk size (default 1e8 photons, steps_per_launch 16)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys
import physicl.light as light
import physicl.newton as newton
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
K = (None if sys.argv[2] == "auto" else int(sys.argv[2])) if len(sys.argv) > 2 else 16
for rep in range(3):
    sim = phys.Simulation(cl_on=True, seed=7, steps_per_launch=K)
    sim.add_objs(light.generate_photons_bulk(n, min=1.0, max=1.0, seed=7))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(2, light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
    m = light.ScatterMeasureStep(None, True, [[1.0 / (0.001 * 0.001), np.nan, np.nan]])
    sim.add_step(3, m)
    sim._to_device()
    sim._dev.sync()
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.runcall(sim.run)
    el = time.perf_counter() - t0
    print("rep %d: %d passes in %.4f s; schedule %s" % (rep, len(sim.ts), el, dict(sim.schedule)))
    sim.close()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
