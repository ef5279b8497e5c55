"""Host overhead per pass of the Simulation loop at small N (1e4 photons, 1000 passes), one pass per launch vs 32."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import physicl as phys, physicl.light, physicl.newton
def build(K, n=10000, steps=1000):
    sim = phys.Simulation(cl_on=True, seed=1, exit=lambda s: s.t >= 0.001 * (steps - 0.5), steps_per_launch=K)
    sim.add_objs(phys.light.generate_photons_bulk(n, min=1.0, max=1.0, seed=1))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    sg = phys.light.ScatterSignMeasureStep(None, True)
    sim.add_step(3, sg)
    return sim
for K in (1, 32):
    sim = build(K); sim.run(); sim.close()      # warm
    sim = build(K)
    t0 = time.perf_counter(); sim.run(); el = time.perf_counter() - t0
    print("K=%d: %d passes in %.3f s -> %.1f us/pass" % (K, len(sim.ts), el, el / len(sim.ts) * 1e6))
    sim.close()
sim = build(1, steps=300)
pr = cProfile.Profile(); pr.enable(); sim.run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
