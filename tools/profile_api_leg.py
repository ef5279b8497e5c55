"""cProfile of the bench's api leg: Simulation(steps_per_launch=32), 1e8 photons, exit at t >= steps * dt."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl_amd as phys
import physicl_amd.light as light
import physicl_amd.newton as newton
from bench import PROFILES
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prof = PROFILES["example"]
for rep in range(3):
    sim = phys.Simulation(cl_on=True, seed=1234, steps_per_launch=32, exit=lambda c: len(c.ts) >= steps)
    sim.add_objs(light.generate_photons_bulk(100_000_000, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=1234))
    sim.add_step(0, phys.UpdateTimeStep(lambda c: prof["dt"]))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(2, light.ScatterIsotropicStep(n=1e-15, A=1e-19, wavelength_dep_scattering=True, variable_n=True, variable_n_fn=prof["expr"]))
    sim.add_step(3, light.ScatterSignMeasureStep(None, True))
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.runcall(sim.run)
    el = time.perf_counter() - t0
    print("rep %d: %d passes in %.4f s -> %.3g particle-steps/s" % (rep, len(sim.ts), el, 1e8 * len(sim.ts) / el), flush=True)
    sim.close()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
