"""One scatter simulation at 1e8 photons, then two delete-until-empty simulations in the same process: run times of the
latter two (the case that showed the allocation stall recorded in DESIGN.md, "Device memory between stores")."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys
import physicl.light as light
import physicl.newton as newton
n = 100_000_000
sim = phys.Simulation(cl_on=True, seed=1234, exit=lambda c: c.t >= 0.2495, steps_per_launch=50)
sim.add_step(0, phys.UpdateTimeStep(lambda c: 0.005))
sim.add_step(1, newton.NewtonianKinematicsStep())
sim.add_step(2, light.ScatterIsotropicStep(n=1e-15, A=1e-19, wavelength_dep_scattering=True, variable_n=True,
                                           variable_n_fn="0.000000001 * exp(r0[gid] - 5)"))
sg = light.ScatterSignMeasureStep(None, True)
sim.add_step(3, sg)
sim.add_objs(light.generate_photons_bulk(n, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=1234))
sim.start(); sim.join()
print("config 3: %d steps, run_time %.3f" % (len(sim.ts), sim.run_time), flush=True)
t = time.perf_counter(); sim.close(); print("close %.3f s" % (time.perf_counter() - t), flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    sim = phys.Simulation(cl_on=True, seed=7, steps_per_launch=16)
    sim.add_objs(light.generate_photons_bulk(n, min=1.0, max=1.0, seed=7))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(2, light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
    m = light.ScatterMeasureStep(None, True, [[1.0 / (0.001 * 0.001), np.nan, np.nan]])
    sim.add_step(3, m)
    t1 = time.perf_counter()
    sim.start(); sim.join()
    print("rep %d: build %.3f s, run_time %.3f s, passes %d, schedule %s" % (rep, t1 - t0, sim.run_time, len(sim.ts), dict(sim.schedule)), flush=True)
    t = time.perf_counter(); sim.close(); print("close %.3f s" % (time.perf_counter() - t), flush=True)
