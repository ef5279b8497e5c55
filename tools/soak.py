"""Full-size runs through the plugin API: BASELINE config 3 (1e8 photons x 500 steps, variable-n scattering) and
config 2(ii) (1e8 photons, delete until empty), K passes per launch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys
import physicl.light as light
import physicl.newton as newton

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
sim = phys.Simulation(cl_on=True, seed=1234, exit=lambda c: c.t >= 2.4995, steps_per_launch=50)
sim.add_step(0, phys.UpdateTimeStep(lambda c: 0.005))
sim.add_step(1, newton.NewtonianKinematicsStep())
sim.add_step(2, light.ScatterIsotropicStep(n=1e-15, A=1e-19, wavelength_dep_scattering=True, variable_n=True,
                                           variable_n_fn="0.000000001 * exp(r0[gid] - 5)"))
sg = light.ScatterSignMeasureStep(None, True)
sim.add_step(3, sg)
sim.add_objs(light.generate_photons_bulk(n, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=1234))
sim.start(); sim.join()
assert sim.error is None and len(sim.ts) == 500 and len(sg.data) == 500
print("config 3: %d photons x %d steps in %.2f s (incl. hipRTC) -> %.3g particle-steps/s; last row %s hits %d"
      % (n, len(sim.ts), sim.run_time, n * len(sim.ts) / sim.run_time, [float(x) for x in sg.data[-1]], sim.hits))
sim.close()

sim = phys.Simulation(cl_on=True, seed=7, steps_per_launch=16)          # default exit: until empty
sim.add_objs(light.generate_photons_bulk(n, min=1.0, max=1.0, seed=7))
sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
sim.add_step(1, newton.NewtonianKinematicsStep())
sim.add_step(2, light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
m = light.ScatterMeasureStep(None, True, [[1.0 / (0.001 * 0.001), np.nan, np.nan]])
sim.add_step(3, m)
sim.start(); sim.join()
assert sim.error is None and len(sim.objects) == 0
alive = [int(r[1]) for r in m.data]
work = n + sum(alive[:-1])
print("config 2(ii): %d photons until empty: %d steps in %.3f s -> %.3g particle-steps/s; crossing plane row %s"
      % (n, len(alive), sim.run_time, work / sim.run_time, [float(x) for x in m.data[3]]))
