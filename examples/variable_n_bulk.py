#!/usr/bin/env python3
"""Variable number density + wavelength-dependent scattering (the physics of the reference's
examples/variable_n_scattering.ipynb) on 1e8 photons that never exist as Python objects.

    python examples/variable_n_bulk.py [n_photons] [steps_per_launch]

steps_per_launch > 1 (default 25) runs that many passes of the loop per pass over the device store -- same rows,
same final state, ~2.5x the particle-steps/s of one launch per pass.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys          # noqa: E402
import physicl.light as light   # noqa: E402
import physicl.newton as newton  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
spl = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cl_n = "0.000000001 * exp(r0[gid] - 5)"                      # OpenCL-C expression, compiled into the kernel by hipRTC

sim = phys.Simulation(cl_on=True, seed=1234, exit=lambda cond: cond.t >= 0.2495, steps_per_launch=spl)
sim.add_step(2, phys.UpdateTimeStep(lambda c: 0.005))
sim.add_step(1, newton.NewtonianKinematicsStep())
sim.add_step(3, light.ScatterIsotropicStep(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True,
                                           variable_n=True, variable_n_fn=cl_n))
signs = light.ScatterSignMeasureStep(None, True)
sim.add_step(0, signs)
sim.add_objs(light.generate_photons_bulk(n, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=1234))

t0 = time.time()
sim.start()
sim.join()
steps = len(sim.ts)
print("%d photons x %d steps in %.2f s  ->  %.3g particle-steps/s" % (n, steps, sim.run_time, n * steps / sim.run_time))
print("last row [t, N, #vx>0, #vy>0, #vz>0]:", signs.data[-1], "  scattered in the last step:", sim.hits)
x = sim.download("r")[:, 0]
print("x range after %.3f s: [%.3g, %.3g] m, %.1f %% of the photons have escaped to x < 0" % (float(sim.t), x.min(), x.max(), 100 * (x < 0).mean()))
