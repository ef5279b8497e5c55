#!/usr/bin/env python3
"""Scattering AND absorption in one loop -- BASELINE.json configs[4]'s step list
[UpdateTime, Newton, ScatterIsotropic, sign rows, Newton, ScatterDelete, plane rows] -- with K whole passes per pass
over the device store and one compaction (pcl_step_mixed_multi); K = 1 runs one launch per light step.  Same rows
either way.

    python examples/mixed_loop.py [photons] [steps_per_launch]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys          # noqa: E402
import physicl.light as light   # noqa: E402
import physicl.newton as newton  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sim = phys.Simulation(seed=11, exit=lambda s: len(s.ts) >= 64, steps_per_launch=K)
sim.add_objs(light.generate_photons_bulk(N, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=11))
sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
sim.add_step(1, newton.NewtonianKinematicsStep())
sim.add_step(2, light.ScatterIsotropicStep(n=np.double(0.001), A=np.double(0.001)))          # pcoll ~ 0.30 per step
sign = light.ScatterSignMeasureStep(None, True)
sim.add_step(3, sign)
sim.add_step(4, newton.NewtonianKinematicsStep())
sim.add_step(5, light.ScatterDeleteStep(np.double(0.00002), np.double(0.001)))               # 0.6 % absorbed per step
planes = light.ScatterMeasureStep(None, True, [np.array([3.0e5, np.nan, np.nan])])
sim.add_step(6, planes)
sim.start()
sim.join()
if sim.error is not None:
    raise sim.error
work = sum(int(r[1]) for r in sign.data) + sum(int(r[1]) for r in planes.data)
print("%d passes, %d photons left of %d, last pass: %d scattered, %d absorbed" % (len(sim.ts), len(sim.objects), N, sim.hits,
                                                                               sim.steps[5].removed))
print("run time %.3f s  ->  %.3g light steps x photons per second" % (sim.run_time, work / sim.run_time))
print("fraction moving along +x after 64 passes: %.4f (all of them at the start; 1/2 once every photon has scattered)"
      % (sign.data[-1][2] / sign.data[-1][1]))
sim.close(download=False)
