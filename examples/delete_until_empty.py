#!/usr/bin/env python3
"""Absorption: photons are removed with probability n*A*|dr| per step until none is left, and the number
crossing the plane x = 1/(nA) is compared with N/e (the reference's test/test_light.py:45-66 and
examples/code_unit_scale_test.ipynb), here with the code scale of the metre set to 1e-3.

    python examples/delete_until_empty.py [steps_per_launch]      (default: the constructor's own choice -- exit only asks
                                                                  whether objects are left, so up to 32 passes run per launch)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys          # noqa: E402
phys.Measurement.set_code_scale("m", 0.001)                   # before the light module creates c and h
import physicl.light as light   # noqa: E402
import physicl.newton as newton  # noqa: E402

N = 1_000_000
spl = int(sys.argv[1]) if len(sys.argv) > 1 else None          # passes of the loop per launch (1 = one launch per pass)
sim = phys.Simulation(cl_on=True, seed=7, exit=lambda cond: len(cond.objects) == 0, steps_per_launch=spl)
sim.add_objs(light.generate_photons_bulk(N, min=light.E_from_wavelength(phys.Measurement(700e-9, "m**1")),
                                         max=light.E_from_wavelength(phys.Measurement(200e-9, "m**1")), seed=7))
sim.add_step(0, phys.UpdateTimeStep(lambda s: phys.Measurement(0.00001, "s**1")))
sim.add_step(1, newton.NewtonianKinematicsStep())
n = phys.Measurement(2.0e25, "m**-3")
A = phys.Measurement(5.1e-31, "m**2")
sim.add_step(2, light.ScatterDeleteStep(n, A))
m1 = light.ScatterMeasureStep(None, True, [phys.Measurement([1 / (n * A), np.nan, np.nan], "m**1")])
sim.add_step(3, m1)
sim.start()
sim.join()

idx = int(((1 / (n * A)) / sim.dt / light.c).__unscaled__())
print("steps until empty:", len(m1.data), "  run time %.3f s" % sim.run_time)
print("photons crossing x = 1/(nA) (step %d): %d   expected N/e = %.0f" % (idx, m1.data[idx][2], N / np.e))
