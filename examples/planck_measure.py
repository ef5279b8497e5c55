#!/usr/bin/env python3
"""The flow of the reference's examples/planck_distribution.ipynb with its own spellings (``import phys``, dict
constructor, ScatterSphericalStep, TracePathMeasureStep(id_info_fn, trace_dv), ScatterMeasureStep(measure_E=True)):
photons drawn from a 2000 K Planck distribution, wavelength-dependent scattering, and at four distances the energy
spectrum of the photons passing by -- blue is scattered out of the beam first.

    python examples/planck_measure.py [n_photons] [steps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phys                      # noqa: E402   (the reference's older package name)
import phys.light                # noqa: E402
import phys.newton               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 220
T = 2000
np.random.seed(3)
E = [phys.light.planck_phot_distribution(phys.light.E_from_wavelength(500000e-9), phys.light.E_from_wavelength(100e-9), T,
                                         bins=5000) for x in range(n)]
phot = phys.light.generate_photons_from_E(E)

sim = phys.Simulation({"cl_on": True, "exit": lambda cond: cond.t >= 0.0005 * (steps - 0.5)})
sim.add_objs(phot)
sim.add_step(0, phys.UpdateTimeStep(lambda x: 0.0005))
sim.add_step(1, phys.newton.NewtonianKinematicsStep())
sim.add_step(2, phys.light.ScatterSphericalStep(0.00000000000001, 0.000000000000005, wavelength_dep_scattering=True))
sim.add_step(3, phys.light.TracePathMeasureStep(None, id_info_fn=lambda x: str(x.E), trace_dv=True))
sim.add_step(4, phys.light.ScatterMeasureStep(None, measure_n=True,
                                              measure_locs=[[x * (phys.light.c) * 0.0005 * 50, 0, 0] for x in range(1, 5)],
                                              measure_E=True))
sim.start()
sim.join()
assert sim.error is None

trace, planes = sim.steps[3].data, sim.steps[4].data
scatterings = sum(z[1] for z in trace[1:])
print("%d photons, %d steps, %d scatterings in total, run time %.2f s" % (n, len(planes), scatterings, sim.run_time))
for y in range(4):
    k = 50 * (y + 1) - 1                                      # the step in which unscattered photons reach plane y
    if k < len(planes):
        Es = np.array(planes[k][3 + 2 * y], dtype=float)
        print("plane %d (step %d): %4d photons pass, median wavelength %.0f nm"
              % (y + 1, k + 1, len(Es), 1e9 * float(phys.light.wavelength_from_E(np.median(Es))) if len(Es) else float("nan")))
