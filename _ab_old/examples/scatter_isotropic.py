#!/usr/bin/env python3
"""Isotropic scattering of 1e4 photons -- the set-up of the reference's test/test_light.py:27-43,
written exactly as a PhysiCL user would (only the import resolves to the MI355X build).

    python examples/scatter_isotropic.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys          # noqa: E402
import physicl.light            # noqa: E402
import physicl.newton           # noqa: E402

sim = phys.Simulation(bounds=np.array([1000, 1000, 1000]), cl_on=True, exit=lambda cond: cond.t >= 0.100)
for i in range(10000):
    sim.add_obj(phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1)))

sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
sim.add_step(1, phys.newton.NewtonianKinematicsStep())
sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
signs = phys.light.ScatterSignMeasureStep(None, True)
sim.add_step(3, signs)

sim.start()
while sim.running:
    time.sleep(0.05)
    print(sim.get_state())
sim.join()

mean_xp = sum(row[2] for row in signs.data) / len(signs.data)
print("steps: %d   run time: %.3f s   mean #(v_x > 0): %.0f of %d" % (len(signs.data), sim.run_time, mean_xp, signs.data[0][1]))
print("first photon: r =", sim.objects[0].r, " v =", sim.objects[0].v)
