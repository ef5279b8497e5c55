#!/usr/bin/env python3
"""``Simulation(cl_on=False)``: the reference's switch to its Python paths (examples/runtime1.py:21, delete_ex.py:12).
This build has no CPU implementation of the hot path; the switch selects the SEMANTICS of those paths -- the order in
which ScatterIsotropicStep.__run_py consumes np.random (one draw per photon, phi and theta only on a hit), dv = v_old,
and the skip-after-removal iteration of ScatterDeleteStepReference.__run_py -- executed by the HIP kernels.  A seeded
run therefore reproduces what the reference's CPU path computes, and leaves np.random where the reference leaves it.

    python examples/cpu_path_semantics.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phys                      # noqa: E402  (the older package name the shipped examples use)
import phys.light                # noqa: E402
import phys.newton               # noqa: E402


def prep(cl_on, step):
    sim = phys.Simulation(params={"bounds": np.array([1000, 1000, 1000]), "cl_on": cl_on, "exit": lambda cond: cond.t >= 0.0095})
    sim.add_objs(phys.light.generate_photons(2000, bins=1, dist="constant", min=phys.light.E_from_wavelength(200e-9),
                                             max=phys.light.E_from_wavelength(700e-9)))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, step)
    return sim


for cl_on in (False, True):
    np.random.seed(1)
    sim = prep(cl_on, phys.light.ScatterSphericalStep(np.double(0.001), np.double(0.001)))
    sim.start(); sim.join()
    print("isotropic, cl_on=%-5s: %4d of 2000 scattered in the last step; next np.random number %.6f"
          % (cl_on, sim.hits, np.random.random()))
    sim.close(download=False)
    np.random.seed(1)
    sim = prep(cl_on, phys.light.ScatterDeleteStepReference(np.double(0.001), np.double(0.001)))
    sim.exit = lambda cond: len(cond.ts) >= 1
    sim.start(); sim.join()
    print("delete (reference class), cl_on=%-5s: %4d of 2000 left after one step (pcoll = 0.30; the Python path skips the "
          "object behind every removal)" % (cl_on, len(sim.objects)))
    sim.close(download=False)
