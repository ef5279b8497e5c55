#!/usr/bin/env python3
"""Several GPUs from ONE process: the variable-n scattering example (examples/variable_n_scattering.ipynb of the
reference) with the photons sharded by index over the devices named on the command line.  No launcher, no
torch.distributed: the simulation thread fans every launch out to one library context per device and sums the
counters it gets back, so this script -- or a notebook cell -- is all there is.  ``0 0`` puts two contexts on one GPU.

    python examples/multi_device.py 0 1 2 3        (default: 0 0)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys          # noqa: E402
import physicl.light as light   # noqa: E402
import physicl.newton as newton  # noqa: E402

devices = [int(a) for a in sys.argv[1:]] or [0, 0]
N = 4_000_000 * len(devices)
sim = phys.Simulation(devices=devices, seed=1234, exit=lambda cond: cond.t >= 0.2495)
sim.add_objs(light.generate_photons_bulk(N, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=1234))
sim.add_step(0, phys.UpdateTimeStep(lambda s: 0.005))
sim.add_step(1, newton.NewtonianKinematicsStep())
sim.add_step(2, light.ScatterSphericalStep(0.000000000000001, 0.0000000000000000001, wavelength_dep_scattering=True,
                                           variable_n=True, variable_n_fn="0.000000001 * exp(r0[gid] - 5)"))
m = light.ScatterSignMeasureStep(None, True)
sim.add_step(3, m)
sim.start()
while sim.running or not sim.ts:            # the reference's notebooks poll get_state() like this
    time.sleep(0.05)
    print(sim.get_state())
    if not sim.is_alive():
        break
sim.join()
print("%d photons on %d contexts, %d passes in %.3f s (%s); last row %s" % (N, len(devices), len(sim.ts), sim.run_time, dict(sim.schedule),
                                                                           [float(x) for x in m.data[-1]]))
x = sim.download("r")[:, 0]
print("x range of the photons: %.3g .. %.3g m" % (x.min(), x.max()))
