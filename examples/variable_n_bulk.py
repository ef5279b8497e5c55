#!/usr/bin/env python3
"""Variable number density + wavelength-dependent scattering -- the step list of the reference's
examples/variable_n_scattering.ipynb:52-60 (UpdateTimeStep, NewtonianKinematicsStep, ScatterSphericalStep, TracePathMeasureStep),
with a sign-count measure beside it -- on 1e8 photons that never exist as Python objects.

    python examples/variable_n_bulk.py [n_photons] [steps_per_launch] [tracked] [passes]

steps_per_launch > 1 (default 25) runs that many passes of the loop per pass over the device store -- same rows,
same final state, ~2.5x the particle-steps/s of one launch per pass.  TracePathMeasureStep follows the first ``tracked``
photons (default 1000, the notebook's population): their positions are worked out on the device ahead of every launch, so
the step costs the run nothing it would notice.
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
tracked = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 500       # the notebook runs to t = 2.5 in steps of 0.005
cl_n = "0.000000001 * exp(r0[gid] - 5)"                      # OpenCL-C expression, compiled into the kernel by hipRTC

sim = phys.Simulation(cl_on=True, seed=1234, exit=lambda cond: cond.t >= 0.005 * (passes - 0.5), steps_per_launch=spl)
sim.add_step(2, phys.UpdateTimeStep(lambda c: 0.005))
sim.add_step(1, newton.NewtonianKinematicsStep())
sim.add_step(3, light.ScatterIsotropicStep(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True,
                                           variable_n=True, variable_n_fn=cl_n))
tp = light.TracePathMeasureStep(None, track=tracked)
sim.add_step(0, tp)
signs = light.ScatterSignMeasureStep(None, True)
sim.add_step(4, signs)
sim.add_objs(light.generate_photons_bulk(n, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=1234))

sim.prepare()                                                 # the photons are created now: run_time below is stepping only
t0 = time.time()
sim.start()
sim.join()
steps = len(sim.ts)
print("%d photons x %d steps in %.2f s  ->  %.3g particle-steps/s" % (n, steps, sim.run_time, n * steps / sim.run_time))
print("launches by formulation:", dict(sim.schedule), sim.launch_note or "")
print("last row [t, N, #vx>0, #vy>0, #vz>0]:", signs.data[-1], "  scattered in the last step:", sim.hits)
path = np.array(tp.data[1][1:]) if len(tp.data) > 1 else np.zeros((1, 3))                               # photon 0: one position per pass (variable_n_scattering.ipynb:132-134)
print("traced %d photons over %d passes; photon 0 went from x = %.3g m to (%.3g, %.3g, %.3g) m, turning %d times in x"
      % (len(tp.data) - 1, len(path), path[0, 0], path[-1, 0], path[-1, 1], path[-1, 2], int((np.diff(np.sign(np.diff(path[:, 0]))) != 0).sum())))
x = sim.download("r")[:, 0]
print("x range after %.3f s: [%.3g, %.3g] m, %.1f %% of the photons have escaped to x < 0" % (float(sim.t), x.min(), x.max(), 100 * (x < 0).mean()))
