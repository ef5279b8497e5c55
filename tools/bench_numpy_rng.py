#!/usr/bin/env python3
"""rng="numpy" (host-drawn randoms in the reference's order) at bulk size: seconds per pass of
[UpdateTime, Newton, ScatterIsotropic] and of [UpdateTime, Newton, ScatterDelete], and where the global np.random
stream stands afterwards (must not depend on how the numbers were handed to the device).

    python tools/bench_numpy_rng.py [photons] ;  PCL_RAND3=0 python tools/bench_numpy_rng.py   (the three-array path)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys  # noqa: E402
import physicl.light  # noqa: E402
import physicl.newton  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
out = {"photons": n, "rand3": os.environ.get("PCL_RAND3", "1")}
for kind in ("iso", "delete"):
    np.random.seed(4321)
    T = 4
    sim = phys.Simulation(rng="numpy", exit=lambda s: len(s.ts) >= T)
    sim.add_objs(phys.light.generate_photons_bulk(n, min=1.0, max=1.0, seed=1))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    if kind == "iso":
        sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    else:
        sim.add_step(2, phys.light.ScatterDeleteStep(np.double(0.001), np.double(0.0001)))
    m = phys.light.ScatterSignMeasureStep(None, True)
    sim.add_step(3, m)
    sim._to_device()
    sim._dev.sync()
    t0 = time.perf_counter()
    sim.run()
    el = time.perf_counter() - t0
    out[kind] = {"s_per_pass": el / T, "particle_steps_per_s": n * T / el, "rows": [[float(x) for x in r] for r in m.data][-1],
                 "hits": int(sim.hits), "next_random": float(np.random.random_sample())}
    sim.close(download=False)
print(json.dumps(out))
