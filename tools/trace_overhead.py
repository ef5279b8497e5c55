#!/usr/bin/env python3
"""Where the time of a device-traced run goes: the example's four steps at 1e8 photons x 500 passes, with the calls the
TracePathMeasureStep adds (pcl_store_trace_ahead per launch, the flush at terminate) timed on their own.
    python tools/trace_overhead.py [n_photons] [tracked]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys          # noqa: E402
import physicl.light as light   # noqa: E402
import physicl.newton as newton  # noqa: E402
from physicl_amd import _hip    # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
tracked = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
acc = {"trace_ahead": 0.0, "terminate": 0.0, "multi": 0.0, "calls": 0}


def timed(fn, key):
    def wrapper(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            acc[key] += time.perf_counter() - t0
            acc["calls"] += key == "trace_ahead"
    return wrapper


_hip.Device.trace_ahead = timed(_hip.Device.trace_ahead, "trace_ahead")
_hip.Device.step_fused_multi = timed(_hip.Device.step_fused_multi, "multi")
light.TracePathMeasureStep.terminate = timed(light.TracePathMeasureStep.terminate, "terminate")
for rep in range(3):
    for k in acc:
        acc[k] = 0
    sim = phys.Simulation(cl_on=True, seed=1234, exit=lambda cond: cond.t >= 0.005 * 499.5)
    sim.add_step(2, phys.UpdateTimeStep(lambda c: 0.005))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(3, light.ScatterSphericalStep(0.000000000000001, 0.0000000000000000001, wavelength_dep_scattering=True, variable_n=True,
                                               variable_n_fn="0.000000001 * exp(r0[gid] - 5)"))
    if tracked:
        sim.add_step(0, light.TracePathMeasureStep(None, track=tracked))
    sim.add_objs(light.generate_photons_bulk(n, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=1234))
    sim.prepare()
    sim.start()
    sim.join()
    print("run %.1f ms: %d launches, K-step calls %.1f ms, trace_ahead %.2f ms, terminate %.2f ms, rest (planning, replay, Python) %.1f ms"
          % (sim.run_time * 1e3, acc["calls"], acc["multi"] * 1e3, acc["trace_ahead"] * 1e3, acc["terminate"] * 1e3,
             (sim.run_time - acc["multi"] - acc["trace_ahead"] - acc["terminate"]) * 1e3), flush=True)
    sim.close(download=False)
