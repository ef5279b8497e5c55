#!/usr/bin/env python3
"""Cost of one pcl_store_trace_ahead call (1000 tracked ids, K = 32) beside the K-step launch it precedes, at 1e8 photons."""
import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip as hip
C, H = 299792458.0, 6.62607015e-34
N = 100_000_000
d = hip.Device(0)
d.store_alloc(N)
d.fill_photons(N, 0, C, H * C / 700e-9, H * C / 200e-9, 1)
sc = dict(A=1e-15, n=1e-19, flags=3, c=C, h=H, n_expr="0.000000001 * exp(r0[gid] - 5)", rng_mode=1, seed=1, step=0)
ids = np.arange(1000, dtype=np.int64)
d.step_fused_multi(5e-3, 32, sc)
for rep in range(3):
    d.sync()
    t0 = time.perf_counter()
    for k in range(20):
        rows = d.trace_ahead(ids, 5e-3, 32, ("iso",), 0, sc, None, 1, 32 + k)
    t1 = time.perf_counter()
    print("trace_ahead: %.3f ms per call" % ((t1 - t0) / 20 * 1e3))
d.prof_enable(True)
t0 = time.perf_counter(); d.step_fused_multi(5e-3, 32, dict(sc, step=100)); t1 = time.perf_counter()
print("multi launch %.3f ms" % ((t1 - t0) * 1e3))
