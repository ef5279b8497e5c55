#!/usr/bin/env python3
"""Delete-until-empty, one call per loop body (BASELINE configs[1](ii)): per-body wall time, kernel time and extent.

    python tools/bench_delete_bodies.py --photons 1e7
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip  # noqa: E402

C_LIT = 299792458.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--photons", type=float, default=1e7)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-plane", action="store_true")
    ap.add_argument("--no-counters", action="store_true", help="planes=None: no measure counters at all (timing experiment)")
    ap.add_argument("--no-prof", action="store_true", help="no HIP-event pairs around the kernels (pcl_prof_*)")
    a = ap.parse_args()
    N = int(a.photons)
    dev = _hip.Device(0)
    dev.store_alloc(N)
    plane = np.zeros((0, 3)) if a.no_plane else np.array([[1.0 / (1e-3 * 1e-3), np.nan, np.nan]])
    if a.no_counters:
        plane = None
    for rep in range(a.reps + 1):
        dev.fill_photons(N, 0, C_LIT, 1.0, 1.0, 1234)
        dev.prof_enable(not a.no_prof)
        dev.sync()
        t0 = time.perf_counter()
        bodies, work, k = [], 0, 0
        nb = sb = N
        t1 = time.perf_counter()
        while nb > 0:
            o = dev.step_fused_delete(1e-3, 1e-3, 1e-3, _hip.RNG_PHILOX, 1234, k, plane, lazy=True)
            sa = dev.slots
            t2 = time.perf_counter()
            bodies.append((nb, sb, o["N"], (t2 - t1) * 1e6))
            t1 = t2
            work += nb
            nb, sb = o["N"], sa
            k += 1
        dev.sync()
        el = time.perf_counter() - t0
        kern = {name: dev.prof_read(kid) for kid, name in _hip.PROF_NAMES.items()}
        dev.prof_enable(False)
        if rep == 0:
            continue
        print(json.dumps({"photons": N, "bodies": len(bodies), "ms_total": round(el * 1e3, 3), "value": work / el,
                          "kernels_ms": {n: round(v["total_ms"], 3) for n, v in kern.items() if v["launches"]},
                          "launches": {n: v["launches"] for n, v in kern.items() if v["launches"]},
                          "first_bodies_us": [(nb, sb, round(us, 1)) for nb, sb, na, us in bodies[:12]],
                          "tail_body_us_median": round(float(np.median([us for nb, sb, na, us in bodies[25:]])), 1)}), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
