#!/usr/bin/env python3
"""Config 2(ii) of BASELINE.json: Newton + ScatterDeleteStep (+ plane counter) until few photons are left.
Prints per-kernel timings and effective GB/s of the delete/compaction passes (HIP events, pcl_prof_*).

    python tools/bench_delete.py --photons 1e7 [--steps 12]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip  # noqa: E402

C_LIT, H_LIT = 299792458.0, 6.62607015e-34


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--photons", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--mode", choices=("multi", "fused", "fused-eager", "separate"), default="multi",
                    help="multi: --steps-per-launch loop bodies per pass and one compaction (pcl_step_fused_delete_multi)")
    ap.add_argument("--steps-per-launch", type=int, default=8)
    ap.add_argument("--no-counters", action="store_true", help="multi mode: alive counts only (timing experiment)")
    a = ap.parse_args()
    N = int(a.photons)
    dev = _hip.Device(0)
    dev.store_alloc(N)
    plane = [[1.0 / (1e-3 * 1e-3), np.nan, np.nan]]          # test/test_light.py:58
    rows = []
    for rep in range(2):                                      # rep 0 = warm-up (allocations, first touch)
        dev.fill_photons(N, 0, C_LIT, 1.0, 1.0, a.seed)
        dev.prof_enable(True)
        dev.sync()
        t0 = time.perf_counter()
        work = 0
        per_step = []
        k = 0
        while a.mode == "multi" and k < a.steps:
            ks = min(a.steps_per_launch, a.steps - k)
            n_before = dev.count
            for o in dev.step_fused_delete_multi(1e-3, ks, 1e-3, 1e-3, a.seed, k, None if a.no_counters else plane):
                work += n_before
                per_step.append((n_before, o["N"]))
                n_before = o["N"]
            k += ks
        for k in range(a.steps if a.mode != "multi" else 0):
            n_before = dev.count
            if a.mode == "separate":
                dev.step_newton(1e-3)
                alive, removed = dev.step_scatter_delete(1e-3, 1e-3, _hip.RNG_PHILOX, a.seed, k)
                cnt = dev.step_counters(plane)
            else:
                alive = dev.step_fused_delete(1e-3, 1e-3, 1e-3, _hip.RNG_PHILOX, a.seed, k, plane,
                                              lazy=(a.mode == "fused"))["N"]
            work += n_before
            per_step.append((n_before, alive))
        dev.sync()
        el = time.perf_counter() - t0
        kern = {name: dev.prof_read(kid) for kid, name in _hip.PROF_NAMES.items()}
        dev.prof_enable(False)
    s_mean = np.mean([al / nb for nb, al in per_step if nb])
    out = {"workload": "config 2(ii): Newton + ScatterDelete(A=n=1e-3) + plane counter, %d photons, %d steps" % (N, a.steps),
           "particle_steps_per_s": work / el, "ms_total": el * 1e3, "survivor_fraction": s_mean,
           "alive_per_step": [al for _, al in per_step],
           "kernels_total_ms": {k: round(v["total_ms"], 4) for k, v in kern.items() if v["launches"]},
           "first_step_ms": None}
    # effective bandwidth of the three delete passes over the whole run (bytes from DESIGN.md section 4)
    tot = sum(nb for nb, _ in per_step)
    surv = sum(al for _, al in per_step)
    nf = {"separate": 13, "fused-eager": 13, "fused": 10, "multi": 10}[a.mode]             # 8-byte fields moved per survivor (+ ids)
    out["mode"] = a.mode
    out["GBps"] = {
        "k_compact(algorithmic: mask bit + survivors read+written)": (tot * 0.125 + surv * 2 * 8 * (nf + 1)) /
                                                                     (kern["k_compact"]["total_ms"] * 1e-3) / 1e9}
    if a.mode == "multi":
        out["steps_per_launch"] = a.steps_per_launch
        out["GBps"] = {}
    elif a.mode == "separate":
        out["GBps"]["k_newton"] = tot * 96 / (kern["k_newton"]["total_ms"] * 1e-3) / 1e9
        out["GBps"]["k_delete_mask"] = tot * 24.125 / (kern["k_delete_mask"]["total_ms"] * 1e-3) / 1e9
    else:
        out["GBps"]["k_newton_mask"] = tot * (72.125 if a.mode == "fused" else 96.125) / \
            (kern["k_delete_mask"]["total_ms"] * 1e-3) / 1e9
    print(json.dumps(out))
    dev.close()


if __name__ == "__main__":
    main()
