#!/usr/bin/env python3
"""Calibration runs for the instruction counts of k_delete_ahead_live (profiles/isa_counts.json, "k_delete_ahead_live"):
one launch per case on a fresh store of 1e8 photons -- K bodies, first step even or odd -- and the kernel's own work tally
of that launch (pcl_store_ahead_work).  Run under rocprofv3 --pmc SQ_INSTS_VALU (tools/prof_calib_ahead.sh); the fit is
tools/summarize_calib_ahead.py.

    python tools/calib_ahead.py [photons]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip  # noqa: E402

C_LIT = 299792458.0
CASES = [(2, 0), (4, 0), (8, 0), (12, 0), (16, 0), (3, 1), (12, 1), (7, 0), (6, 1), (2, 1), (32, 0), (64, 0), (33, 1), (64, 1)]


def main():
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    dev = _hip.Device(0)
    dev.store_alloc(N)
    plane = np.array([[1.0 / (1e-3 * 1e-3), np.nan, np.nan]])
    for K, step0 in CASES:
        _hip.set_knob("PCL_AHEAD_K_BIG", str(K))
        _hip.set_knob("PCL_AHEAD_MAX_SLOTS", "0")                       # (any size takes the big stores' form: K as asked)
        dev.fill_photons(N, 0, C_LIT, 1.0, 1.0, 1234)
        w0, l0 = dev.ahead_work(), dev.ahead_stats()[0]
        o = dev.step_fused_delete(1e-3, 1e-3, 1e-3, _hip.RNG_PHILOX, 1234, step0, plane, lazy=True)
        w1, l1 = dev.ahead_work(), dev.ahead_stats()[0]
        assert l1 == l0 + 1
        print(json.dumps({"K": K, "step0": step0, "slots": N, "clock_GHz": round(dev.ahead_clock(), 4), "groups_two": w1[0] - w0[0], "groups_one": w1[1] - w0[1], "rounds_two": w1[2] - w0[2], "rounds_one": w1[3] - w0[3],
                          "alive_after_first": o["N"]}), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
