#!/usr/bin/env python3
"""The iso_1e7 leg of bench.py alone (BASELINE configs[1](i)), for A/B runs:  python tools/bench_iso.py [photons]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from physicl_amd import _hip  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
with _hip.Device(0) as d:
    r = bench.iso_leg(d, _hip, n, 1234)
print(json.dumps({"photons": n, "per_step_ms": r["per_step"]["roofline"]["avg_launch_ms"], "per_step_frac": r["per_step"]["roofline"]["frac"],
                  "per_step_value": r["per_step"]["value"], "multi_value": r["multi"]["value"], "multi_ms_per_step": r["multi"]["ms_per_step"]}))
