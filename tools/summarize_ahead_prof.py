#!/usr/bin/env python3
"""Per-dispatch SQ counters of the kernels in gpurun_out/prof_ahead*/ (tools/prof_ahead.sh), biggest dispatches first."""
import collections
import csv
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("prof_ahead", "prof_ahead2"):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "pmc_counter_collection.csv"), recursive=True)
    if not f:
        continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], int(r["Grid_Size"]))
        per.setdefault(key, {})
        per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = [kv for kv in per.items() if "ahead" in kv[0][1] or "compact" in kv[0][1]]
    rows = sorted(rows, key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", kv[1].get("SQ_BUSY_CYCLES", 0)))[:4]
    for (did, k, g), c in rows:
        print(did, k, "grid", g)
        print("   ", {n: round(v) for n, v in c.items()})
        if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
            print("    valu_busy %.3f  lane_util %.3f  wait_inst_any/wave_cycles %.3f  VALU %.3g SALU %.3g" % (
                c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64),
                c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"]))
