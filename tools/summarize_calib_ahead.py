#!/usr/bin/env python3
"""Fit of k_delete_ahead_live's VALU instruction counts to its own work tally: SQ_INSTS_VALU of each launch of
tools/calib_ahead.py (gpurun_out/prof_calib_ahead, tools/prof_calib_ahead.sh) = a2 x groups_two + a1 x groups_one + b x rounds_two
+ c x rounds_one (least squares over the cases).  Writes the four counts, the fit's residuals and the hash of the device sources
they belong to into profiles/isa_counts.json under "k_delete_ahead_live<double>" and prints the table (profiles/r06_calib_ahead.md)."""
import csv
import glob
import json
import os
import subprocess

import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cases = [json.loads(ln) for ln in open(os.path.join(ROOT, "gpurun_out", "calib_ahead.jsonl")) if ln.strip().startswith("{")]
    f = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_calib_ahead", "**", "pmc_counter_collection.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_delete_ahead_live" not in r["Kernel_Name"] or int(r["Grid_Size"]) < 4_000_000:
            continue
        d = per.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    tr = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_calib_ahead", "**", "pmc_kernel_trace.csv"), recursive=True)[0]
    for r in csv.DictReader(open(tr)):                 # the launch's duration: the trace's own timestamps (ns)
        if int(r["Dispatch_Id"]) in per:
            per[int(r["Dispatch_Id"])]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    disp = [per[k] for k in sorted(per)]
    assert len(disp) == len(cases), (len(disp), len(cases))
    A = np.array([[c["groups_two"], c["groups_one"], c["rounds_two"], c["rounds_one"]] for c in cases], dtype=np.float64)
    y = np.array([d["SQ_INSTS_VALU"] for d in disp])
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    fit = A @ coef
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    L = ["# `k_delete_ahead_live<double>`: VALU wave-instructions against the kernel's own work tally", "",
         "One launch per case on a fresh store of 1e8 photons (`tools/calib_ahead.py` under `rocprofv3 --pmc SQ_INSTS_VALU`, commit %s)." % commit,
         "Model: SQ_INSTS_VALU = a2 x groups of 128 slots loaded whose first pass (decided where the photons are loaded) takes two bodies + a1 x groups whose",
         "first pass takes one + b x rounds of 64 listed photons deciding two bodies + c x rounds deciding one;",
         "least squares over the cases: **a2 = %.1f, a1 = %.1f, b = %.1f, c = %.1f** wave-instructions." % tuple(coef), "",
         "| K | first step | groups (first pass: two) | groups (first pass: one) | rounds (two bodies) | rounds (one body) | SQ_INSTS_VALU | model | model / measured | SQ_INSTS_SALU | duration ms (trace) | GRBM_GUI_ACTIVE / 8 / duration (GHz) |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for c, d, m in zip(cases, disp, fit):
        L.append("| %d | %d | %d | %d | %d | %d | %.4g | %.4g | %.4f | %.4g | %.3f | %.2f |" % (
            c["K"], c["step0"], c["groups_two"], c["groups_one"], c["rounds_two"], c["rounds_one"], d["SQ_INSTS_VALU"], m, m / d["SQ_INSTS_VALU"], d.get("SQ_INSTS_SALU", 0),
            d["ms"], d.get("GRBM_GUI_ACTIVE", 0) / 8 / (d["ms"] * 1e-3) / 1e9))
    from physicl_amd import build
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    price = isa.get("aot", {}).get("k_delete_ahead_live<double, false>", {}).get("cycles_per_valu", 0.0)
    L += ["", "Priced (tools/isa_count.py --aot: %.2f cycles per instruction of the kernel's static mix; available = GRBM_GUI_ACTIVE / 8 x 1024 SIMDs; busy = "
          "SQ_ACTIVE_INST_VALU x 4), case by case:" % price, "",
          "| K | first step | priced / available | busy / available | GHz the kernel measured (s_memtime / s_memrealtime) |", "|---|---|---|---|---|"]
    for c, d in zip(cases, disp):
        avail = d.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024
        if avail:
            L.append("| %d | %d | %.3f | %.3f | %s |" % (c["K"], c["step0"], d["SQ_INSTS_VALU"] * price / avail, d.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / avail,
                                                   c.get("clock_GHz", "-")))
    open(os.path.join(ROOT, "profiles", "r06_calib_ahead.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(L))
    p = os.path.join(ROOT, "profiles", "isa_counts.json")
    j = json.load(open(p))
    j["k_delete_ahead_live<double>"] = {"valu_per_group_first_pass_two_bodies": round(float(coef[0]), 1), "valu_per_group_first_pass_one_body": round(float(coef[1]), 1),
                                        "valu_per_round_two_bodies": round(float(coef[2]), 1), "valu_per_round_one_body": round(float(coef[3]), 1), "max_relative_residual": round(float(np.max(np.abs(fit / y - 1))), 4),
                                        "source": "profiles/r06_calib_ahead.md (SQ_INSTS_VALU of %d launches, least squares on the kernel's own tally)" % len(cases) + "", "commit": commit,
                                        # (the sources the launches ran: this script is run on the tree that was profiled)
                                        "csrc_sha": build.csrc_sha()}
    json.dump(j, open(p, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
