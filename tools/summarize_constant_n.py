#!/usr/bin/env python3
"""VALU instructions per particle-step of the constant-n K-step kernels (tools/prof_constant_n.sh): every K-step dispatch of
tools/bench_iso.py's multi leg, per kernel, on the 128-per-wave form (PCL_MULTI_NQ3=0) and on the default.  Prints
profiles/r06_constant_n_pmc.md."""
import csv
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 10_000_000
STEPS = 400           # bench.iso_leg: one warm-up run + 3 timed runs of 100 steps each


def read(tag):
    d = os.path.join(ROOT, "gpurun_out", "prof_constant_n_" + tag)
    f = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_multi" not in r["Kernel_Name"]:
            continue
        e = per.setdefault(int(r["Dispatch_Id"]), {"kernel": re.search(r"k_multi\w*(<[^>]*>)?", r["Kernel_Name"]).group(0)})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    tr = glob.glob(os.path.join(d, "**", "pmc_kernel_trace.csv"), recursive=True)[0]
    for r in csv.DictReader(open(tr)):
        if int(r["Dispatch_Id"]) in per:
            per[int(r["Dispatch_Id"])]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return [per[k] for k in sorted(per)]


def main():
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
    L = ["# r06: the constant-n K-step kernels' instruction counts (BASELINE configs[1](i): [Newton, ScatterIsotropic(A = n = 1e-3)], 1e7 photons, K = 32 per launch)", "",
         "`tools/prof_constant_n.sh` (rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE around `tools/bench_iso.py 1e7`), commit %s." % commit,
         "Every K-step dispatch of the run's multi leg is summed per kernel; K = 32 for each (the leg runs 100 steps as 32 + 32 + 32 + 4: the K = 4 launch is the smaller rows' share).",
         "VALU instructions / particle-step = SQ_INSTS_VALU x 64 / (N x steps covered); valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024); lane_util = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64).", "",
         "| setting | kernel | dispatches | total ms (trace) | SQ_INSTS_VALU | VALU instr / particle-step | valu_busy | lane_util |", "|---|---|---|---|---|---|---|---|"]
    tot = {}
    for tag, label in (("0", "PCL_MULTI_NQ3=0 (128 photons per wave)"), ("default", "default (192 per wave)")):
        by = {}
        for d in read(tag):
            b = by.setdefault(d["kernel"], {"n": 0, "ms": 0.0, "v": 0.0, "a": 0.0, "t": 0.0, "g": 0.0})
            b["n"] += 1
            b["ms"] += d.get("ms", 0.0)
            b["v"] += d.get("SQ_INSTS_VALU", 0.0)
            b["a"] += d.get("SQ_ACTIVE_INST_VALU", 0.0)
            b["t"] += d.get("SQ_THREAD_CYCLES_VALU", 0.0)
            b["g"] += d.get("GRBM_GUI_ACTIVE", 0.0)
        allv = sum(b["v"] for b in by.values())
        allms = sum(b["ms"] for b in by.values())
        tot[tag] = (allv, allms)
        for k, b in sorted(by.items()):
            L.append("| %s | `%s` | %d | %.3f | %d | - | %.3f | %.3f |" % (label, k, b["n"], b["ms"], b["v"], b["a"] * 4.0 / (b["g"] / 8.0 * 1024.0) if b["g"] else 0.0,
                                                                           b["t"] / (b["a"] * 64.0) if b["a"] else 0.0))
        # the multi leg: 3 warm-up + timed runs of 100 steps each, every one through these kernels: steps covered = total K over the dispatches
        L.append("| %s | all K-step dispatches | %d | %.3f | %d | %.1f | | |" % (label, sum(b["n"] for b in by.values()), allms, allv, allv * 64.0 / (N * STEPS)))
    v0, m0 = tot["0"]
    v1, m1 = tot["default"]
    L += ["", "Both runs launch the same schedule over the same store (same seeds, same counts per step: the forms are bit-identical, tests/test_gpu_multi.py), so the particle-steps covered are equal and the ratio of the sums is the ratio per particle-step:",
          "", "**SQ_INSTS_VALU 192-per-wave / 128-per-wave = %.4f (%.1f %% fewer VALU instructions per particle-step); kernel time %.4f.**" % (v1 / v0, 100.0 * (1.0 - v1 / v0), m1 / m0)]
    out = os.path.join(ROOT, "profiles", "r06_constant_n_pmc.md")
    open(out, "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
