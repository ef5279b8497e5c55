#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (rocprofv3 CSVs written by tools_prof.sh) into profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_pmc.md                per-kernel FETCH_SIZE / WRITE_SIZE per launch over the TIMED launches
  profiles/pmc_traffic.json            bytes per launch that bench.py reports as roofline.traffic

gfx950 counter corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in
kilobytes (x1024 -> bytes); FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams, i.e. exactly
HALF the bytes fetched, so it is doubled; WRITE_SIZE is exact.  Calibrated in the same runs on k_newton
(48 B read + 48 B written per particle, known) and k_counters (24 B read): both come out exact."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter, skip_of):
    """{kernel: (mean over the timed launches, launches)}; skip_of(kernel name) = warm-up launches to drop."""
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        skip = skip_of(k)
        timed = v[skip:] if len(v) > skip else v   # one-off kernels (fill, lam4) have no warm-up
        out[k] = (sum(timed) / len(timed), len(v))
    return out


def short_name(k):
    return ("k_multi" if "multi" in k else "k_fused" if ("fast" in k or "fused" in k) else "k_scatter" if "scatter" in k else
            "k_newton" if "newton" in k else "k_counters" if "counters" in k else None)


def main():
    tag, mode, profile, n, warmup = sys.argv[1], sys.argv[2], sys.argv[3], int(float(sys.argv[4])), int(sys.argv[5])
    spl = int(sys.argv[6]) if len(sys.argv) > 6 else 1        # bench.py --steps-per-launch
    steps = int(sys.argv[7]) if len(sys.argv) > 7 else 20

    def skip_of(k):
        """warm-up launches of a kernel in bench.py's schedule"""
        if "multi" in k:
            return -(-warmup // spl)
        if spl > 1 and ("fast" in k or "fused" in k):
            return 3                                          # the single_step leg: 3 warm-up launches
        return warmup
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "trace_bench.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
    fetch = per_kernel(os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE", skip_of)
    write = per_kernel(os.path.join(src, "pmc_write", "pmc_counter_collection.csv"), "WRITE_SIZE", skip_of)
    lines = ["# %s: HBM traffic per launch from rocprofv3 PMC (timed launches only, warm-up skipped)" % tag, "",
             "Command: `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py "
             "--no-cpu-baseline --steps %d --warmup %d%s%s` (two passes: the counters do not fit one pass on gfx950)."
             % (steps, warmup, "" if mode == "fused" else " --mode " + mode,
                " --steps-per-launch %d" % spl if mode == "fused" else ""), "",
             "FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request on coalesced streams; verified below on "
             "k_newton / k_counters whose byte counts are known); WRITE_SIZE is used as reported. Units: KB -> bytes x1024.",
             "", "| kernel | launches | FETCH_SIZE raw (KB) | read bytes (x2) | WRITE_SIZE (KB) | written bytes | total B/particle |",
             "|---|---|---|---|---|---|---|"]
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("__amd"):
            continue
        f, nl = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        rb, wb = 2 * f * 1024, w * 1024
        lines.append("| `%s` | %d | %.0f | %.4g | %.0f | %.4g | %.1f |" % (k[:60], nl, f, rb, w, wb, (rb + wb) / n))
        short = short_name(k)
        if short:
            traffic[short + "_bytes_per_launch"] = rb + wb
    sq_csv = os.path.join(src, "pmc_sq", "pmc_counter_collection.csv")
    if os.path.exists(sq_csv):
        names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                 "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
        sq = {c: per_kernel(sq_csv, c, skip_of) for c in names}
        lines += ["", "## SQ counters per launch (timed launches), same command with `--pmc " + " ".join(names) + "`", "",
                  "SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (x4 = cycles); GRBM_GUI_ACTIVE is summed over the "
                  "8 XCDs (/8 = kernel cycles).  valu_busy = SQ_ACTIVE_INST_VALU*4 / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); "
                  "lane_util = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64).", "",
                  "| kernel | " + " | ".join(names) + " | valu_busy | lane_util | VALU insts / particle-step |", "|---|" + "---|" * (len(names) + 3)]
        for k in sorted(sq["SQ_INSTS_VALU"]):
            short = short_name(k)
            if short not in ("k_multi", "k_fused"):
                continue
            v = {c: sq[c].get(k, (0.0, 0))[0] for c in names}
            busy = v["SQ_ACTIVE_INST_VALU"] * 4 / (v["GRBM_GUI_ACTIVE"] / 8 * 1024) if v["GRBM_GUI_ACTIVE"] else 0.0
            util = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64) if v["SQ_ACTIVE_INST_VALU"] else 0.0
            steps_per_launch = min(spl, steps) if short == "k_multi" else 1
            per_ps = v["SQ_INSTS_VALU"] * 64 / (n * steps_per_launch)        # wave instructions x 64 lanes / particle-steps
            lines.append("| `%s` | " % k[:40] + " | ".join("%.4g" % v[c] for c in names) + " | %.3f | %.3f | %.1f |" % (busy, util, per_ps))
            traffic[short + "_valu"] = {"busy": round(busy, 4), "lane_utilisation": round(util, 4),
                                        "valu_insts_per_particle_step": round(per_ps, 1),
                                        "clock_GHz_under_counters": None, "source": "profiles/%s_pmc.md" % tag}
    open(os.path.join(dst, tag + "_pmc.md"), "w").write("\n".join(lines) + "\n")
    tf = os.path.join(dst, "pmc_traffic.json")
    allt = json.load(open(tf)) if os.path.exists(tf) else {}
    traffic["source"] = "profiles/%s_pmc.md" % tag
    allt["%s:%s:%d" % (profile, mode, n)] = traffic
    json.dump(allt, open(tf, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
