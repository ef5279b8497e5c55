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

ROOT = os.path.dirname(os.path.abspath(__file__))


def per_kernel(path, counter, skip_launches):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        timed = v[skip_launches:] if len(v) > skip_launches else v   # one-off kernels (fill, lam4) have no warm-up
        out[k] = (sum(timed) / len(timed), len(v))
    return out


def main():
    tag, mode, profile, n, warmup = sys.argv[1], sys.argv[2], sys.argv[3], int(float(sys.argv[4])), int(sys.argv[5])
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "trace_bench.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
    fetch = per_kernel(os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE", warmup)
    write = per_kernel(os.path.join(src, "pmc_write", "pmc_counter_collection.csv"), "WRITE_SIZE", warmup)
    lines = ["# %s: HBM traffic per launch from rocprofv3 PMC (timed launches only, warm-up skipped)" % tag, "",
             "Command: `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py "
             "--no-cpu-baseline --steps 20 --warmup %d%s` (two passes: the counters do not fit one pass on gfx950)."
             % (warmup, "" if mode == "fused" else " --mode " + mode), "",
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
        short = ("k_fused" if ("fast" in k or "fused" in k) else "k_scatter" if "scatter" in k else
                 "k_newton" if "newton" in k else "k_counters" if "counters" in k else None)
        if short:
            traffic[short + "_bytes_per_launch"] = rb + wb
    open(os.path.join(dst, tag + "_pmc.md"), "w").write("\n".join(lines) + "\n")
    tf = os.path.join(dst, "pmc_traffic.json")
    allt = json.load(open(tf)) if os.path.exists(tf) else {}
    traffic["source"] = "profiles/%s_pmc.md" % tag
    allt["%s:%s:%d" % (profile, mode, n)] = traffic
    json.dump(allt, open(tf, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
