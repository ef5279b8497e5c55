#!/usr/bin/env python3
"""Per-dispatch table of the compaction kernels from tools/prof_compact.sh's three passes (gpurun_out/prof_compact):
duration (trace pass), FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes: x 1024; FETCH x 2 is the gfx950 correction of
MI355X_MICROARCH.md for 16-byte loads), and the bytes the compaction has to move.

    python tools/summarize_compact_prof.py [dir] > profiles/r04_compact_pmc.md
"""
import csv
import json
import os
import sys

D = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_compact"
WANT = ("k_compact_lds", "k_compact_count", "k_delete_alive", "k_tile_scan", "k_flag_mask2", "k_small_delete")


def dispatches(path, counter=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out = []
    for r in rows:
        name = r["Kernel_Name"]
        if not any(w in name for w in WANT):
            continue
        short = name.split("(")[1 if name.startswith("(anonymous") else 0]
        short = name.replace("(anonymous namespace)::", "").split("(")[0]
        out.append({"name": short, "grid": int(r.get("Grid_Size_X", r.get("Grid_Size", 0))),
                    "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                    "val": float(r["Counter_Value"]) if counter else None})
    return out


def main():
    tr = dispatches(os.path.join(D, "trace", "trace_kernel_trace.csv"))
    fe = dispatches(os.path.join(D, "pmc_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
    wr = dispatches(os.path.join(D, "pmc_write", "pmc_counter_collection.csv"), "WRITE_SIZE")
    assert [d["name"] for d in tr] == [d["name"] for d in fe] == [d["name"] for d in wr], "the three passes launched different kernels"
    info = json.loads(open(os.path.join(D, "trace.json")).read().strip().splitlines()[-1])
    print("# compaction dispatches of delete-until-empty, one call per loop body, %d photons\n" % info["photons"])
    print("`tools/prof_compact.sh` (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE -- separate runs of "
          "`tools/bench_delete_bodies.py --reps 1 --no-prof`; both repetitions of the tool are listed).  HBM bytes = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024.  "
          "A compaction's grid = one 256-thread workgroup per 2048-slot tile, so slots = grid x 8.  Of the two pass-3 kernels "
          "enqueued per compaction the one the scan did not choose returns at once (rows under 8 us are those).\n")
    print("| # | kernel | slots | duration us | FETCH x2 MB | WRITE MB | HBM B / slot | HBM GB/s | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    for i, (t, f, w) in enumerate(zip(tr, fe, wr)):
        if "compact" not in t["name"] or t["us"] < 8.0:
            continue
        slots = t["grid"] * 8
        fb, wb = f["val"] * 2 * 1024, w["val"] * 1024
        print("| %d | `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.0f | %.3f |" % (
            i, t["name"], slots, t["us"], fb / 1e6, wb / 1e6, (fb + wb) / slots, (fb + wb) / t["us"] / 1e3,
            (fb + wb) / t["us"] / 1e3 / 8000))
    print("\n## every dispatch of the second repetition in order (trace pass)\n")
    last = max(i for i, d in enumerate(tr) if d["grid"] == tr[0]["grid"] and d["name"] == tr[0]["name"])
    # second repetition = from the last full-extent k_delete_alive of the first body on
    firsts = [i for i, d in enumerate(tr) if "k_delete_alive" in d["name"] and d["grid"] == max(x["grid"] for x in tr if "k_delete_alive" in x["name"])]
    start = firsts[len(firsts) // 2] if len(firsts) > 1 else 0
    print("| # | kernel | grid (threads) | duration us | FETCH x2 MB | WRITE MB |")
    print("|---|---|---|---|---|---|")
    for i in range(start, len(tr)):
        t, f, w = tr[i], fe[i], wr[i]
        print("| %d | `%s` | %d | %.1f | %.2f | %.2f |" % (i, t["name"], t["grid"], t["us"], f["val"] * 2 * 1024 / 1e6, w["val"] * 1024 / 1e6))
    print("\nbodies of that run (alive at start, slots at start, wall us): %s" % info["first_bodies_us"])


if __name__ == "__main__":
    main()
