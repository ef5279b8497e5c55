#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (tools/prof_driver_cmd.sh: rocprofv3 of the driver's bench command) into profiles/:

  profiles/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_kernel_trace.csv          every kernel dispatch of the run: kernel, grid, start, duration (ns)
  profiles/<tag>_bench_under_rocprof.json  the line bench.py printed in the trace pass
  profiles/<tag>_pmc.md                    per-dispatch HBM bytes (PMC) and SQ counters of the timed kernels, and the
                                           recomputation of every roofline fraction in the bench line from the trace
  profiles/pmc_traffic.json                the entry bench.py shows under "static_profile"

gfx950 counter corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes
(x1024 -> bytes); FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams, i.e. exactly HALF the bytes fetched,
so it is doubled; WRITE_SIZE is exact.  (Calibrated in round 1 on k_newton = 96.0 B and k_counters = 24.0 B per particle;
8-byte-per-lane kernels -- the delete passes -- are listed with the same correction and marked uncalibrated.)"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8000.0


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0][:48]


def read_counter(path):
    """{kernel: [ {counter: value} per dispatch, in order ]}"""
    per = collections.defaultdict(dict)
    order = []
    for r in csv.DictReader(open(path)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        if key not in per:
            order.append(key)
        per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = collections.defaultdict(list)
    for key in order:
        out[short(key[1])].append(per[key])
    return out


def read_counter_seq(path, kernels):
    """[(kernel, {counter: value})] of the dispatches of ``kernels``, in dispatch order"""
    per, order = {}, []
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k not in kernels:
            continue
        key = int(r["Dispatch_Id"])
        if key not in per:
            per[key] = (k, {})
            order.append(key)
        per[key][1][r["Counter_Name"]] = per[key][1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [per[k] for k in sorted(order)]


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    args = open(os.path.join(src, "bench_args.txt")).read().strip()
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    def record(name):
        """The run's full record of one pass: bench.py's detail file (round 5 on: stdout carries only the contract line)."""
        detail = os.path.join(src, name + "_detail.json")
        if os.path.exists(detail):
            return json.load(open(detail))
        return json.loads([ln for ln in open(os.path.join(src, name + ("_bench.json" if name == "trace" else ".json"))) if ln.strip()][-1])

    line = record("trace")
    json.dump(line, open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w"))
    # the records are about the build that RAN (the hash bench.py put into its record), not about the tree as it is now
    sha = line.get("csrc_sha")
    if not sha:
        raise SystemExit("the run's record carries no csrc_sha (a bench.py from before round 5?): profile again")
    sys.path.insert(0, ROOT)
    from physicl_amd import build
    if sha != build.csrc_sha():
        print("NOTE: this profile was taken on device sources %s, the tree has %s: bench.py will not quote its records" % (sha, build.csrc_sha()),
              file=sys.stderr)
    N, steps, warmup, R = line["config"]["photons_per_gpu"], line["steps"], line["warmup"], line["repeats"]
    S = line["config"].get("steps_per_launch_max", line["config"].get("steps_per_launch"))
    # ---- per-dispatch trace
    rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    disp = collections.defaultdict(list)
    with open(os.path.join(dst, tag + "_kernel_trace.csv"), "w") as f:
        f.write("dispatch,kernel,grid_x,start_ns,duration_ns\n")
        for r in rows:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            k = short(r["Kernel_Name"])
            disp[k].append(d)
            f.write("%s,%s,%s,%d,%d\n" % (r["Dispatch_Id"], k, r["Grid_Size_X"], int(r["Start_Timestamp"]) - t0, d))
    fetch = read_counter(os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv"))
    write = read_counter(os.path.join(src, "pmc_write", "pmc_counter_collection.csv"))
    sq = read_counter(os.path.join(src, "pmc_sq", "pmc_counter_collection.csv"))
    L = ["# %s: rocprofv3 of `python3 bench.py %s`" % (tag, args), "",
         "Four runs of the same command (`--kernel-trace --stats`; `--kernel-trace --pmc FETCH_SIZE`; `... WRITE_SIZE`; "
         "`... SQ_*` -- the PMC passes with `--no-cpu-baseline`, which launches no kernel).  Schedule of the run: %d warm-up "
         "steps (one K = %d launch of the K-step kernel), %d timed blocks of %d steps (one K = %d launch each), then the "
         "`single_step` leg (3 + %d x %d launches of the one-step kernel), the `delete`, `iso_1e7` and `mixed` legs and the `api` leg "
         "(three 500-pass simulations and a delete-until-empty run through the plugin API)."
         % (warmup, min(S, warmup), R, steps, min(S, steps), R, steps), "",
         "## K-step pass (`pcl_rtc_multi2*_e1`: 256 photons per wave, `pcl_rtc_multi3*_e1`: 192, `pcl_rtc_multi*_e1`: 128; `s` = with the saturation probe), every dispatch in order (bench.py's own "
         "launches first; the later ones belong to the `api` leg); the form is in brackets behind K", "",
         "Bytes = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024.  valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); "
         "lane_util = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64); VALU instr / particle-step = SQ_INSTS_VALU x 64 / (N x K).", "",
         "| # | K | duration ms (trace pass) | HBM bytes / photon | 128 B x N / t (GB/s) | frac of 8 TB/s | valu_busy | lane_util | VALU instr / particle-step |",
         "|---|---|---|---|---|---|---|---|---|"]
    # the K-step pass has two forms (128 / 256 photons per wave: pcl_rtc_multi_e1 / pcl_rtc_multi2_e1), picked per launch
    kms = ("pcl_rtc_multi_e1", "pcl_rtc_multi2_e1", "pcl_rtc_multis_e1", "pcl_rtc_multi2s_e1", "pcl_rtc_multi3_e1", "pcl_rtc_multi3s_e1")
    FORM = {"pcl_rtc_multi2_e1": "256", "pcl_rtc_multis_e1": "128 probe", "pcl_rtc_multi2s_e1": "256 probe", "pcl_rtc_multi3_e1": "192",
            "pcl_rtc_multi3s_e1": "192 probe", "pcl_rtc_multi_e1": "128"}
    KERNEL_OF = {v: k for k, v in FORM.items()}
    seq = sorted(((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                  for r in rows if short(r["Kernel_Name"]) in kms))
    f_seq = read_counter_seq(os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv"), kms)
    w_seq = read_counter_seq(os.path.join(src, "pmc_write", "pmc_counter_collection.csv"), kms)
    s_seq = read_counter_seq(os.path.join(src, "pmc_sq", "pmc_counter_collection.csv"), kms)
    ks = [min(S, warmup)] + [min(S, steps)] * R
    # the kernel's own work tallies of the SQ pass's process (bench line key k_step_launch_work), launch by launch
    sq_line = record("pmc_sq")
    work = sq_line.get("k_step_launch_work", [])
    isa = json.load(open(os.path.join(dst, "isa_counts.json"))) if os.path.exists(os.path.join(dst, "isa_counts.json")) else {}
    per_block = -(-steps // S)                       # launches per timed block
    n_main = -(-warmup // S) + R * per_block
    exprs = [line["config"]["variable_n_fn"]] * n_main + (["2.5E+25 * exp(r2[gid] / 8600.0)"] * n_main if "tame" in sq_line else [])
    calib = collections.defaultdict(list)
    priced_rows = []
    W = ["", "### The same dispatches against the kernel's own work tally (SQ pass): wave-instructions = SQ_INSTS_VALU", "",
         "model = decision instructions x wave-steps + dense-pass instructions x dense passes (profiles/isa_counts.json; the dense "
         "pass is a straight-line loop body counted in the assembly, the decision count is what this table calibrates: "
         "(SQ_INSTS_VALU - dense x passes) / wave-steps).", "",
         "| # | expression | form | K | hit fraction | wave-steps | dense passes | passes / wave-step | SQ_INSTS_VALU | (INSTS - dense x passes) / wave-steps | static upper estimate |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
    for i, wk in enumerate(work):
        if i >= len(s_seq) or i >= len(exprs):
            break
        kname, c = s_seq[i]
        e = exprs[i]
        cnt = isa.get(e, {}).get("kernels", {}).get(kname)
        if not cnt or not c.get("SQ_INSTS_VALU"):
            continue
        steps_i, hits_i, passes_i, ws_i, ppw_i = wk[:5]
        sat_i = wk[5] if len(wk) > 5 else -1
        a_dyn = (c["SQ_INSTS_VALU"] - cnt["dense_pass_valu"] * passes_i) / ws_i
        regime = "" if sat_i < 0 else ("_shortcut" if sat_i > 0.99 * ws_i else ("" if sat_i < 0.01 * ws_i else "_mixed"))
        calib[(e, kname, regime)].append((steps_i, a_dyn))
        W.append("| %d | `%s` | %s | %d | %.3f | %d | %d | %.3f | %.0f | %.1f | %.1f |" % (i, e[:24], "%d%s" % (ppw_i, "" if sat_i < 0 else " probe %.0f %%" % (100.0 * sat_i / ws_i)),
                                                                                          steps_i, hits_i / float(N * steps_i), ws_i, passes_i,
                                                                                          passes_i / float(ws_i), c["SQ_INSTS_VALU"], a_dyn,
                                                                                          cnt.get("decision_valu_per_wave_step_static", 0)))
        if "dense_pass_cycles" in cnt and c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
            # the same dispatch priced: every instruction at its class's cycles (tools/isa_count.py) against the counters' own cycles
            dec_i = c["SQ_INSTS_VALU"] - cnt["dense_pass_valu"] * passes_i
            priced = dec_i * cnt["decision_cycles_per_valu"] + cnt["dense_pass_cycles"] * passes_i
            priced4 = dec_i * cnt["decision_cycles_per_valu_at_4_waves"] + cnt["dense_pass_cycles_at_4_waves"] * passes_i
            busy_c, avail_c = c["SQ_ACTIVE_INST_VALU"] * 4.0, c["GRBM_GUI_ACTIVE"] / 8.0 * 1024
            ghz_sq = (wk[6] if len(wk) > 6 else 0.0)
            priced_rows.append("| %d | %s | %.4g | %.4g | %.4g | %.4g | %.3f | %.3f | %.3f | %.3f | %s |" % (
                i, kname, priced, priced4, busy_c, avail_c, priced / avail_c, priced4 / avail_c, busy_c / avail_c, priced4 / busy_c,
                ("%.3f" % ghz_sq) if ghz_sq else "-"))
    # The decision part's count per wave-step carries the launch's fixed work (a grid-stride trip's loads, stores and
    # restores) spread over its K steps: A(K) = a + c / K.  Every profiled run adds its (K, A) points to the table; with
    # two different K the fit gives a (per wave-step) and c (per wave and trip), which is what bench.py multiplies out.
    for (e, kname, regime), vals in sorted(calib.items()):
        if regime == "_mixed":
            continue
        ent = isa[e]["kernels"][kname]
        pts = ent.setdefault("decision_calibration_points" + regime, {})
        for st in sorted({st for st, _ in vals if st >= 16}):
            same = [a for st2, a in vals if st2 == st]
            pts[str(st)] = {"mean": round(sum(same) / len(same), 2), "min": round(min(same), 2), "max": round(max(same), 2), "launches": len(same),
                            "source": "profiles/%s_pmc.md" % tag}
        if not pts:
            continue
        kpts = sorted(int(k) for k in pts)
        if len(kpts) >= 2:
            x = [1.0 / k for k in kpts]
            y = [pts[str(k)]["mean"] for k in kpts]
            mx, my = sum(x) / len(x), sum(y) / len(y)
            c_fit = sum((xi - mx) * (yi - my) for xi, yi in zip(x, y)) / sum((xi - mx) ** 2 for xi in x)
            a_fit = my - c_fit * mx
        else:
            a_fit, c_fit = pts[str(kpts[0])]["mean"], 0.0
        ent["decision_valu_per_wave_step" + regime] = round(a_fit, 1)
        ent["decision_valu_per_wave_trip" + regime] = round(c_fit, 1)
        W.append("")
        W.append("`%s`%s, `%s`: decision instructions per wave-step at the launch lengths profiled so far %s -> **%.1f per wave-step + %.1f per "
                 "wave and trip**" % (kname, " on exp's saturation shortcut" if regime else "", e,
                                      ", ".join("K = %d: %.1f" % (k, pts[str(k)]["mean"]) for k in kpts), a_fit, c_fit))
    if priced_rows:
        W += ["", "### The same dispatches priced in SIMD-cycles", "",
              "priced = (SQ_INSTS_VALU - dense x passes) x the mean price of the decision part's instruction mix + dense-pass cycles x passes, an "
              "instruction costing 2 / 4 / 8 / 16 cycles by class (tools/isa_count.py; tools/valu_issue_probe.hip: profiles/r05_valu_issue_probe.txt); "
              "\"at 4 waves\" prices every opcode at what the probe measured with four waves on the SIMD (what these kernels keep).  busy = "
              "SQ_ACTIVE_INST_VALU x 4 (the counter ticks every fourth cycle), available = GRBM_GUI_ACTIVE / 8 x 1024 SIMDs -- both of the SQ pass "
              "itself, so no clock and no duration enters the ratios.  What separates priced-at-4-waves from busy is the time an instruction "
              "stays in flight beyond its issue slots while no other wave of the SIMD has one ready (dependent chains: the probe's own Philox "
              "round mix runs 1.35 x its priced cycles at four waves).  GHz = the clock the kernel measured in that pass (s_memtime / s_memrealtime).", "",
              "| # | kernel | priced cycles | priced at 4 waves | busy cycles | available cycles | priced / available | at 4 waves / available | busy / available | at 4 waves / busy | GHz |",
              "|---|---|---|---|---|---|---|---|---|---|---|"] + priced_rows
    if calib:
        json.dump(isa, open(os.path.join(dst, "isa_counts.json"), "w"), indent=1, sort_keys=True)
    multi_rows = []
    forms = []
    for i, (_, kname, d) in enumerate(seq[:len(ks) + 2]):
        K = ks[i] if i < len(ks) else None
        forms.append(FORM.get(kname, "128"))
        fb = f_seq[i][1].get("FETCH_SIZE", 0) * 2 * 1024 if i < len(f_seq) else 0
        wb = w_seq[i][1].get("WRITE_SIZE", 0) * 1024 if i < len(w_seq) else 0
        c = s_seq[i][1] if i < len(s_seq) else {}
        busy = c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) if c.get("GRBM_GUI_ACTIVE") else 0
        util = c.get("SQ_THREAD_CYCLES_VALU", 0) / (c["SQ_ACTIVE_INST_VALU"] * 64) if c.get("SQ_ACTIVE_INST_VALU") else 0
        per_ps = c.get("SQ_INSTS_VALU", 0) * 64 / (N * K) if K else 0
        gb = 128.0 * N / (d * 1e-9) / 1e9
        multi_rows.append((K, d, fb + wb, busy, util, per_ps))
        L.append("| %d | %s | %.3f | %.1f | %.0f | %.4f | %.3f | %.3f | %s |" % (i, ("%d (%s)" % (K, forms[-1])) if K else "api (%s)" % forms[-1], d * 1e-6,
                                                                              (fb + wb) / N, gb, gb / HBM_PEAK, busy, util, "%.1f" % per_ps if K else "-"))
    L += W
    timed = [r for r in multi_rows[1:1 + R]]
    hbm_line = line["roofline"].get("hbm", line["roofline"])
    if timed:
        avg = sum(r[1] for r in timed) / len(timed)
        L += ["", "Timed launches (rows 1..%d): average %.3f ms -> 128 B x N / t = %.0f GB/s = **%.4f** of peak (bench line under rocprof: "
              "`roofline.avg_launch_ms` %.3f, `roofline.hbm.frac` %.4f); in the per-step form 104 B x %d x N / t = %.0f GB/s."
              % (R, avg * 1e-6, 128.0 * N / (avg * 1e-9) / 1e9, 128.0 * N / (avg * 1e-9) / 1e9 / HBM_PEAK, line["roofline"]["avg_launch_ms"],
                 hbm_line["frac"], min(S, steps), 104.0 * min(S, steps) * N / (avg * 1e-9) / 1e9)]
    # ---- one-step kernel
    kf = "pcl_rtc_fast_e1"
    d_f = disp.get(kf, [])
    n_single = 3 + R * steps
    single = d_f[:n_single][3:]
    if single:
        avg = sum(single) / len(single)
        fb = [x.get("FETCH_SIZE", 0) * 2 * 1024 for x in fetch.get(kf, [])[:n_single][3:]]
        wb = [x.get("WRITE_SIZE", 0) * 1024 for x in write.get(kf, [])[:n_single][3:]]
        c = sq.get(kf, [])[:n_single][3:]
        tot = (sum(fb) / len(fb) if fb else 0) + (sum(wb) / len(wb) if wb else 0)
        busy = sum(x.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (x["GRBM_GUI_ACTIVE"] / 8 * 1024) for x in c if x.get("GRBM_GUI_ACTIVE")) / max(1, len(c))
        L += ["", "## One-step kernel `pcl_rtc_fast_e1` (`single_step` leg: %d timed launches)" % len(single), "",
              "average %.4f ms (min %.4f, max %.4f) -> 104 B x N / t = %.0f GB/s = **%.4f** of peak (bench line under rocprof: "
              "`single_step.roofline.frac` %.4f); PMC: %.1f B / photon per launch (algorithmic 104); valu_busy %.3f."
              % (avg * 1e-6, min(single) * 1e-6, max(single) * 1e-6, 104.0 * N / (avg * 1e-9) / 1e9, 104.0 * N / (avg * 1e-9) / 1e9 / HBM_PEAK,
                 line["single_step"]["roofline"]["frac"], tot / N, busy)]
    # ---- delete legs: the alive-mask kernel dispatch by dispatch (the first run of the 1e8 leg), then totals per kernel
    ka = [k for k in disp if k.startswith("k_delete_alive<double, true")]
    big, idx_of = [], {}
    if ka:
        ka = ka[0]
        big = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == ka and int(r["Grid_Size_X"]) >= 4000000]
        L += ["", "## `k_delete_alive<double, true>` (one delete loop body on the alive mask; counters + plane crossings): the 1e8-photon "
              "dispatches of the `delete` leg", "",
              "A dispatch sweeps ``slots`` = the store's extent; algorithmic bytes per SLOT: alive bit read + written (0.25), v (24), r along the "
              "plane's axis (8; all of r, 24, until round 4), the id (8) once ids are explicit.  HBM bytes = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 (16-byte loads: the calibrated case).", "",
              "| dispatch | grid (threads) | duration us | HBM bytes (MB) | GB/s (PMC bytes / duration) | frac of 8 TB/s |", "|---|---|---|---|---|---|"]
        idx_of = {}
        for j, r in enumerate(rows):
            if short(r["Kernel_Name"]) == ka:
                idx_of[j] = len(idx_of)
        for j in big[:8]:
            i = idx_of[j]
            d = int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])
            fb = fetch[ka][i].get("FETCH_SIZE", 0) * 2 * 1024 if i < len(fetch.get(ka, [])) else 0
            wb = write[ka][i].get("WRITE_SIZE", 0) * 1024 if i < len(write.get(ka, [])) else 0
            g = (fb + wb) / (d * 1e-9) / 1e9
            L.append("| %d | %s | %.1f | %.1f | %.0f | %.3f |" % (i, rows[j]["Grid_Size_X"], d * 1e-3, (fb + wb) / 1e6, g, g / HBM_PEAK))
    # ---- the compactions of >= 1e7 slots, dispatch by dispatch
    kc = [k for k in disp if k.startswith("k_compact_lds<double") or k.startswith("k_compact_count<double")]
    comp_rows = []
    for k in kc:
        js = [j for j, r in enumerate(rows) if short(r["Kernel_Name"]) == k]
        for i, j in enumerate(js):
            slots = int(rows[j]["Grid_Size_X"]) * 8
            d = int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])
            if slots < 10_000_000 or d < 20_000:
                continue
            fb = fetch[k][i].get("FETCH_SIZE", 0) * 2 * 1024 if i < len(fetch.get(k, [])) else 0
            wb = write[k][i].get("WRITE_SIZE", 0) * 1024 if i < len(write.get(k, [])) else 0
            comp_rows.append((k, slots, d, fb, wb))
    if comp_rows:
        L += ["", "## `k_compact_lds` / `k_compact_count` (stable compaction, pass 3): the dispatches of >= 1e7 slots", "",
              "slots = grid x 8 (one 256-thread workgroup per 2048-slot tile).  HBM bytes = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024.  `<..., 7>` = the "
              "`delete` leg (r, v, E move; dv known to be zero): the staged kernel reads every 128-byte line of its seven source rows (56 B per "
              "SLOT: at a third of the photons surviving at random every line holds a survivor) and writes 64 B per survivor; the direct one "
              "(chosen by the scan below 15 % survivors -- where the compactions of the bodies worked out ahead land) reads 8 B per surviving "
              "lane.  `<..., 10>` = the `mixed` leg (the dv / vprev rows travel too).", "",
              "| kernel | slots | duration us | FETCH x2 MB | WRITE MB | HBM B / slot | HBM GB/s | frac of 8 TB/s |", "|---|---|---|---|---|---|---|---|"]
        for k, slots, d, fb, wb in comp_rows[:24]:
            L.append("| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.0f | %.3f |" % (k, slots, d * 1e-3, fb / 1e6, wb / 1e6, (fb + wb) / slots,
                                                                               (fb + wb) / (d * 1e-9) / 1e9, (fb + wb) / (d * 1e-9) / 1e9 / HBM_PEAK))
    # (runs with at most one plane take k_delete_ahead_live: the bench's delete legs do)
    kah = [k for k in disp if k.startswith("k_delete_ahead_live<double")] or [k for k in disp if k.startswith("k_delete_ahead<double")]
    ahead_bytes = []
    if kah:
        kah = kah[0]
        js = [j for j, r in enumerate(rows) if short(r["Kernel_Name"]) == kah]
        L += ["", "## `%s` (K delete loop bodies worked out in one launch): the dispatches with the capped grid (extents of >= 3.4e7 slots)" % kah, "",
              "| dispatch | grid (threads) | duration us | FETCH x2 MB | WRITE MB | HBM GB/s |", "|---|---|---|---|---|---|"]
        shown = 0
        for i, j in enumerate(js):
            g = int(rows[j]["Grid_Size_X"])
            if g < 4_000_000 or shown >= 8:
                continue
            d = int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])
            fb = fetch[kah][i].get("FETCH_SIZE", 0) * 2 * 1024 if i < len(fetch.get(kah, [])) else 0
            wb = write[kah][i].get("WRITE_SIZE", 0) * 1024 if i < len(write.get(kah, [])) else 0
            L.append("| %d | %d | %.1f | %.1f | %.1f | %.0f |" % (i, g, d * 1e-3, fb / 1e6, wb / 1e6, (fb + wb) / (d * 1e-9) / 1e9))
            shown += 1
            if fb > 3.0e9:                   # (the 1e8-slot extent: reads of ~32 B per slot)
                ahead_bytes.append((fb + wb) / 1e8)
    L += ["", "## Delete legs (`delete` record): kernel totals over the whole run (warm-up repetition included; of the two pass-3 "
          "kernels enqueued per compaction the one the scan did not choose returns at once)", "",
          "| kernel | dispatches | total ms | FETCH x2 (GB) | WRITE (GB) |", "|---|---|---|---|---|"]
    for k in sorted(disp):
        if any(t in k for t in ("k_delete_alive", "k_delete_ahead", "k_ahead_commit", "k_apply_pending", "k_newton_mask", "k_flag_mask2", "k_compact_count", "k_compact_lds", "k_tile_scan",
                                "k_mixed", "k_any_nonzero", "k_delete_onepass")):
            fb = sum(x.get("FETCH_SIZE", 0) for x in fetch.get(k, [])) * 2 * 1024 / 1e9
            wb = sum(x.get("WRITE_SIZE", 0) for x in write.get(k, [])) * 1024 / 1e9
            L.append("| `%s` | %d | %.3f | %.2f | %.2f |" % (k, len(disp[k]), sum(disp[k]) * 1e-6, fb, wb))
    dl = line.get("delete", {}).get("sizes", {})
    for size, rec in dl.items():
        p = rec["per_step"]
        L.append("")
        pa = p.get("roofline_alive", p["roofline"])        # (k_delete_alive's record; "roofline" is k_delete_ahead_live's when it ran)
        L.append("`delete` %s photons under the profiler, per_step: %.4g particle-steps/s in %d loop bodies (%d of them compact); "
                 "`k_delete_alive` %.0f GB/s on the slots it sweeps = %.3f of peak, the compactions %.0f GB/s = %.3f; %.1f B per alive "
                 "particle-step; multi (K = %d): %.4g particle-steps/s."
                 % (size, p["value"], p["loop_bodies"], p["roofline_compaction"]["compactions"], pa["achieved"], pa["frac"],
                    p["roofline_compaction"]["achieved"], p["roofline_compaction"]["frac"], pa["bytes_per_alive_particle_step"],
                    rec["multi"]["steps_per_launch"], rec["multi"]["value"]))
        for mode in ("per_step", "multi"):
            v = rec[mode].get("roofline") or {}
            if v.get("bound") == "valu":
                L.append("  `%s`: `k_delete_ahead_live` %.4g wave-instructions (its own tally x profiles/isa_counts.json) in %.3f ms = %.3g / s = **%.3f** of "
                         "the VALU issue peak; 33 B per slot = %.0f GB/s." % (mode, v["wave_instructions"], v["total_ms"], v["achieved"], v["frac"], v["hbm"]["achieved"]))
    L += ["", "## All kernels of the trace pass (count, total ms)", "", "| kernel | dispatches | total ms | avg ms |", "|---|---|---|---|"]
    for k in sorted(disp, key=lambda k: -sum(disp[k])):
        L.append("| `%s` | %d | %.3f | %.4f |" % (k, len(disp[k]), sum(disp[k]) * 1e-6, sum(disp[k]) / len(disp[k]) * 1e-6))
    open(os.path.join(dst, tag + "_pmc.md"), "w").write("\n".join(L) + "\n")
    # ---- static_profile entry of bench.py
    tf = os.path.join(dst, "pmc_traffic.json")
    allt = json.load(open(tf)) if os.path.exists(tf) else {}
    # records of other device sources than the ones profiled here are history (git has them): bench.py would not quote them
    allt = {k: v for k, v in allt.items() if k == "kernels" or v.get("csrc_sha") == sha}
    allt["kernels"] = {k: v for k, v in allt.get("kernels", {}).items() if v.get("csrc_sha") == sha}
    # per-kernel HBM bytes per unit of work: what bench.py's records carry as ``traffic`` (with this file and commit as source)
    commit = os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip()
    kern_t = allt.setdefault("kernels", {})
    src_md = "profiles/%s_pmc.md" % tag
    for i, (K, d, b_hbm, busy, util, per_ps) in enumerate(multi_rows[1:1 + R]):
        kern_t[KERNEL_OF.get(forms[1 + i], "pcl_rtc_multi_e1")] = {
            "bytes_per_unit": round(b_hbm / N, 2), "unit": "photon (per launch)", "valu_busy": round(busy, 4), "lane_utilisation": round(util, 4),
            "source": src_md, "commit": commit, "csrc_sha": sha}
    if single:
        kern_t["pcl_rtc_fast_e1"] = {"bytes_per_unit": round(tot / N, 2), "unit": "photon (per launch)", "source": src_md, "commit": commit, "csrc_sha": sha}
    if ka and big:
        per = []
        for j in big[:8]:
            i = idx_of[j]
            if i < len(fetch.get(ka, [])) and i < len(write.get(ka, [])):
                # (the kernel's grid is capped and strides over the tiles: a grid of >= 4e6 threads is the 1e8-slot extent)
                per.append((fetch[ka][i].get("FETCH_SIZE", 0) * 2 * 1024 + write[ka][i].get("WRITE_SIZE", 0) * 1024) / 1e8)
        if per:
            kern_t["k_delete_alive<double, true>"] = {"bytes_per_unit": round(sum(per) / len(per), 2), "unit": "slot", "source": src_md, "commit": commit, "csrc_sha": sha}
    if ahead_bytes:
        csq = [c for c in sq.get(kah, []) if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_INSTS_VALU", 0) > 1e8]
        kern_t[kah] = {"bytes_per_unit": round(sum(ahead_bytes) / len(ahead_bytes), 2), "unit": "slot (per launch, whatever K)",
                       "source": src_md, "commit": commit, "csrc_sha": sha}
        if csq:
            kern_t[kah]["valu_busy"] = round(sum(c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) for c in csq) / len(csq), 4)
            kern_t[kah]["lane_utilisation"] = round(sum(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64) for c in csq) / len(csq), 4)
        kern_t["k_delete_ahead_live<double>"] = dict(kern_t[kah])    # (the name bench.py asks for: either template form)
    for kname in sorted({r[0] for r in comp_rows if ", 7>" in r[0]}):
        first = [r for r in comp_rows if r[0] == kname and r[1] >= 90_000_000]
        if first:
            kern_t[kname] = {"bytes_per_unit": round(sum((r[3] + r[4]) / r[1] for r in first) / len(first), 2),
                             "unit": "slot (the compactions of the 1e8-slot extent in the delete leg)", "source": src_md, "commit": commit, "csrc_sha": sha}
    # k_mixed (configs[4]): VALU issue of its 16-iteration launches -- SQ_INSTS_VALU of the SQ pass over their durations in the
    # trace pass (the run is deterministic: same dispatches, same order) -- for the bench line's ``mixed.roofline``
    # (configs[4]'s constant-n loop at a hit probability of 0.3 takes k_mixed3, three rows per wave and trip; k_mixed otherwise)
    for knames, tag_t in ((("k_mixed3<double, false>", "k_mixed<double, false, 0>"), "f64"), (("k_mixed3<float, false>", "k_mixed<float, false, 0>"), "f32")):
        kname = next((k for k in knames if any(c.get("SQ_INSTS_VALU", 0) > 1e9 for c in sq.get(k, []))), knames[0])
        pairs = [(c, d) for c, d in zip(sq.get(kname, []), disp.get(kname, [])) if c.get("SQ_INSTS_VALU", 0) > 1e9 and c.get("GRBM_GUI_ACTIVE")]
        if pairs:
            insts, dur = sum(c["SQ_INSTS_VALU"] for c, _ in pairs), sum(d for _, d in pairs) * 1e-9
            kern_t["k_mixed valu " + tag_t] = {
                "wave_instructions_per_s": insts / dur, "launches": len(pairs), "wave_instructions": insts, "seconds": dur,
                "valu_busy": round(sum(c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) for c, _ in pairs) / len(pairs), 4),
                "lane_utilisation": round(sum(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64) for c, _ in pairs) / len(pairs), 4),
                # (the SQ pass's own GRBM cycles over the trace pass's durations of the same dispatches)
                "clock_GHz": round(sum(c["GRBM_GUI_ACTIVE"] for c, _ in pairs) / 8.0 / (sum(d for _, d in pairs)), 4),
                "kernel": kname, "source": src_md, "commit": commit, "csrc_sha": sha}
            mixk = isa.get("aot", {}).get(kname, {})
            ghz = kern_t["k_mixed valu " + tag_t]["clock_GHz"]
            L += ["", "`%s`, the %d launches of 16 iterations: %.4g VALU wave-instructions in %.2f ms at %.3f GHz (GRBM_GUI_ACTIVE / 8 / duration); "
                  "VALU busy %.3f (SQ_ACTIVE_INST_VALU x 4 / available SIMD-cycles), lane utilisation %.3f; priced at %.2f cycles per instruction "
                  "(the kernel's static mix, tools/isa_count.py --aot): %.3f of the available cycles (%.3f at 4 waves per SIMD)." % (
                      kname, len(pairs), insts, dur * 1e3, ghz, kern_t["k_mixed valu " + tag_t]["valu_busy"],
                      kern_t["k_mixed valu " + tag_t]["lane_utilisation"], mixk.get("cycles_per_valu", 0.0),
                      insts * mixk.get("cycles_per_valu", 0.0) / (dur * ghz * 1e9 * 1024),
                      insts * mixk.get("cycles_per_valu_at_4_waves", 0.0) / (dur * ghz * 1e9 * 1024))]
    if timed:
        t_ok = [r for r in timed if r[0]]
        ent = {"source": "profiles/%s_pmc.md" % tag, "command": "bench.py " + args, "K": min(S, steps), "commit": commit, "csrc_sha": sha,
               "hit_fraction_of_timed_blocks": line["repeat_hit_fraction"],
               "k_multi_bytes_per_launch": sum(r[2] for r in t_ok) / len(t_ok),
               "k_multi_valu": {"busy": round(sum(r[3] for r in t_ok) / len(t_ok), 4),
                                "lane_utilisation": round(sum(r[4] for r in t_ok) / len(t_ok), 4),
                                "valu_insts_per_particle_step_per_block": [round(r[5], 1) for r in t_ok]}}
        if single:
            ent["k_fast_bytes_per_launch"] = tot
        allt["%s:%s:%d:K%d:steps%d" % (line["config"]["profile"], line["config"]["mode"], N, S, steps)] = ent
        json.dump(allt, open(tf, "w"), indent=1, sort_keys=True)
    print("\n".join(L))


if __name__ == "__main__":
    main()
