#!/usr/bin/env python3
"""Static instruction counts of the K-step kernels, read off the gfx950 assembly of the very translation unit hipRTC
compiles for an expression (``#define PCL_N_EXPR ...`` + physicl_amd/csrc/pcl_device.h, same options), compiled offline:

    python tools/isa_count.py ["<variable_n_fn>"] [--json profiles/isa_counts.json] [--md]

For each kernel (pcl_rtc_multi_e1: 128 photons per wave, pcl_rtc_multi2_e1: 256, pcl_rtc_multi3_e1: 192) the loop nest is recovered from the
backward branches: the grid-stride loop, the K loop inside it, the dense pass of the hit queue inside that.  Reported per
kernel: VALU instructions (v_*) of the K loop's body outside the dense-pass loop, split into the blocks every step runs and
the blocks only some steps run (the Philox decision block: every second step; the vprev store: the last step), and of one
dense pass; v_readlane / v_writelane inside the K loop; registers and spills from the metadata.  The bench's VALU roofline
record multiplies these with the wave-steps and dense passes the kernel itself tallies (pcl_store_last_multi_work).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# ---------------------------------------------------------------------------------------------------------------------
# What a wave64 vector instruction costs its SIMD, in cycles of VALU issue, by class: measured on gfx950 by
# tools/valu_issue_probe.hip (profiles/r05_valu_issue_probe.txt: every SIMD holding 1, 2, 4 and 8 waves of ONE opcode; the
# figures below are the values the 8-wave streams approach -- 2.13-2.25, 4.05-4.17, 8.05, 16.08 -- so that cost x count
# never overstates the issue work).  They confirm MI355X_MICROARCH.md:54 ("2 cycles" for 32-bit VALU on a busy SIMD, 4 for one
# wave alone) and its constants table (:473 v_fma_f32 2 cyc, transcendentals 8) and add what the guide does not list: EVERY
# fp64 arithmetic instruction, every integer multiply (v_mad_u64_u32, v_mul_lo/hi_u32: full "4-cycle" rate, not quarter
# rate), every three-operand integer VOP3 (v_add3, v_lshl_or, v_alignbit, v_perm, v_mad_u32_u24, v_mbcnt), every compare,
# carry, conversion, select and cross-lane read/write costs 4; fp64 rcp / rsq / sqrt cost 16.
# ---------------------------------------------------------------------------------------------------------------------
COST_CLASSES = {
    "fp64 transcendental (16)": (16, ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")),
    "fp32 transcendental (8)": (8, ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")),
    "32-bit simple (2)": (2, ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
                              "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32",
                              "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_max_f32", "v_min_f32", "v_max_u32", "v_min_u32", "v_max_i32",
                              "v_min_i32", "v_bitop3_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32")),
}
DEFAULT_CLASS = "fp64 / 64-bit / multiply / compare / select / convert / cross-lane (4)"


def valu_class(op):
    """(class name, cycles) of one vector opcode as it stands in the assembly (suffixes _e32 / _e64 / _sdwa / _dpp dropped;
    an SDWA or DPP form of a 2-cycle opcode is priced at 4)."""
    base = re.sub(r"_(e32|e64)$", "", op)
    if base.endswith("_sdwa") or base.endswith("_dpp"):
        return DEFAULT_CLASS, 4
    for name, (cyc, ops) in COST_CLASSES.items():
        if base in ops:
            return name, cyc
    return DEFAULT_CLASS, 4


_PROBE = None


def probed_cost(op, waves=4):
    """Cycles per wave-instruction with ``waves`` waves on the SIMD as tools/valu_issue_probe.hip measured them
    (profiles/r05_valu_issue_probe.json); opcodes the probe did not run take their class's nominal price x the mean
    measured / nominal ratio of the probed opcodes of that class."""
    global _PROBE
    if _PROBE is None:
        _PROBE = {"by_op": {}, "ratio": {}}
        path = os.path.join(ROOT, "profiles", "r05_valu_issue_probe.json")
        if os.path.exists(path):
            acc = {}
            for r in json.load(open(path))["rows"]:
                if r["instruction"].startswith("v_") and " " not in r["instruction"] and r["instruction"] != "v_cndmask_b32":
                    _PROBE["by_op"][(r["instruction"], r["waves_per_simd"])] = r["cycles_per_wave_instruction"]
                    name, cyc = valu_class(r["instruction"])
                    acc.setdefault((name, r["waves_per_simd"]), []).append(r["cycles_per_wave_instruction"] / cyc)
            _PROBE["ratio"] = {k: sum(v) / len(v) for k, v in acc.items()}
    base = re.sub(r"_(e32|e64)$", "", op)
    hit = _PROBE["by_op"].get((base, waves))
    if hit is not None:
        return hit
    name, cyc = valu_class(op)
    return cyc * _PROBE["ratio"].get((name, waves), 1.0)


def class_histogram(ops):
    """{class: count}, total instructions, total cycles (nominal), total cycles at 4 waves per SIMD of a list of vector opcodes."""
    hist, cycles, cycles4 = {}, 0, 0.0
    for op in ops:
        name, cyc = valu_class(op)
        hist[name] = hist.get(name, 0) + 1
        cycles += cyc
        cycles4 += probed_cost(op, 4)
    return hist, len(ops), cycles, cycles4


def build_sha():
    from physicl_amd import build
    return build.csrc_sha()


def rtc_source(expr, expr_f32=None, dt=0, use_e=1, extra=()):
    from physicl_amd import build
    build._generate_rtc_source()
    inc = open(os.path.join(ROOT, "physicl_amd", "csrc", "pcl_rtc_source.inc")).read()
    text = inc[inc.index('R"PCLRTC(') + 9:inc.rindex(')PCLRTC"')]
    if expr_f32 is None:
        expr_f32 = re.sub(r"(?<![\w.])(\d+\.\d*|\d*\.\d+|\d+)([eE][-+]?\d+)?(?![\w.])",
                          lambda m: m.group(0) + ("f" if ("." in m.group(0) or m.group(2)) else ".0f"), expr)
    head = "#define PCL_RTC 1\n#define PCL_RTC_DT %d\n#define PCL_RTC_E %d\n" % (dt, use_e)
    head += "".join("#define %s 1\n" % x for x in extra)
    return head + "#define PCL_N_EXPR (%s)\n#define PCL_N_EXPR_F (%s)\n" % (expr, expr_f32) + text


# what the library adds to every compile, ahead-of-time and hipRTC (physicl_amd/build.py: EXTRA_OPTS); PCL_ISA_OPTS overrides
def _extra_opts():
    if "PCL_ISA_OPTS" in os.environ:
        return os.environ["PCL_ISA_OPTS"].split()
    from physicl_amd import build
    return list(build.EXTRA_OPTS)


EXTRA_OPTS = _extra_opts()


def compile_asm(src, workdir):
    hip, asm = os.path.join(workdir, "tu.hip"), os.path.join(workdir, "tu.s")
    open(hip, "w").write(src)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-include", "hip/hip_runtime.h", "--offload-arch=gfx950", "-O3",
                           "-ffp-contract=off", "-std=c++17", "--cuda-device-only", "-Wno-unused-command-line-argument"] + EXTRA_OPTS + ["-S", "-o", asm, hip],
                          stderr=subprocess.DEVNULL)
    return open(asm).read()


def function_body(asm, name):
    lines = asm.splitlines()
    start = lines.index(next(ln for ln in lines if ln.startswith(name + ":")))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start + 1:end]


def analyse(asm, name):
    body = function_body(asm, name)
    ins, labels = [], {}
    for ln in body:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                labels[m.group(1)] = len(ins)
            continue
        ins.append(t.split(";")[0].strip())
    loops = []                     # (first instruction, branch instruction) of every backward branch
    for i, t in enumerate(ins):
        m = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    head = {}                      # several backward branches to one header are one loop: keep the last
    for a, b in loops:
        head[a] = max(b, head.get(a, b))
    loops = sorted(head.items(), key=lambda lp: (lp[0], -lp[1]))

    def depth_of(lp):
        return sum(1 for o in loops if o != lp and o[0] <= lp[0] and lp[1] <= o[1])

    def count(lo, hi, pred, skip=()):
        return sum(1 for i in range(lo, hi + 1) if pred(ins[i]) and not any(a <= i <= b for a, b in skip))

    valu = lambda t: t.startswith("v_")                                  # noqa: E731
    grid = max((lp for lp in loops if depth_of(lp) == 0), key=lambda lp: lp[1] - lp[0])
    inner1 = [lp for lp in loops if lp != grid and grid[0] <= lp[0] and lp[1] <= grid[1] and depth_of(lp) == 1]
    kloop = max(inner1, key=lambda lp: lp[1] - lp[0])
    inner2 = [lp for lp in loops if lp != kloop and kloop[0] <= lp[0] and lp[1] <= kloop[1]]
    # the dense pass: the inner loop that holds the direction block's Philox multiplies and the sincos polynomials
    innermost = [lp for lp in inner2 if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in inner2)]
    dense = max(innermost, key=lambda lp: count(lp[0], lp[1], valu))     # (the 256-photon form wraps it in its loop over queue rounds)
    others = [lp for lp in inner2 if lp != dense and not (dense[0] <= lp[0] and lp[1] <= dense[1])]
    # basic blocks of the K loop outside its inner loops, with what they hold
    starts = sorted({kloop[0]} | {v for v in labels.values() if kloop[0] <= v <= kloop[1]} |
                    {i + 1 for i in range(kloop[0], kloop[1]) if re.match(r"^s_c?branch", ins[i])})
    blocks = []
    for a, b in zip(starts, starts[1:] + [kloop[1] + 1]):
        if any(x <= a and b - 1 <= y for x, y in inner2):
            continue
        n = count(a, b - 1, valu)
        if n == 0:
            continue
        mads = count(a, b - 1, lambda t: t.startswith("v_mad_u64_u32"))
        stores = count(a, b - 1, lambda t: t.startswith("global_store") or t.startswith("buffer_store"))
        blocks.append({"first": a - kloop[0], "valu": n, "philox_mads": mads, "global_stores": stores})
    # the Philox decision block of a photon (ten rounds: >= 10 v_mad_u64_u32) and its tail block; run on every second step
    philox = 0
    for i, b in enumerate(blocks):
        if b["philox_mads"] >= 10:
            philox += b["valu"]
            if i + 1 < len(blocks) and 1 <= blocks[i + 1]["philox_mads"] < 10:
                philox += blocks[i + 1]["valu"]
    last_step = sum(b["valu"] for b in blocks if b["global_stores"] >= 3 and b["philox_mads"] < 10)
    always = sum(b["valu"] for b in blocks) - philox - last_step
    meta = {}
    m = re.search(r"\.name:\s+%s\n(.*?)\n\s+- \.", asm[asm.index(".amdgpu_metadata") if ".amdgpu_metadata" in asm else 0:] + "\n  - .", re.S)
    md = asm[asm.rindex(".name:           %s" % name):]
    for key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
        mm = re.search(r"\.%s:\s+(\d+)" % key, md)
        if mm:
            meta[key] = int(mm.group(1))
    lanes = lambda t: t.startswith("v_readlane") or t.startswith("v_writelane")   # noqa: E731
    # cycle-weighted: the dense pass is a straight-line loop body (exact); the decision part by the mean price of its static
    # instruction mix (its branches skip blocks of much the same make-up) x the dynamic count the PMC run calibrates
    dense_ops = [ins[i].split()[0] for i in range(dense[0], dense[1] + 1) if valu(ins[i])]
    dec_ops = [ins[i].split()[0] for i in range(kloop[0], kloop[1] + 1) if valu(ins[i]) and not any(a <= i <= b for a, b in inner2)]
    dense_hist, dense_n, dense_cyc, dense_cyc4 = class_histogram(dense_ops)
    dec_hist, dec_n, dec_cyc, dec_cyc4 = class_histogram(dec_ops)
    return {
        "dense_pass_classes": dense_hist, "dense_pass_cycles": dense_cyc, "dense_pass_cycles_at_4_waves": round(dense_cyc4, 1),
        "decision_classes_static": dec_hist, "decision_cycles_per_valu": round(dec_cyc / float(dec_n), 4) if dec_n else 0.0,
        "decision_cycles_per_valu_at_4_waves": round(dec_cyc4 / float(dec_n), 4) if dec_n else 0.0,
        "dense_pass_cycles_per_valu": round(dense_cyc / float(dense_n), 4) if dense_n else 0.0,
        "kernel": name,
        "k_loop": {"instructions": kloop[1] - kloop[0] + 1, "valu_total_static": count(kloop[0], kloop[1], valu)},
        "decision_blocks": blocks,
        "decision_valu_every_step": always,
        "decision_valu_philox_block_every_second_step": philox,
        "decision_valu_last_step_only": last_step,
        # a static estimate of what a wave-step costs: every block outside the dense pass once, the Philox decision block
        # every second step (it serves two consecutive steps).  Blocks that exclude each other (the odd step's R::uniform
        # against the even step's block tail, the per-counter "if (w_x)" adds) make this an upper estimate; the bench uses
        # "decision_valu_per_wave_step", which tools/calibrate_isa_counts.py sets from SQ_INSTS_VALU of a rocprofv3 run
        # together with the dense passes the kernel tallied in that run (and which must lie below this figure).
        "decision_valu_per_wave_step_static": always + 0.5 * philox,
        "decision_valu_per_wave_step": always + 0.5 * philox,
        "dense_pass_valu": count(dense[0], dense[1], valu),
        "other_inner_loops_valu": [count(lp[0], lp[1], valu) for lp in others],
        "readlane_in_k_loop": count(kloop[0], kloop[1], lambda t: t.startswith("v_readlane")),
        "writelane_in_k_loop": count(kloop[0], kloop[1], lambda t: t.startswith("v_writelane")),
        "readlane_writelane_in_dense_pass": count(dense[0], dense[1], lanes),
        "registers": meta,
    }


def aot_kernels(names, workdir):
    """Whole-kernel static instruction mix of ahead-of-time kernels of libphysicl_hip (demangled names as rocprofv3 prints
    them, without the argument list): {name: {classes, valu, cycles_per_valu, registers}}.  For the kernels whose
    instruction counts per unit of work are calibrated against SQ_INSTS_VALU (k_delete_ahead_live) or taken from the
    counters of a committed run (k_mixed): the mean price of an instruction of theirs."""
    from physicl_amd import build
    build._generate_rtc_source()
    asm_path = os.path.join(workdir, "aot.s")
    if not (os.path.exists(asm_path) and all(os.path.getmtime(asm_path) > os.path.getmtime(f) for f in build.SOURCES)):
        subprocess.check_call([build.HIPCC] + [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] +
                              ["--cuda-device-only", "-S", "-o", asm_path, build.SOURCES[0]], stderr=subprocess.DEVNULL)
    asm = open(asm_path).read()
    labels = re.findall(r"^(_Z[\w$.]+):", asm, re.M)
    dem = subprocess.check_output(["c++filt"], input="\n".join(labels).encode()).decode().splitlines()
    out = {}
    for want in names:
        hits = [m for m, d in zip(labels, dem) if d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] == want]
        if not hits:
            raise SystemExit("no kernel %r in the library's assembly" % want)
        body = function_body(asm, hits[0])
        ops = [t.strip().split(";")[0].split()[0] for t in body if t.strip().startswith("v_")]
        hist, n, cyc, cyc4 = class_histogram(ops)
        md = asm[asm.rindex(".name:           %s" % hits[0]):]
        meta = {}
        for key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
            mm = re.search(r"\.%s:\s+(\d+)" % key, md)
            if mm:
                meta[key] = int(mm.group(1))
        out[want] = {"classes_static": hist, "valu_static": n, "cycles_per_valu": round(cyc / float(n), 4),
                     "cycles_per_valu_at_4_waves": round(cyc4 / float(n), 4), "registers": meta}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("expr", nargs="?", default="0.000000001 * exp(r0[gid] - 5)")
    ap.add_argument("--json", default=None, help="merge the result into this file under the expression's text")
    ap.add_argument("--kernels", default="pcl_rtc_multi_e1,pcl_rtc_multis_e1,pcl_rtc_multi2_e1,pcl_rtc_multi2s_e1,pcl_rtc_multi3_e1,pcl_rtc_multi3s_e1")
    ap.add_argument("--extra", default="", help="comma separated PCL_RTC_EXTRA names")
    ap.add_argument("--keep", default=None, help="directory to keep tu.hip / tu.s in")
    ap.add_argument("--aot", default=None, help="comma separated ahead-of-time kernels (demangled, e.g. 'k_delete_ahead_live<double, false>'): "
                                                "their static instruction mix, merged under \"aot\" of --json")
    a = ap.parse_args()
    work = a.keep or tempfile.mkdtemp(prefix="pcl_isa_")
    os.makedirs(work, exist_ok=True)
    if a.aot:
        rec = aot_kernels([x.strip() for x in a.aot.split(";") if x.strip()], work)
        print(json.dumps(rec, indent=1))
        if a.json:
            table = json.load(open(a.json)) if os.path.exists(a.json) else {}
            table.setdefault("aot", {}).update(rec)
            table["csrc_sha"] = build_sha()
            json.dump(table, open(a.json, "w"), indent=1, sort_keys=True)
        return
    asm = compile_asm(rtc_source(a.expr, extra=[x for x in a.extra.split(",") if x]), work)
    out = {k: analyse(asm, k) for k in a.kernels.split(",")}
    hipcc = subprocess.check_output(["/opt/rocm/bin/hipcc", "--version"]).decode().splitlines()[0]
    rec = {"expression": a.expr, "compiler": hipcc, "options": "--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 " + " ".join(EXTRA_OPTS), "kernels": out}
    print(json.dumps(rec, indent=1))
    if a.json:
        table = json.load(open(a.json)) if os.path.exists(a.json) else {}
        # (calibrations of OTHER device sources are history: a changed kernel starts from its static counts again)
        old = table.get(a.expr, {}).get("kernels", {}) if table.get("csrc_sha") == build_sha() else {}
        for k, v in rec["kernels"].items():        # calibrated counts (tools/summarize_driver_prof.py) survive a re-count of the statics
            calibrated = any("calibration" in key for key in old.get(k, {}))
            for key, val in old.get(k, {}).items():
                if "calibration" in key or (calibrated and not key.endswith("_static") and
                                            (key.startswith("decision_valu_per_wave_step") or key.startswith("decision_valu_per_wave_trip"))):
                    v[key] = val
        table[a.expr] = rec
        table["csrc_sha"] = build_sha()
        json.dump(table, open(a.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
