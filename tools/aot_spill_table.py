"""Register / spill table of every ahead-of-time kernel of libphysicl_hip, and what the spills cost inside the loops:

    python tools/aot_spill_table.py [--asm DIR/aot.s] [--json profiles/r06_isa_counts_aot.json]

Per kernel: VGPRs, SGPR / VGPR spill counts and scratch bytes from the code object's metadata, and for its LARGEST loop
(the K loop of the K-step kernels, the body loop of k_delete_ahead_live, the tile loop elsewhere) the VALU instructions and
how many of them are v_readlane / v_writelane -- SGPR spills live in VGPR lanes, so a restore inside a loop is a 4-cycle VALU
instruction in a kernel bound by VALU issue.  The assembly is what tools/isa_count.py --keep DIR --aot ... leaves behind
(hipcc -S with physicl_amd/build.py's FLAGS); without --asm it is compiled here (about a minute)."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*$", "", re.sub(r"^void \(anonymous namespace\)::|^void ", "", o)) for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--json", default=os.path.join(ROOT, "profiles", "r06_isa_counts_aot.json"))
    a = ap.parse_args()
    from physicl_amd import build
    asm_path = a.asm
    if not asm_path:
        build._generate_rtc_source()
        asm_path = os.path.join(tempfile.mkdtemp(), "aot.s")
        subprocess.check_call([build.HIPCC] + [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] +
                              ["--cuda-device-only", "-S", "-o", asm_path, build.SOURCES[0]], stderr=subprocess.DEVNULL)
    lines = open(asm_path).read().split("\n")
    starts = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.match(r"(_Z\w+):", ln)] if m]
    meta = {}
    text = "\n".join(lines)
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(.*?)\.wavefront_size", text, re.S):
        blk = m.group(2)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        meta[m.group(1)] = {"vgprs": get("vgpr_count"), "sgpr_spill": get("sgpr_spill_count"), "vgpr_spill": get("vgpr_spill_count"),
                            "scratch": get("private_segment_fixed_size")}
    names = [n for _, n in starts if n in meta]
    pretty = dict(zip(names, demangle(names)))
    out = {}
    for st, name in starts:
        if name not in meta:
            continue
        en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
        f = lines[st:en]
        lab = {m.group(1): i for i, ln in enumerate(f) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
        best = None
        for i, ln in enumerate(f):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in lab and lab[m.group(1)] < i and (best is None or i - lab[m.group(1)] > best[1] - best[0]):
                best = (lab[m.group(1)], i)
        rec = dict(meta[name])
        rec["v_readlane"] = sum("v_readlane" in ln for ln in f)
        rec["v_writelane"] = sum("v_writelane" in ln for ln in f)
        if best:
            seg = f[best[0]:best[1] + 1]
            rec["largest_loop"] = {"valu": sum(bool(re.match(r"\s+v_", ln)) for ln in seg),
                                   "v_readlane": sum("v_readlane" in ln for ln in seg),
                                   "v_writelane": sum("v_writelane" in ln for ln in seg)}
        out[pretty[name]] = rec
    total = {"kernels": len(out), "kernels_with_sgpr_spills": sum(r["sgpr_spill"] > 0 for r in out.values()),
             "sgprs_spilled": sum(r["sgpr_spill"] for r in out.values()), "kernels_with_vgpr_spills": sum(r["vgpr_spill"] > 0 for r in out.values()),
             "kernels_with_scratch": sum(r["scratch"] > 0 for r in out.values()), "v_readlane": sum(r["v_readlane"] for r in out.values()),
             "v_readlane_in_largest_loops": sum(r.get("largest_loop", {}).get("v_readlane", 0) for r in out.values())}
    doc = {"csrc_sha": build.csrc_sha(), "options": " ".join(f for f in build.FLAGS if f not in ("-shared", "-fPIC")),
           "total": total, "kernels": dict(sorted(out.items()))}
    json.dump(doc, open(a.json, "w"), indent=1)
    print(json.dumps(total))


if __name__ == "__main__":
    main()
