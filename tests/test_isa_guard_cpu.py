"""CPU (hipcc cross-compiles gfx950 here): the tuned ISA is still what DESIGN.md section 4.2 describes.

The K-step kernels are bound by VALU issue and were tuned to one compiler: ``-mllvm -disable-machine-licm``, register pinning
through ``asm volatile("" : "+s"...)`` and ``__builtin_amdgcn_sched_barrier`` placement keep the loops free of scratch, of VGPR
spills and (nearly) of v_readlane restores, at four waves per SIMD (five for the 192-photon form with the saturation probe).
A ROCm bump that undoes any of that costs 10 % without failing a parity test -- so it fails this one:

* the hipRTC translation unit of the bench's expression (what ``get_rtc`` compiles, same options): every K-step specialisation
  has 0 scratch bytes, 0 spilled VGPRs, enough free registers for its waves per SIMD, and at most 12 v_readlane inside its K loop
  (5-9 as tuned, profiles/isa_counts.json; 79-108 with the machine-LICM pass on, profiles/r05_ab_machine_licm.md);
  the mixed-loop and trace kernels of the same unit have no scratch and no VGPR spill;
* the ahead-of-time library (every kernel, from its code-object metadata): no kernel with scratch or VGPR spills, the
  occupancy-critical ones within their register budget, the library's spilled-SGPR total below the level measured with the
  machine-LICM pass on (8500; 5676-6056 without).
"""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

EXPR = "0.000000001 * exp(r0[gid] - 5)"
VGPR_FILE = 512                       # registers per SIMD lane (arch + acc) on gfx950, allocated in blocks of 8


def waves_per_simd(vgprs):
    return min(8, VGPR_FILE // (-(-vgprs // 8) * 8))


@pytest.fixture(scope="module")
def rtc_asm():
    import isa_count
    with tempfile.TemporaryDirectory(prefix="pcl_isa_guard_") as work:
        yield isa_count, isa_count.compile_asm(isa_count.rtc_source(EXPR), work)


# kernel -> waves per SIMD it must keep (four; the 192-photon form with the probe: five)
KSTEP = {"pcl_rtc_multi_e1": 4, "pcl_rtc_multis_e1": 4, "pcl_rtc_multi2_e1": 4, "pcl_rtc_multi2s_e1": 4, "pcl_rtc_multi3_e1": 4,
         "pcl_rtc_multi3s_e1": 5}


@pytest.mark.parametrize("kernel", sorted(KSTEP))
def test_k_step_specialisations_keep_their_registers_and_loops(rtc_asm, kernel):
    isa_count, asm = rtc_asm
    rec = isa_count.analyse(asm, kernel)
    reg = rec["registers"]
    assert reg["private_segment_fixed_size"] == 0, reg              # nothing in scratch
    assert reg["vgpr_spill_count"] == 0, reg
    assert waves_per_simd(reg["vgpr_count"]) >= KSTEP[kernel], reg
    assert rec["readlane_in_k_loop"] <= 12, rec["readlane_in_k_loop"]  # SGPR spills restored inside the K loop are VALU work
    assert rec["readlane_writelane_in_dense_pass"] == 0
    assert rec["dense_pass_valu"] < 260 and rec["decision_valu_per_wave_step_static"] < 700   # (209 / 251-410 when tuned)


@pytest.mark.parametrize("kernel", ["pcl_rtc_mixed_e1", "pcl_rtc_mixed3_e1", "pcl_rtc_fast_e1", "pcl_rtc_fastg_e1", "pcl_rtc_trace_e1"])
def test_the_other_specialisations_have_no_scratch(rtc_asm, kernel):
    isa_count, asm = rtc_asm
    md = asm[asm.rindex(".name:           %s" % kernel):]
    get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, md).group(1))
    assert get("private_segment_fixed_size") == 0 and get("vgpr_spill_count") == 0
    if kernel == "pcl_rtc_mixed3_e1":
        assert waves_per_simd(get("vgpr_count")) >= 4


@pytest.fixture(scope="module")
def aot_meta():
    """{demangled kernel name: metadata} of the whole ahead-of-time library (one -S compile, about a minute; kept in /tmp per
    source hash so that a second run of the suite is free)."""
    from physicl_amd import build
    build._generate_rtc_source()
    path = os.path.join(tempfile.gettempdir(), "pcl_aot_%s.s" % build.csrc_sha())
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        subprocess.check_call([build.HIPCC] + [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] +
                              ["--cuda-device-only", "-S", "-o", tmp, build.SOURCES[0]], stderr=subprocess.DEVNULL)
        os.replace(tmp, path)
    text = open(path).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(.*?)\.wavefront_size", text, re.S):
        blk = m.group(2)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        meta[m.group(1)] = {"vgprs": get("vgpr_count"), "sgpr_spill": get("sgpr_spill_count"), "vgpr_spill": get("vgpr_spill_count"),
                            "scratch": get("private_segment_fixed_size")}
    names = sorted(meta)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    clean = lambda d: re.sub(r"\(.*$", "", d.replace("(anonymous namespace)::", "").replace("void ", ""))
    return {clean(d): meta[n] for n, d in zip(names, dem)}


def test_ahead_of_time_library_has_no_scratch_and_keeps_its_occupancy(aot_meta):
    assert len(aot_meta) > 200
    bad = {k: v for k, v in aot_meta.items() if v["scratch"] or v["vgpr_spill"]}
    assert not bad, bad
    assert sum(v["sgpr_spill"] for v in aot_meta.values()) < 7500      # (5676-6056 tuned; 8500 with the machine-LICM pass on)
    budget = {"k_mixed3<double, false>": 4, "k_mixed3<float, false>": 4, "k_multi<double, false, 0>": 4, "k_multi<double, true, 0>": 4,
              "k_mixed<double, false, 0>": 4, "k_delete_ahead_live<double, false>": 5, "k_delete_ahead_live<double, true>": 5,
              "k_multi3_e0": 5, "k_multi3_e1": 4}
    for name, waves in budget.items():
        assert name in aot_meta, (name, [k for k in aot_meta if k.startswith(name.split("<")[0])][:8])
        assert waves_per_simd(aot_meta[name]["vgprs"]) >= waves, (name, aot_meta[name])
