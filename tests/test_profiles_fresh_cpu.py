"""CPU: the counter records bench.py quotes were measured on the device sources that are in the tree.

PMC counters cannot be read from inside a process, so ``traffic`` / ``valu_busy`` in bench.py's records come from a
committed rocprofv3 run (profiles/pmc_traffic.json).  Every record carries ``csrc_sha`` = physicl_amd.build.csrc_sha() of the
sources it was measured on; bench.py drops a record whose value differs (the line then says ``traffic: null``), and this
test fails until the profile has been taken again (tools/prof_driver_cmd.sh + tools/summarize_driver_prof.py) -- so a kernel
change cannot leave a stale counter figure in the line unnoticed."""
import json
import os

from physicl_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_counter_record_was_measured_on_the_device_sources_in_the_tree():
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    sha = build.csrc_sha()
    assert table.get("kernels"), "no per-kernel counter records"
    stale = {k: v.get("csrc_sha") for k, v in table["kernels"].items() if v.get("csrc_sha") != sha}
    stale.update({k: v.get("csrc_sha") for k, v in table.items() if k != "kernels" and v.get("csrc_sha") != sha})
    assert not stale, ("records measured on other device sources than %s -- profile again: bash tools/prof_driver_cmd.sh <tag> && "
                       "python tools/summarize_driver_prof.py <tag>: %r" % (sha, stale))
    for name in ("pcl_rtc_fast_e1", "pcl_rtc_multi2s_e1", "k_delete_ahead_live<double>", "k_mixed valu f64"):
        assert name in table["kernels"], name


def test_the_instruction_mixes_are_those_of_the_sources_in_the_tree():
    """profiles/isa_counts.json is produced offline from the same sources (tools/isa_count.py): its records of the K-step code
    objects carry the priced form bench.py multiplies out."""
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    k = isa["0.000000001 * exp(r0[gid] - 5)"]["kernels"]["pcl_rtc_multi2s_e1"]
    assert sum(k["dense_pass_classes"].values()) == k["dense_pass_valu"]
    assert 2.0 <= k["decision_cycles_per_valu"] <= 4.0 and k["dense_pass_cycles"] >= 2 * k["dense_pass_valu"]
    assert isa.get("csrc_sha") == build.csrc_sha(), "tools/isa_count.py has not been run since the device sources changed"
    # the code object the driver's command runs has its dynamic decision count calibrated against SQ_INSTS_VALU (without it the
    # static count, an upper estimate, would price the launch above the cycles there were)
    assert any("calibration" in key for key in k), "profile again: the K-step pass of the driver's command has no calibrated count"
    assert k["decision_valu_per_wave_step_shortcut"] < k["decision_valu_per_wave_step_static"]
    # ... and the four counts of k_delete_ahead_live (tools/prof_calib_ahead.sh + tools/summarize_calib_ahead.py)
    assert isa["k_delete_ahead_live<double>"].get("csrc_sha") == build.csrc_sha(), "fit k_delete_ahead_live's counts again"
