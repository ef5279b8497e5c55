"""GPU: bench.py prints ONE JSON line that carries the contract's keys (small N so it runs in seconds)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "roofline_hbm", "cpu_baseline", "collective", "detail_file")


def check_line(stdout, only_line=True):
    """What the driver does with bench.py's stdout: it keeps the last ~8 000 characters and parses the last line.  The
    line must survive that (round 4's grew to 31 kB and did not), carry the contract's keys and stay under 4 000 characters."""
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 or (lines and not only_line), lines
    assert len(lines[-1]) < 4000, len(lines[-1])
    line = json.loads(stdout.rstrip()[-8000:].splitlines()[-1])
    for k in LINE_KEYS:
        assert k in line, k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert set(line["config"]) <= {"workload", "photons_per_gpu", "profile", "mode", "steps_per_launch_max", "steps_per_timed_launch",
                                   "variable_n_fn", "dt", "rng", "parallelism"} and "workload" in line["config"]
    assert line["roofline"]["traffic"] is None or isinstance(line["roofline"]["traffic"], (int, float))
    return line


def run_bench(*extra, photons="300000"):
    """Runs bench.py, checks the stdout line as the driver would read it, and returns the run's FULL record (the detail
    file bench.py writes beside the line) with the parsed line under "_line"."""
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "bench_detail.json")
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--photons", photons, "--steps", "6",
                                       "--warmup", "3", "--repeats", "3", "--cpu-photons", "20000", "--cpu-seconds", "0.5",
                                       "--delete-photons", "30000", "--iso-photons", "30000",
                                       "--mixed-photons", "30000", *extra], cwd=ROOT, env=dict(os.environ, PCL_BENCH_DETAIL=detail),
                                      stderr=subprocess.DEVNULL)
        line = check_line(out.decode())
        d = json.load(open(detail))
    # the line is a projection of the detail record: same headline, same roofline figures
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "data"):
        assert line[k] == d[k], k
    for k in ("bound", "achieved", "peak", "unit", "frac"):
        assert line["roofline"][k] == d["roofline"][k], k
    assert line["detail_file"].endswith("bench_detail.json")
    d["_line"] = line
    return d


@pytest.mark.parametrize("extra", [(), ("--steps-per-launch", "1"), ("--mode", "separate"), ("--dtype", "f32", "--steps-per-launch", "4")])
def test_bench_line_has_the_contract_keys(extra):
    d = run_bench(*extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "repeats", "repeat_ms_per_step"):
        assert k in d, k
    assert d["metric"] == "particle-steps/sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 / 300000 - 1.0) < 1e-9      # value = N / time per step
    assert d["repeats"] == 3 and len(d["repeat_ms_per_step"]) == 3
    assert min(d["repeat_ms_per_step"]) <= d["ms_per_step"] + 1e-4 <= max(d["repeat_ms_per_step"]) + 2e-4       # the median block
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["slab_selection"]["candidates_GBps"] == []        # a 300000-photon store is too small to be chosen among candidates
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if not extra:
        # the K-step pass says what binds it: VALU issue.  achieved = (decision instructions x wave-steps + dense-pass
        # instructions x dense passes, both tallied by the kernel itself) / launch time; the HBM form rides along
        assert r["bound"] == "valu" and r["unit"] == "SIMD-cycles/s"
        # the ceiling is in SIMD-cycles at the clock the launch HELD (measured in the kernel), not at the data sheet's 2.4 GHz
        assert 1.0 < r["clock_GHz"] < 2.6 and abs(r["peak"] - 1024 * r["clock_GHz"] * 1e9) < 1e-3 * r["peak"]
        assert 0 < r["frac"] <= 1.0 and r["frac"] <= r["frac_at_4_waves_per_simd"] <= 1.1        # priced issue work never exceeds the cycles there were
        assert 0 < r["lane_util"] <= 1 and r["wave_steps"] in (-(-300000 // 128) * 6, -(-300000 // 256) * 6, -(-300000 // 192) * 6) and r["dense_passes"] > 0
        (form, ic), = r["instruction_counts"].items()      # the median block's one launch: one code object
        assert form in ("pcl_rtc_multi_e1", "pcl_rtc_multis_e1", "pcl_rtc_multi2_e1", "pcl_rtc_multi2s_e1", "pcl_rtc_multi3_e1", "pcl_rtc_multi3s_e1")
        assert r["kernel_forms"] == {form: 1}
        sat, ws, trips = r["saturated_wave_steps"], r["wave_steps"], r["wave_steps"] / 6.0   # (6 steps per launch in this run)
        A, As = ic["decision_valu_per_wave_step"], ic.get("decision_valu_per_wave_step_shortcut", ic["decision_valu_per_wave_step"])
        T, Ts = ic.get("decision_valu_per_wave_trip", 0.0), ic.get("decision_valu_per_wave_trip_shortcut", 0.0)
        dec = (A * (ws - sat) + As * sat + (Ts * sat / ws + T * (1 - sat / ws)) * trips) if sat >= 0 and form.endswith("s_e1") else A * ws + T * trips
        assert abs(r["wave_instructions"] - (dec + ic["dense_pass_valu"] * r["dense_passes"])) < 2 and 0 <= sat <= ws
        # ... priced: the decision part at the mean price of its mix, the dense pass at its own cycles (2 / 4 / 8 / 16 per class)
        cyc = dec * ic["decision_cycles_per_valu"] + ic["dense_pass_cycles"] * r["dense_passes"]
        assert abs(r["issue_cycles"] - cyc) <= 1e-9 * cyc and abs(r["achieved"] - cyc / (r["avg_launch_ms"] * 1e-3)) <= 1e-6 * r["achieved"]
        assert sum(ic["dense_pass_classes"].values()) == ic["dense_pass_valu"] and 2.0 < ic["decision_cycles_per_valu"] < 4.0
        assert r["hbm"]["peak"] == 8000.0 and r["hbm"]["algorithmic_bytes_per_particle"] == 128.0 and len(r["per_block"]) == 3
        if r["traffic"] is not None:                        # (None: the kernels have changed since the committed counter run)
            assert r["traffic"]["source"].startswith("profiles/") and abs(r["traffic"]["bytes"] / 300000 - 128) < 2
            assert r["valu_busy"] is None or (0 < r["valu_busy"] <= 1 and abs(r["useful"] - r["valu_busy"] * r["lane_util"]) < 1e-12)
        h = d["roofline_hbm"]                                # the north_star kernel's own record, in the line and not in an extra key
        assert h["bound"] == "hbm" and h["peak"] == 8000.0 and h["algorithmic_bytes_per_particle"] == 104.0 and h["frac"] > 0
        assert h["traffic"] is None or abs(h["traffic"]["bytes_per_unit"] - 104) < 1
        assert h == dict(d["single_step"]["roofline"], value=h["value"], ms_per_step=h["ms_per_step"])
        t = d["tame"]                                        # SURVEY 8(d) config 3's second profile
        assert t["roofline"]["bound"] == "valu" and "exp(r2[gid] / 8600.0)" in t["workload"] and len(t["repeat_hit_fraction"]) == 3
        assert 0 < min(t["repeat_hit_fraction"]) and max(t["repeat_hit_fraction"]) < 1 and t["value"] > 0
    else:
        assert r["bound"] == "hbm" and r["peak"] == 8000.0
    assert r["launches"] == (1 if "--mode" not in extra else 6) or "--steps-per-launch" in extra        # of the median block
    assert d["counters_last_step"]["N"] == 300000 and d["collective"] is None
    if "--dtype" not in extra:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == d["unit"]
        for key in ("cpu_baseline_python", "cpu_baseline_python_1e5", "cpu_baseline_numpy"):   # BASELINE.md section 4's other sizes
            assert d[key]["cores"] == 1 and d[key]["value"] > 0
        for size, rec in d["delete"]["sizes"].items():
            for mode in ("per_step", "multi"):
                assert rec[mode]["value"] > 0 and rec[mode]["particle_steps"] >= 30000
            assert rec["per_step"]["particle_steps"] == rec["multi"]["particle_steps"]       # same photons removed at the same steps
            # k_delete_alive's own record comes from the run with the bodies-ahead path off; with it on, a store of this size
            # never launches that kernel: k_delete_ahead answers every body
            assert rec["per_step_no_ahead"]["roofline"]["bound"] == "hbm" and rec["per_step_no_ahead"]["roofline"]["achieved"] > 0
            assert rec["per_step_no_ahead"]["particle_steps"] == rec["per_step"]["particle_steps"]
            assert rec["per_step_no_ahead"]["bodies_answered_by"]["ahead_launch"] == 0
            assert rec["per_step"]["ahead"]["launches"] >= 1 and rec["per_step"]["ahead"]["achieved_GBps"] > 0
            by = rec["per_step"]["bodies_answered_by"]
            assert by["kernel"] + by["ahead_launch"] + by["ahead"] == rec["per_step"]["loop_bodies"] and by["ahead"] > by["ahead_launch"] >= 1
            # the run's own record is the kernel that did its work: k_delete_ahead_live, priced by its own tally
            for mode in ("per_step", "multi"):
                v = rec[mode]["roofline"]
                assert v["bound"] == "valu" and v["unit"] == "SIMD-cycles/s" and 0 < v["frac"] <= 1.0 and 1.0 < v["clock_GHz"] < 2.6
                assert abs(v["peak"] - 1024 * v["clock_GHz"] * 1e9) < 1e-3 * v["peak"]
                w, c = v["work"], v["instruction_counts"]
                instr = (c["valu_per_group_first_pass_two_bodies"] * w["groups_of_128_slots_first_pass_two_bodies"]
                         + c["valu_per_group_first_pass_one_body"] * w["groups_of_128_slots_first_pass_one_body"]
                         + c["valu_per_round_two_bodies"] * w["rounds_two_bodies"] + c["valu_per_round_one_body"] * w["rounds_one_body"])
                assert abs(instr - v["wave_instructions"]) <= 1e-9 * instr
                assert abs(v["achieved"] - instr * v["cycles_per_wave_instruction"] / (v["total_ms"] * 1e-3)) <= 1e-6 * v["achieved"]
                assert w["groups_of_128_slots_first_pass_two_bodies"] >= 30000 // 128 and w["rounds_two_bodies"] >= 1
            assert rec["per_step"]["roofline_alive"]["bound"] == "hbm"
        assert d["api"]["steps_per_launch_32"]["steps"] == 6 and d["api"]["steps_per_launch_1"]["rows"] == 6
        # the constructor as a reference script calls it takes the K-pass launches by itself
        assert d["api"]["default"]["steps"] == 6 and d["api"]["default"]["schedule"] == {"fused_multi": 1} and d["api"]["default"]["note"] is None
        assert d["api"]["steps_per_launch_1"]["schedule"] == {"fused": 6}
        assert d["api"]["delete_default"]["schedule"].get("fused_delete_multi", 0) >= 1 and d["api"]["delete_default"]["particle_steps"] >= 30000
        iso = d["iso_1e7"]
        assert iso["per_step"]["roofline"]["algorithmic_bytes_per_particle"] == 96.0 and iso["multi"]["value"] > 0
        assert abs(iso["per_step"]["hit_fraction"] - iso["multi"]["hit_fraction"]) < 1e-12        # same photons, same decisions
        mx = d["mixed"]
        assert mx["iterations"] == 100 and mx["seconds_f64"] > 0 and mx["seconds_f32"] > 0 and set(mx["fp32_vs_fp64"]) == {"1", "10", "100"}
        # k_mixed keeps no tally of its own: its VALU record is the committed PMC run's, and says so
        assert mx["roofline"] is None or (mx["roofline"]["bound"] == "valu" and 0 < mx["roofline"]["frac"] <= 1.0 and
                                          0 < mx["roofline"]["valu_busy"] < 1.2 and "not this process" in mx["roofline"]["note"])
        assert mx["fp32_vs_fp64"]["1"]["decision_mismatch_rate"] < 1e-3
    if not extra:
        assert d["config"]["steps_per_launch_max"] == 32 and d["config"]["steps_per_timed_launch"] == 6
        assert d["single_step"]["roofline"]["bound"] == "hbm" and d["roofline"]["launches"] == 1      # the median block's one launch
        assert d["hbm_target"]["target"] == 0.6 and d["single_step"]["repeats"] == 3


def test_counters_do_not_depend_on_steps_per_launch_or_mode():
    a = run_bench("--no-cpu-baseline", "--no-extra")["counters_last_step"]
    b = run_bench("--no-cpu-baseline", "--no-extra", "--steps-per-launch", "1")["counters_last_step"]
    c = run_bench("--no-cpu-baseline", "--no-extra", "--mode", "separate")["counters_last_step"]
    assert a == b == c


def test_two_ranks_started_by_bench_itself_gloo_rehearsal_on_one_device():
    """``python bench.py --gpus 2`` launched directly (no torchrun): the parent starts the two ranks, rank 0's line comes
    back, the line says which collective backend carried the counters, and the global counters are those of one
    process running both shards' photons (ids are global: results do not depend on the sharding)."""
    two = run_bench("--gpus", "2", "--backend", "gloo", "--device", "0", photons="150000")
    assert two["n_gpus"] == 2 and two["collective"]["backend"] == "gloo" and two["collective"]["ranks_seen"] == 2
    assert [d["rank"] for d in two["collective"]["devices"]] == [0, 1]
    assert "gloo all-reduce" in two["config"]["workload"] and "cpu_baseline" not in two and "single_step" not in two
    # every rank's own block times and the memory its store got are in the line: a straggler or an unlucky slab shows
    c = two["collective"]
    assert len(c["per_rank_block_ms_per_step"]) == 2 and all(len(r) == 3 for r in c["per_rank_block_ms_per_step"])
    assert all(lo <= hi for lo, hi in zip(c["block_min_ms_per_step"], c["block_max_ms_per_step"]))
    assert max(c["block_max_ms_per_step"]) <= max(two["repeat_ms_per_step"]) + 1e-3
    assert all("slab_selection" in d and "pci" in d for d in c["devices"])
    one = run_bench("--no-cpu-baseline", "--no-extra")
    assert two["counters_last_step"] == one["counters_last_step"]


def test_rccl_backend_on_one_device_is_an_error_not_a_silent_fallback():
    """Two ranks on ONE device make RCCL refuse; with the default --backend nccl that must fail the whole job."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "0", "--photons", "100000",
                        "--steps", "2", "--warmup", "1", "--repeats", "1"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "RCCL process group could not be brought up" in p.stderr or "CollectiveError" in p.stderr, p.stderr[-3000:]


def test_two_ranks_started_by_torchrun_the_way_the_driver_documents_it():
    """``python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py
    --gpus 2 ...``: the ranks find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment and do not spawn again."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                                   "--gpus", "2", "--backend", "gloo", "--device", "0", "--photons", "150000", "--steps", "6",
                                   "--warmup", "3", "--repeats", "2"], cwd=ROOT, stderr=subprocess.DEVNULL, timeout=900)
    d = check_line(out.decode(), only_line=False)     # (torchrun may write lines of its own; the driver reads the last one)
    assert d["n_gpus"] == 2 and d["collective"]["backend"] == "gloo" and d["collective"]["ranks_seen"] == 2
    assert d["counters_last_step"]["N"] == 300000 and d["repeats"] == 2


def test_dry_run_brings_the_ranks_and_the_collective_up_and_stops():
    """``bench.py --gpus N --dry-run``: ranks, process groups and one all-reduce, every rank's device in the line, no store."""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0",
                                   "--dry-run"], cwd=ROOT, timeout=600)
    lines = [ln for ln in out.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["ok"] is True and d["n_gpus"] == 2 and d["collective"]["ranks_seen"] == 2
    assert [x["rank"] for x in d["collective"]["devices"]] == [0, 1] and d["distinct_pci"] == 1      # both ranks on the one GPU here
