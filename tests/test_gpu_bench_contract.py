"""GPU: bench.py prints ONE JSON line that carries the contract's keys (small N so it runs in seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--photons", "300000", "--steps", "6",
                                   "--warmup", "3", "--cpu-photons", "20000", "--cpu-seconds", "0.5", *extra], cwd=ROOT)
    lines = [ln for ln in out.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [(), ("--steps-per-launch", "1"), ("--mode", "separate"), ("--dtype", "f32", "--steps-per-launch", "4")])
def test_bench_line_has_the_contract_keys(extra):
    d = run_bench(*extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "particle-steps/sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 / 300000 - 1.0) < 1e-9      # value = N / time per step
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["counters_last_step"]["N"] == 300000
    if "--dtype" not in extra:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == d["unit"]
    if not extra:
        assert d["config"]["steps_per_launch"] == 32 and d["single_step"]["roofline"]["bound"] == "hbm"
        assert d["hbm_target"]["target"] == 0.6


def test_counters_do_not_depend_on_steps_per_launch_or_mode():
    a = run_bench("--no-cpu-baseline")["counters_last_step"]
    b = run_bench("--no-cpu-baseline", "--steps-per-launch", "1")["counters_last_step"]
    c = run_bench("--no-cpu-baseline", "--mode", "separate")["counters_last_step"]
    assert a == b == c
