"""CPU: the stdout line bench.py prints is a bounded projection of the run's record (no GPU: committed full records
of earlier runs are projected again).  Round 4's line had grown to 31 kB and the driver, which keeps the last ~8 000
characters of stdout, could not parse it."""
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


RECORDS = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r0[45]_bench_*.json")) if "_bench_line_" not in p)
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[56]_bench_line_*.json")))


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_contract_line_of_a_full_record_fits_the_drivers_window(path):
    b = bench_module()
    rec = json.load(open(path))
    if "config" not in rec or "roofline" not in rec:
        pytest.skip("not a full record")
    line = b.contract_line(rec, "bench_detail.json")
    s = json.dumps(line)
    assert len(s) < b.LINE_LIMIT == 4000
    back = json.loads(("x" * 9000 + "\n" + s + "\n").rstrip()[-8000:].splitlines()[-1])     # the driver's view
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_hbm", "cpu_baseline", "collective", "detail_file"):
        assert k in back, k
    assert back["value"] == rec["value"] and back["roofline"]["frac"] == rec["roofline"]["frac"]
    assert "model" not in back["config"] and "workload" in back["config"]
    for k in ("per_block", "instruction_counts", "k_step_launch_work", "delete", "api", "mixed"):
        assert k not in back and k not in back["roofline"]


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_the_committed_lines_are_what_the_driver_can_read(path):
    """The stdout lines of the round's own runs (driver's command, default, 2-rank rehearsal), as printed."""
    text = open(path).read()
    assert text.count("\n") == 1 and len(text) < 4000
    line = json.loads(text[-8000:].splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "roofline_hbm",
              "cpu_baseline", "collective", "detail_file"):
        assert k in line, k
    r = line["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "SIMD-cycles/s" and 0 < r["frac"] <= 1.0 and 1.5 < r["clock_GHz"] < 2.6
    assert isinstance(r["traffic"], (int, float)) and abs(r["traffic"] / line["config"]["photons_per_gpu"] - 128) < 2
    if line["n_gpus"] == 1:
        assert line["roofline_hbm"]["frac"] >= 0.6 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1


def test_line_sheds_optional_parts_rather_than_outgrow_the_window():
    b = bench_module()
    rec = json.load(open(RECORDS[0]))
    rec["config"]["workload"] = "w" * 1500
    rec["cpu_baseline"] = dict(rec.get("cpu_baseline") or {}, sample="s" * 900, value=1.0, cores=1, kind="port")
    s = json.dumps(b.contract_line(rec, "bench_detail.json"))
    assert len(s) < 4000 and "roofline" in json.loads(s)


def test_detail_file_is_written_atomically_and_is_the_full_record(tmp_path, monkeypatch):
    b = bench_module()
    rec = json.load(open(RECORDS[0]))
    monkeypatch.setenv("PCL_BENCH_DETAIL", str(tmp_path / "d.json"))
    rel = b.write_detail(rec)
    assert json.load(open(tmp_path / "d.json")) == rec and rel.endswith("d.json")
    assert not [p for p in os.listdir(tmp_path) if ".tmp" in p]
