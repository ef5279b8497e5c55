"""CPU: how ``bench.py --gpus N`` starts its own ranks (physicl_amd/launch.py) -- environment of each rank, rank 0's
line forwarded, a failing rank fails the job and takes the others down, and the parent never loads torch or the HIP
library (it must not have touched the GPU when it starts the children)."""
import json
import os
import subprocess
import sys
import time

from physicl_amd.launch import rank_env, spawn_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys, time
r = int(os.environ["RANK"])
if os.environ.get("PCL_TEST_FAIL_RANK") == str(r):
    sys.exit(7)
if os.environ.get("PCL_TEST_HANG_RANK") == str(r):
    time.sleep(600)
rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                     "HSA_ENABLE_IPC_MODE_LEGACY")}
sys.stderr.write("rank %d up\n" % r)
if r == 0:
    print(json.dumps(rec))
else:
    print("noise from rank %d" % r)       # must not reach the parent's stdout
"""


def test_rank_env_sets_the_torchrun_variables():
    e = rank_env({"PATH": "/bin", "OMP_NUM_THREADS": "4"}, 3, 8, 29511)
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "29511")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["OMP_NUM_THREADS"] == "4" and e["PATH"] == "/bin"
    assert rank_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, 0, 1, 1)["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # caller's choice wins


def test_spawn_forwards_rank0_only_and_reports_success():
    rc, out = spawn_ranks(3, [sys.executable, "-c", CHILD])
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["RANK"] == "0" and rec["WORLD_SIZE"] == "3" and rec["MASTER_ADDR"] == "127.0.0.1" and int(rec["MASTER_PORT"]) > 0


def test_a_failing_rank_fails_the_job_and_stops_the_others():
    t0 = time.time()
    rc, out = spawn_ranks(3, [sys.executable, "-c", CHILD], env=dict(os.environ, PCL_TEST_FAIL_RANK="1", PCL_TEST_HANG_RANK="2"),
                          grace_s=5.0)
    assert rc == 7
    assert time.time() - t0 < 60          # rank 2 (sleeping "in a collective") was terminated, not waited for


def test_bench_parent_starts_ranks_without_loading_torch_or_the_hip_library():
    """bench.py --gpus 2 launched directly: the parent goes through spawn_ranks before importing anything that could
    initialise the GPU.  PCL_BENCH_TRACE_PARENT makes it report its own sys.modules on stderr; the children (no GPU in
    this container, or a deliberately impossible device) fail, so the job must exit non-zero and print no result."""
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--backend', 'gloo', '--device', '99', '--photons', '1000', '--steps', '1',\n"
            "            '--warmup', '0', '--no-cpu-baseline']\n"
            "try:\n"
            "    runpy.run_path(%r, run_name='__main__')\n"
            "except SystemExit as e:\n"
            "    bad = [m for m in ('torch', 'physicl_amd._hip') if m in sys.modules]\n"
            "    sys.stderr.write('PARENT_MODULES_BAD=%%r EXIT=%%r\\n' %% (bad, e.code))\n"
            "    raise\n") % os.path.join(ROOT, "bench.py")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "PARENT_MODULES_BAD=[]" in p.stderr, p.stderr[-3000:]


def test_a_signal_to_the_launcher_takes_the_ranks_down():
    """SIGTERM to the process that called spawn_ranks (a driver's timeout): its ranks are stopped, not left waiting in
    a collective.  The children write their PIDs; after the parent has gone none of them may be alive."""
    import signal
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        child = ("import os, time\n"
                 "open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                 "time.sleep(600)\n") % tmp
        parent = ("import sys\n"
                  "from physicl_amd.launch import spawn_ranks\n"
                  "rc, out = spawn_ranks(2, [sys.executable, '-c', %r], grace_s=5.0)\n"
                  "sys.exit(rc)\n") % child
        p = subprocess.Popen([sys.executable, "-c", parent], cwd=ROOT)
        t_end = time.time() + 60
        while time.time() < t_end and len(os.listdir(tmp)) < 2:
            time.sleep(0.05)
        pids = [int(open(os.path.join(tmp, f)).read()) for f in sorted(os.listdir(tmp))]
        assert len(pids) == 2
        p.send_signal(signal.SIGTERM)
        assert p.wait(timeout=60) == 128 + signal.SIGTERM
        time.sleep(0.2)
        for pid in pids:
            try:
                os.kill(pid, 0)
                alive = open("/proc/%d/stat" % pid).read().split()[2] != "Z"
            except (OSError, IOError):
                alive = False
            assert not alive, "rank process %d outlived its launcher" % pid
