import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


# ---------------------------------------------------------------------------------------------------------------------
# The library's A/B switches ("knobs", pcl_set_knob): which formulation runs must never show in a result.  Instead of
# re-running whole test files in child pytest processes under other environment variables (rounds 2-3), the tests that
# delete, and the tests of the K-step pass, are parametrised over the knob sets below: every case is its own test id.
# ---------------------------------------------------------------------------------------------------------------------
DELETE_KNOBS = [
    # compaction allowed at any size (the suite's stores are mostly below the 65536 slots where the alive path never
    # compacts by itself), pending moves flushed by a separate kernel, the host waits on the stream instead of polling
    ("any_size-flush_kernel-no_poll", {"PCL_ALIVE_MIN_SLOTS": "0", "PCL_ALIVE_FLUSH_KERNEL": "1", "PCL_ALIVE_POLL": "0"}),
    ("any_size-ratio_0.95", {"PCL_ALIVE_MIN_SLOTS": "0", "PCL_ALIVE_RATIO": "0.95"}),         # nearly every body compacts
    ("alive_off", {"PCL_ALIVE": "0"}),                                                         # the round-2 pipeline
    # k_compact_count takes every wave's survivors one by one (its form for sparse waves), whatever their number
    ("any_size-compact_survivor_major", {"PCL_ALIVE_MIN_SLOTS": "0", "PCL_COMPACT_SPARSE": "512"}),
]
# bodies worked out ahead (k_delete_ahead) only exist on the one-call-per-body path: the files that take it
AHEAD_KNOBS = [("ahead_off", {"PCL_AHEAD": "0"}), ("ahead_k3", {"PCL_AHEAD_K": "3"}),
               # the kernel that gives every slot its own lane, also where the one that lists the alive photons would run
               ("ahead_slot_per_lane", {"PCL_AHEAD_LIVE": "0"}),
               # every store takes the big stores' form: few bodies per launch, r left behind at the commit, compaction from the
               # committed masks
               ("ahead_big_form", {"PCL_AHEAD_MAX_SLOTS": "0", "PCL_ALIVE_MIN_SLOTS": "0"})]
# K-body calls on the K-step flag kernel + compaction (what stores with kinds always take): the files with such calls
MULTI_AHEAD_KNOBS = [("multi_flag_kernel", {"PCL_MULTI_AHEAD": "0"})]
MULTI_AHEAD_FILES = {"test_gpu_multi.py", "test_gpu_simulation.py", "test_gpu_random_programs.py"}
AHEAD_FILES = {"test_gpu_parity.py", "test_gpu_simulation.py", "test_gpu_random_programs.py", "test_gpu_fp32.py", "test_gpu_multi.py"}
KSTEP_KNOBS = [("256_per_wave", {"PCL_MULTI_NQ2": "1", "PCL_MULTI_NQ3": "0"}), ("128_per_wave", {"PCL_MULTI_NQ2": "0"}),
               # three photons per lane: the form a launch takes by itself when it starts at a hit fraction of 0.28 .. 0.355
               ("192_per_wave", {"PCL_MULTI_NQ3": "1"}),
               # the variant that tries exp's saturation shortcut wave by wave: always (both of its branches run: the suite's
               # photons start at the origin and fly out of exp's range), never
               ("saturation_probe", {"PCL_MULTI_SAT": "1", "PCL_MULTI_NQ2": "0"}), ("no_saturation_probe", {"PCL_MULTI_SAT": "0"})]
# the mixed K-pass kernel: three rows of 64 particles per wave and trip with the velocities in LDS (k_mixed3: what constant-n loops take by
# themselves below a hit probability of 0.33) forced for every constant-n loop, and never
MIXED_KNOBS = [("mixed_rows3", {"PCL_MIXED_NE3": "1"}), ("mixed_rows2", {"PCL_MIXED_NE3": "0"}),
               # round 6: loops with a delete phase compact each wave's survivors inside its own 512-slot segment and leave the global
               # compaction until half of the slots are dead (pcl_step_mixed_multi); "0": a compaction behind every launch, round 5's
               # form; PCL_MIXED_COMPACT_BELOW=0: in place and never a global compaction by itself (whoever needs the dense store asks)
               ("mixed_compact_every_launch", {"PCL_MIXED_INPLACE": "0"}), ("mixed_inplace_never_compact", {"PCL_MIXED_COMPACT_BELOW": "0"})]
MIXED_FILES = {"test_gpu_mixed.py", "test_gpu_random_programs.py", "test_gpu_trace.py"}
DELETE_FILES = {"test_gpu_parity.py", "test_gpu_multi.py", "test_gpu_mixed.py", "test_gpu_simulation.py", "test_gpu_random_programs.py",
                "test_gpu_fp32.py", "test_gpu_trace.py"}
KSTEP_FILES = {"test_gpu_multi.py", "test_gpu_bench_regime.py", "test_gpu_rtc_background.py"}


def pytest_generate_tests(metafunc):
    if "pcl_knobs" not in metafunc.fixturenames:
        return
    fname = os.path.basename(getattr(metafunc.module, "__file__", ""))
    node = (fname + "::" + metafunc.function.__name__).lower()
    sets = []
    if fname in DELETE_FILES and re.search("delete|random|mixed|program", node):
        # (the mixed K-pass kernel has its own delete phase: of the single calls' knobs it only meets the compaction's)
        sets += [k for k in DELETE_KNOBS if fname != "test_gpu_mixed.py" or "flush_kernel" not in k[0]]
        sets += (AHEAD_KNOBS if fname in AHEAD_FILES else []) + (MULTI_AHEAD_KNOBS if fname in MULTI_AHEAD_FILES else [])
        sets += MIXED_KNOBS if fname in MIXED_FILES else []
    if fname in KSTEP_FILES and "photons_per_wave" not in node:
        sets += KSTEP_KNOBS
    if sets:
        sets = [("default", {})] + sets
        metafunc.parametrize("pcl_knobs", [v for _, v in sets], ids=[k for k, _ in sets], indirect=True)


@pytest.fixture(autouse=True)
def pcl_knobs(request):
    """Sets the knobs of this case for the duration of the test -- through pcl_set_knob (this process) and the
    environment (processes the test starts) -- and takes them back afterwards."""
    knobs = getattr(request, "param", None) or {}
    if not knobs:
        yield knobs
        return
    from physicl_amd import _hip
    saved = {k: os.environ.get(k) for k in knobs}
    for k, v in knobs.items():
        _hip.set_knob(k, v)
        os.environ[k] = v
    try:
        yield knobs
    finally:
        for k, v in saved.items():
            _hip.set_knob(k, None)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
