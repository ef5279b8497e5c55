"""GPU: the drop-in boundary is a C ABI -- a host written in plain C binds it with nothing but include/physicl_hip.h.

tests/native/abi_consumer.c is compiled with gcc (-std=c99 -pedantic: the header must be C, not C++), linked against
libphysicl_hip.so and run; what it prints -- the flags of both delete kernels (physicl/light.py:146-158, 239-249), the
stable survivor indices (light.py:258-260), r and dr after one Newton step on the resident store (newton.py:15-16), the
sign counters (light.py:424-426) -- is compared bit for bit with the CPU oracle on the same inputs, generated on both
sides by the same 64-bit LCG.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import physicl_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "abi_consumer.c")
GROUP_SRC = os.path.join(ROOT, "tests", "native", "abi_group_consumer.c")
LIBDIR = os.path.join(ROOT, "physicl_amd", "_lib")


def build(tmp_path, src=SRC, name="abi_consumer"):
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-I", os.path.join(ROOT, "include"),
                           src, "-o", exe, "-L", LIBDIR, "-lphysicl_hip", "-Wl,-rpath," + LIBDIR, "-lm"])
    return exe


def lcg_inputs(N):
    state, mask = 0x9E3779B97F4A7C15, (1 << 64) - 1
    c, dt = 299792458.0, 1e-3
    cols = np.empty((4, N))
    for i in range(N):
        for k in range(4):
            state = (state * 6364136223846793005 + 1442695040888963407) & mask
            u = float(state >> 11) * (1.0 / 9007199254740992.0)
            cols[k, i] = (2.0 * u - 1.0) * c * dt if k < 3 else u
    return cols


@pytest.mark.skipif(shutil.which("gcc") is None or not os.path.exists(os.path.join(LIBDIR, "libphysicl_hip.so")),
                    reason="needs gcc and the built library")
def test_header_is_plain_c_and_the_program_links(tmp_path):
    """No GPU needed: C99 -pedantic -Werror compile of the header's user and a link against every symbol it uses."""
    assert os.path.exists(build(tmp_path))
    assert os.path.exists(build(tmp_path, GROUP_SRC, "abi_group_consumer"))


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 4099])
def test_c_host_program_gets_the_oracles_results(tmp_path, N):
    exe = build(tmp_path)
    out = subprocess.check_output([exe, str(N)], timeout=300).decode().splitlines()
    rec = {ln.split(" ", 1)[0]: ln.split(" ", 1)[1] for ln in out}
    d0, d1, d2, rand = lcg_inputs(N)
    A, n, dt = 1e-3, 1e-3, 1e-3
    flags = orc.delete_flags(d0, d1, d2, rand, A, n)
    want = "".join("1" if f else "0" for f in flags)
    assert rec["flags"] == want and rec["del"] == want and (N < 100 or 0 < flags.sum() < N)
    keep = orc.survivors(flags)
    assert rec["keep"] == "%d %d" % (len(keep), int(keep.sum()))
    h = [d0, d1, d2]
    v = [h[k] / dt for k in range(3)]
    r = [h[(k + 1) % 3] for k in range(3)]
    rn, dr = orc.newton_euler(r, v, dt)
    hexes = lambda arrs: " ".join("%016x" % x for a in arrs for x in np.asarray(a, dtype=np.float64).view(np.uint64))
    assert rec["r"] == hexes(rn)
    assert rec["dr"] == hexes(dr)
    assert rec["counters"] == "%d %d %d %d" % ((N,) + orc.sign_counts(v))
    # the tracked subset's rows (pcl_store_trace_ahead from C) are the Python binding's on the same store
    from physicl_amd import _hip as hip
    ids = [0, N - 1] if N > 1 else [0]
    with hip.Device(0) as dev:
        dev.store_alloc(N)
        dev.upload_state({"r": np.stack(rn, 1), "v": np.stack(v, 1), "dr": np.stack(dr, 1), "E": rand})
        rows = dev.trace_ahead(ids, dt, 3, ("iso",), 0, dict(A=A, n=n, flags=0, c=299792458.0, h=6.62607015e-34), None, 9, 5)
    assert rec["trace"] == " ".join("%016x" % x for x in rows.reshape(-1).view(np.uint64)) and rec["traced"] == str(len(ids))


@pytest.mark.gpu
def test_c_host_program_on_a_device_group_gets_the_one_context_rows(tmp_path):
    """pcl_group_*: several contexts in one process behind the C ABI (SURVEY.md 8(b) ``pcl_init(n_dev, dev_ids)``).  The
    plain-C program prints the rows of a K-step scatter launch, of a delete-until-empty run one call per body, and the
    survivors' ids in global order; with 1, 2 and 3 contexts on device 0 the output must be identical line for line,
    and the one-context rows are those of the Python binding on a single ``Device``."""
    from physicl_amd import _hip as hip
    exe = build(tmp_path, GROUP_SRC, "abi_group_consumer")
    N = 150_001
    outs = [subprocess.check_output([exe, str(N), str(g)], timeout=600).decode() for g in (1, 2, 3)]
    assert outs[0] == outs[1] == outs[2]
    lines = outs[0].splitlines()
    iso = [[int(x) for x in ln.split()[2:]] for ln in lines if ln.startswith("iso")]
    dele = [[int(x) for x in ln.split()[2:]] for ln in lines if ln.startswith("del")]
    with hip.Device(0) as d:
        d.store_alloc(N)
        d.fill_photons(N, 1000, 299792458.0, 2.8e-19, 9.9e-19, 77)
        rows = d.step_fused_multi(1e-3, 6, dict(A=1e-3, n=1e-3, flags=0, c=299792458.0, h=6.62607015e-34, seed=77, step=1))
        assert iso == [[o["N"]] + [int(x) for x in o["sign"]] + [o["hits"]] for o in rows]
        for body in range(len(dele)):
            o = d.step_fused_delete(1e-3, 1e-3, 1e-3, hip.RNG_PHILOX, 77, 100 + body, [[2.0e6, np.nan, np.nan]], lazy=True)
            assert dele[body] == [o["N"]] + [int(x) for x in o["sign"]] + [int(o["planes"][0]), o["removed"]], body
    assert 0 < iso[0][4] < N and dele[0][5] > 0 and int([ln for ln in lines if ln.startswith("bodies")][0].split()[1]) > 20
