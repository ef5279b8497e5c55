"""GPU: freed stores are kept for the next store of the process (a hipMalloc of tens of GB right after a hipFree of that
size stalls for seconds now and then on this runtime); the pool is bounded, can be emptied and can be switched off."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_freed_store_is_reused_by_the_next_one_of_that_size():
    from physicl_amd import _hip as hip
    hip.pool_trim()
    N = 700_000                                       # 17 rows x 8 B x N = 95 MB: above the pool's 64 MB threshold
    with hip.Device(0) as d:
        d.store_alloc(N)
        p0 = d.field_ptr(hip.R0)
        d.fill_photons(N, 0, 299792458.0, 1.0, 1.0, 3)
        r_before = d.download(hip.R0, 16)
        d.store_free()
        held = hip.pool_bytes()
        assert held >= N * 17 * 8
        d.store_alloc(N)                              # same size: the very block comes back
        assert d.field_ptr(hip.R0) == p0 and hip.pool_bytes() == 0       # nothing idle any more
        d.fill_photons(N, 0, 299792458.0, 1.0, 1.0, 3)
        assert np.array_equal(d.download(hip.R0, 16), r_before)
        d.store_free()
        d.store_alloc(50_000)                         # a small store does not take the big block
        assert hip.pool_bytes() == held
        d.store_free()
    with hip.Device(0) as d2:                         # another context of the process finds it too
        d2.store_alloc(N)
        assert d2.field_ptr(hip.R0) == p0
    assert hip.pool_trim() >= N * 17 * 8 and hip.pool_bytes() == 0


def test_pool_can_be_switched_off():
    code = ("from physicl_amd import _hip as hip\n"
            "d = hip.Device(0); d.store_alloc(700000); d.store_free(); assert hip.pool_bytes() == 0; d.store_alloc(700000)\n"
            "d.fill_photons(700000, 0, 299792458.0, 1.0, 1.0, 3); assert d.step_counters([])[0] == 700000; d.close(); print('ok')\n")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, PCL_POOL_GB="0"), timeout=300)
    assert out.decode().strip().endswith("ok")


def test_copies_over_a_slab_made_of_several_physical_handles():
    """Big blocks are hipMemMap'ed ranges of several physical handles (1.07 GB each; 240 tiles = 67 MB here), and hipMemcpy2DAsync refuses
    rows that reach from one handle into the next: uploads and downloads are cut at the handle boundaries."""
    code = ("import numpy as np\n"
            "from physicl_amd import _hip as hip\n"
            "N = 2_000_003\n"
            "rs = np.random.RandomState(1)\n"
            "d = hip.Device(0); d.store_alloc(N); d.set_count(N, 0)\n"
            "cols = {f: rs.normal(size=N) for f in (hip.R0, hip.R2, hip.V1, hip.E)}\n"
            "for f, a in cols.items(): d.upload(f, a)\n"
            "for f, a in cols.items(): assert np.array_equal(d.download(f, N), a), f\n"
            "assert np.array_equal(d.download(hip.R2, 70001, 1_234_567), cols[hip.R2][1_234_567:1_234_567 + 70001])\n"
            "d.upload(hip.V1, cols[hip.E][:500_000], 777_777); cols[hip.V1][777_777:1_277_777] = cols[hip.E][:500_000]\n"
            "assert np.array_equal(d.download(hip.V1, N), cols[hip.V1])\n"
            "d.step_newton(0.5)\n"
            "d.close(); print('ok')\n")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, PCL_VMM_CHUNK_TILES="240", PCL_POOL_GB="0"), timeout=300)
    assert out.decode().strip().endswith("ok")


def test_a_big_store_takes_the_fastest_of_a_few_candidate_slabs():
    """Stores of >= 512 MB are measured (13-row write sweep) against up to PCL_ALLOC_TRIES - 1 other candidates; the ones that
    lose wait in the pool, where the compaction's second slab takes the best of them; PCL_ALLOC_TRIES=1 takes the first block."""
    code = ("import os, sys, numpy as np\n"
            "from physicl_amd import _hip as hip\n"
            "N = 4_200_000\n"                                   # 17 rows x 8 B x N = 571 MB
            "d = hip.Device(0); d.store_alloc(N)\n"
            "slab = -(-N // 2048) * 2048 * 17 * 8\n"
            "held = hip.pool_bytes()\n"
            "d.fill_photons(N, 0, 299792458.0, 1.0, 1.0, 3)\n"
            "o = d.step_fused_delete(1e-3, 1e-3, 1e-3, hip.RNG_PHILOX, 3, 0, [], lazy=True)\n"   # needs the second slab
            "held2 = hip.pool_bytes()\n"
            "assert 0 < o['N'] < N\n"
            "d.close(); print('RESULT', held // slab, held2 // slab)\n")
    def run(tries):
        out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, PCL_ALLOC_TRIES=str(tries)), timeout=300)
        return [int(x) for x in out.decode().split()[-2:]]
    assert run(1) == [0, 0]
    held, held2 = run(4)
    assert held == 3 and held2 == 3          # three losers parked; the second slab took one of them and parked a fresh loser


def test_temporaries_of_big_stores_go_back_where_they_came_from():
    """The staging buffers of last_delete_flags / step_delete_flags / scatter_pcoll / fill_photons_table come from the
    same allocator as the store (big blocks: mapped virtual-memory ranges, pooled): freeing them any other way leaks the
    physical handles.  PCL_BIG_MIN_MB=1 takes a 300k-photon store through the big-block paths; free device memory plus
    what idles in the pool must not have shrunk after the calls."""
    code = ("import numpy as np\n"
            "from physicl_amd import _hip as hip\n"
            "N = 300_000\n"
            "d = hip.Device(0); d.store_alloc(N)\n"
            "grid = np.linspace(1.0, 2.0, 200_000); cdf = np.linspace(0.0, 1.0, 200_000)\n"
            "def avail():\n"
            "    d.sync(); f, t = d.mem_info(); return f + hip.pool_bytes()\n"
            "d.fill_photons_table(N, 0, 299792458.0, cdf, grid, 3)\n"      # 2 x 1.6 MB table: a big block under the knob
            "t0 = avail()\n"
            "for k in range(6): d.fill_photons_table(N, 0, 299792458.0, cdf, grid, 3)\n"
            "assert t0 - avail() <= (2 << 20), (t0, avail())\n"             # (one-sided: the driver may still be reclaiming an earlier process's memory)
            "d.step_newton(1e-3)\n"
            "o = d.step_fused_delete(1e-3, 1e-3, 1e-3, hip.RNG_PHILOX, 3, 0, [], lazy=True)\n"   # second slab, scratch
            "n = o['N']; d.sync()\n"
            "fl0 = d.last_delete_flags(N); assert fl0.sum() == N - n\n"            # from the alive masks, store untouched
            "assert d.slots == N and d.count == n\n"
            "d.download_ids(1)\n"                                                    # any other access compacts: second slab, ids
            "assert d.slots == n\n"
            "d.last_delete_flags(N); d.scatter_pcoll(1e-3, 1e-3, 0, 299792458.0, 6.6e-34)\n"   # warm: first-use module loads
            "a0 = avail()\n"
            "for k in range(6):\n"
            "    fl = d.last_delete_flags(N); assert np.array_equal(fl, fl0)\n"
            "    assert d.scatter_pcoll(1e-3, 1e-3, 0, 299792458.0, 6.6e-34).shape == (d.count,)\n"
            "for k in range(3):\n"
            "    m = d.count; flags = np.zeros(m, np.int32); flags[::3] = 1\n"
            "    alive, removed = d.step_delete_flags(flags); assert alive + removed == m\n"
            "a1 = avail()\n"
            "assert a0 - a1 <= (2 << 20), (a0, a1)\n"
            "d.close(); print('ok')\n")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, PCL_BIG_MIN_MB="1"), timeout=300)
    assert out.decode().strip().endswith("ok")


@pytest.mark.parametrize("trim", [False, True], ids=["handles_kept", "handles_released"])
def test_a_store_created_after_a_much_bigger_one_was_freed_keeps_what_kernels_wrote(trim):
    """Regression (round 5, profiles/r05_vmm_remap.log): after this process had unmapped and freed the address range of a
    109 GB store, hipMemAddressReserve handed the same addresses out again and the first kernels through the new mapping
    still hit translations of the old one -- 10 % to 76 % of what a fill kernel wrote was not in the new store.  A freed
    range now keeps its addresses reserved (never mapped twice) and its physical handles serve the next range
    (``handles_kept``) unless pcl_pool_trim handed them back to the driver in between (``handles_released``).  The handles
    that are kept are idle memory like the pool's blocks: together with those never more than PCL_POOL_GB (a third of the
    device), the rest of the 109 GB goes back to the driver at once."""
    code = ("import numpy as np, sys\n"
            "from physicl_amd import _hip as hip\n"
            "C = 299792458.0\n"
            "d = hip.Device(0)\n"
            "f, t = d.mem_info()\n"
            "BIG = 800_000_000 if t > 250 * 2**30 else 200_000_000\n"
            "d.store_alloc(BIG); d.fill_photons(BIG, 0, C, 1.0, 2.0, 3); d.sync(); d.store_free()\n"
            "held = hip.pool_bytes(); info = hip.pool_info()\n"
            "assert 0 < held <= t // 3 and held == info['idle_blocks'] + info['idle_handles'], (held, t, info)\n"
            "assert info['parked_va'] >= BIG * 17 * 8 and info['vmm_on']\n"      # the big range's addresses are out of circulation
            "f2, _ = d.mem_info(); assert f2 >= f - held - (4 << 30), (f, f2, held)\n"   # what is not kept is the device's again
            "if sys.argv[1] == 'trim':\n"
            "    assert hip.pool_trim() == held and hip.pool_bytes() == 0\n"
            "N = 100_000_000\n"
            "d.store_alloc(N)\n"
            "for g in range(2):\n"
            "    d.fill_photons(N, g * N, C, 1.0, 2.0, 3)\n"
            "    v0 = d.download(hip.V0); E = d.download(hip.E)\n"
            "    assert int((v0 != C).sum()) == 0 and int(((E < 1.0) | (E > 2.0)).sum()) == 0, (g, int((v0 != C).sum()))\n"
            "    assert d.step_counters([])[0] == N\n"
            "d.close(); print('ok')\n")
    out = subprocess.check_output([sys.executable, "-c", code, "trim" if trim else "keep"], cwd=ROOT, timeout=600)
    assert out.decode().strip().endswith("ok")


def test_thirty_stores_of_1e8_photons_one_after_the_other_park_no_address_space():
    """A script that creates and drops its store over and over (one Simulation per parameter set) gets the pooled block back,
    still mapped: nothing is unmapped, so no address range is parked and the idle memory stays within the pool's limit."""
    code = ("from physicl_amd import _hip as hip\n"
            "N = 100_000_000\n"
            "d = hip.Device(0)\n"
            "f, t = d.mem_info()\n"
            "d.store_alloc(N); d.fill_photons(N, 0, 299792458.0, 1.0, 2.0, 3); d.store_free()\n"
            "i0 = hip.pool_info()\n"
            "for k in range(30):\n"
            "    d.store_alloc(N); d.fill_photons(N, k, 299792458.0, 1.0, 2.0, 3)\n"
            "    assert d.step_counters([])[0] == N\n"
            "    d.store_free()\n"
            "i1 = hip.pool_info()\n"
            "assert i1['parked_va'] - i0['parked_va'] <= N * 17 * 8, (i0, i1)\n"
            "assert i1['idle_blocks'] + i1['idle_handles'] <= t // 3 and i1['vmm_on'], i1\n"
            "d.close(); print('ok', i0, i1)\n")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, timeout=900)
    assert out.decode().strip().splitlines()[-1].startswith("ok")


def test_a_range_whose_addresses_could_not_be_kept_is_never_mapped_over():
    """parking a freed range (free its addresses, reserve the same ones again) can fail, and then the addresses are
    back in circulation.  PCL_VMM_TEST_PARK_FAIL makes every parking "fail": the freed ranges are remembered, a fresh
    reservation that overlaps one of them is set aside, and the next store lies somewhere else -- with everything a kernel
    writes into it in place."""
    code = ("import numpy as np\n"
            "from physicl_amd import _hip as hip\n"
            "N, C = 3_000_000, 299792458.0\n"                       # 408 MB: a mapped range of several 64 MB handles under the knobs
            "d = hip.Device(0)\n"
            "old = []\n"
            "for k in range(4):\n"
            "    d.store_alloc(N)\n"
            "    p = d.field_ptr(hip.R0)\n"
            "    assert all(not (p < q + N * 17 * 8 and q < p + N * 17 * 8) for q in old), (hex(p), [hex(q) for q in old])\n"
            "    d.fill_photons(N, k * N, C, 1.0, 2.0, 3)\n"
            "    v0 = d.download(hip.V0); E = d.download(hip.E)\n"
            "    assert int((v0 != C).sum()) == 0 and int(((E < 1.0) | (E > 2.0)).sum()) == 0\n"
            "    old.append(p)\n"
            "    d.store_free()\n"
            "info = hip.pool_info()\n"
            "assert info['vmm_on'] and info['idle_blocks'] == 0, info\n"
            "d.close(); print('ok', info)\n")
    env = dict(os.environ, PCL_VMM_TEST_PARK_FAIL="1", PCL_POOL_GB="0", PCL_VMM_CHUNK_TILES="240", PCL_ALLOC_TRIES="1")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, env=env, timeout=600)
    assert out.decode().strip().splitlines()[-1].startswith("ok")
