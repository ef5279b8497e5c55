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
