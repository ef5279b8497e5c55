"""Several stores allocated side by side in one process: fill rate (the write sweep that tells the two speed modes apart)
of each, twice -- is there a fast one among simultaneous allocations?"""
import os, sys, time
os.environ.setdefault("PCL_POOL_GB", "0")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip
N = 100_000_000
C, H = 299792458.0, 6.62607015e-34
for rnd in range(2):
    devs = []
    for k in range(6):
        d = _hip.Device(0)
        d.store_alloc(N)
        devs.append(d)
    out = []
    for d in devs:
        d.fill_photons(N, 0, C, 1.0, 2.0, 7)
        d.timer_start()
        for k in range(4):
            d.fill_photons(N, 0, C, 1.0, 2.0, 7)
        f = 104.0 * N / (d.timer_stop() / 4) / 1e9
        p = d.field_ptr(_hip.R0)
        nbytes = (N + 2047) // 2048 * 2048 * 17 * 8
        _hip.check(d.lib.pcl_dev_memset(d.ctx, p, 0, nbytes))
        d.timer_start()
        for k in range(3):
            _hip.check(d.lib.pcl_dev_memset(d.ctx, p, 0, nbytes))
        m = nbytes / (d.timer_stop() / 3) / 1e9
        d.timer_start()
        for k in range(3):
            _hip.check(d.lib.pcl_dev_memset(d.ctx, p, 0, nbytes // 16))
        m16 = nbytes / 16 / (d.timer_stop() / 3) / 1e9
        out.append("%.2f/%.2f/%.2f" % (f, m, m16))
    print("round %d: fill / memset / memset of the first 1/16, TB/s, of 6 simultaneous stores: %s" % (rnd, " ".join(out)), flush=True)
    for d in devs:
        d.close()
