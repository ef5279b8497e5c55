"""Does the one-step kernel's rate depend on WHICH allocation the store got?  Same process, pool off: allocate, fill, time
30 single steps (HIP events), free; several times, with and without a spacer allocation in between."""
import os, sys, time
os.environ.setdefault("PCL_POOL_GB", "0")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PROFILES, C_LIT, H_LIT
from physicl_amd import _hip
N = 100_000_000
prof = PROFILES["example"]
d = _hip.Device(0)
sc = lambda k: dict(A=prof["A_kernel"], n=prof["n_kernel"], flags=3, c=C_LIT, h=H_LIT, n_expr=prof["expr"], rng_mode=_hip.RNG_PHILOX, seed=7, step=k)
spacers = []
for trial in range(8):
    d.store_alloc(N)
    p = d.field_ptr(_hip.R0)
    d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 7)
    d.timer_start()
    for k in range(5):
        d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 7)
    fill_ms = d.timer_stop() / 5
    nbytes = (N + 2047) // 2048 * 2048 * 17 * 8
    _hip.check(d.lib.pcl_dev_memset(d.ctx, p, 0, nbytes))
    d.timer_start()
    for k in range(3):
        _hip.check(d.lib.pcl_dev_memset(d.ctx, p, 0, nbytes))
    set_ms = d.timer_stop() / 3
    d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 7)
    d.timer_start()
    for k in range(5):
        d.step_counters([])
    cnt_ms = d.timer_stop() / 5
    for k in range(3):
        d.step_fused(prof["dt"], sc(k), (), lazy=True)
    ms = []
    for rep in range(3):
        d.timer_start()
        for k in range(10):
            d.step_fused(prof["dt"], sc(3 + rep * 10 + k), None, sync=False, lazy=True)
        ms.append(d.timer_stop() / 10)
    print("trial %d: slab at 0x%x  %.4f %.4f %.4f ms/step -> %.3f of peak; memset %.3f ms (%.2f TB/s); fill %.4f ms (%.2f TB/s written), counters %.4f ms (%.2f TB/s read)"
          % (trial, p, ms[0], ms[1], ms[2], 104.0 * N / (min(ms) * 1e-3) / 8e12, set_ms, nbytes / set_ms / 1e9, fill_ms, 104.0 * N / fill_ms / 1e9, cnt_ms, 24.0 * N / cnt_ms / 1e9), flush=True)
    d.store_free()
    if trial % 2 == 1:                     # shift the next allocation
        spacers.append(d.empty(int(3e8 + trial * 1e8)))
d.close()
