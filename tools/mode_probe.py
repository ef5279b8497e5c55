"""One allocation, many measurement blocks separated by idle gaps: does the one-step kernel's speed mode change without
a new allocation?"""
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
d.store_alloc(N)
d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 7)
for k in range(3):
    d.step_fused(prof["dt"], sc(k), (), lazy=True)
step = 3
for blk in range(16):
    gap = [0.0, 0.01, 0.1, 0.5][blk % 4]
    time.sleep(gap)
    if blk == 8:
        d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 7)
        print("(refilled)")
    d.timer_start()
    for k in range(10):
        d.step_fused(prof["dt"], sc(step), None, sync=False, lazy=True); step += 1
    ms = d.timer_stop() / 10
    print("block %2d after %.2f s idle: %.4f ms/step -> %.3f of peak" % (blk, gap, ms, 104.0 * N / (ms * 1e-3) / 8e12), flush=True)
d.close()
