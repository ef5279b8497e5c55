"""How much does a store gain from more candidate slabs?  One fresh process: a 1e8-photon store with PCL_ALLOC_TRIES candidates
(rates in the order tried), then the one-launch-per-step kernel's rate on the chosen slab."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from physicl_amd import _hip as hip
C, H = 299792458.0, 6.62607015e-34
N = 100_000_000
d = hip.Device(0); d.store_alloc(N)
info = d.alloc_info()
d.fill_photons(N, 0, C, H * C / 700e-9, H * C / 200e-9, 1234)
sc = dict(A=1e-15, n=1e-19, flags=3, c=C, h=H, n_expr="0.000000001 * exp(r0[gid] - 5)", rng_mode=hip.RNG_PHILOX, seed=1234)
for k in range(5):
    d.step_fused(5e-3, dict(sc, step=k), (), sync=True, lazy=True)
d.prof_enable(True)
for k in range(5, 35):
    d.step_fused(5e-3, dict(sc, step=k), (), sync=True, lazy=True)
p = d.prof_read([kid for kid, name in hip.PROF_NAMES.items() if name == "k_fused"][0])
print("candidates", info["candidates_GBps"], "chosen", info["chosen_GBps"], "k_fast avg ms %.4f -> %.3f of 8 TB/s" % (p["avg_ms"], 104.0 * N / (p["avg_ms"] * 1e-3) / 8e12), flush=True)
