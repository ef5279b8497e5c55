"""64-bit indexing check at > 2^32 slab elements: 1.2e9 photons (163 GB slab), K-step pass vs single steps from the
same seed: identical per-step counters; ids / positions sampled at the far end of the store."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PROFILES, C_LIT, H_LIT
from physicl_amd import _hip
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_200_000_000
prof = PROFILES["example"]
d = _hip.Device(0)
t0 = time.time(); d.store_alloc(N); print("alloc %.1f s, %.0f GB slab" % (time.time() - t0, N * 17 * 8 / 1e9), flush=True)
sc = lambda k: dict(A=prof["A_kernel"], n=prof["n_kernel"], flags=3, c=C_LIT, h=H_LIT, n_expr=prof["expr"], rng_mode=_hip.RNG_PHILOX, seed=7, step=k)
fill = lambda: d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 7)
fill()
t0 = time.time(); rows = d.step_fused_multi(prof["dt"], 4, sc(0)); t1 = time.time() - t0
multi = [(o["N"], o["hits"], list(o["sign"])) for o in rows]
tail_m = [d.download(_hip.R0 + k, 1000, N - 1000) for k in range(3)]
fill()
single = []
for k in range(4):
    o = d.step_fused(prof["dt"], sc(k), (), lazy=True)
    single.append((o["N"], o["hits"], list(o["sign"])))
tail_s = [d.download(_hip.R0 + k, 1000, N - 1000) for k in range(3)]
assert multi == single, (multi, single)
assert all(np.array_equal(a, b) for a, b in zip(tail_m, tail_s))
print("ok: %d photons, 4 steps, K-step pass %.3f s (%.3g particle-steps/s); rows %s" % (N, t1, 4 * N / t1, multi[-1]))
d.close()
