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

# ---- delete / compaction and the mixed K-pass kernel beyond 2^32 slab elements (two slabs: half the photons) -------
N2 = N // 2
d = _hip.Device(0)
d.store_alloc(N2)
print("delete legs: %d photons, 2 x %.0f GB slabs" % (N2, N2 * 17 * 8 / 1e9), flush=True)
fill2 = lambda: d.fill_photons(N2, 0, C_LIT, 1.0, 1.0, 11)
plane = [[1.0 / (1e-3 * 1e-3), np.nan, np.nan]]


def tails():
    n = d.count
    return [d.download(_hip.R0 + k, 1000, n - 1000) for k in range(3)] + [d.download_ids(1000, n - 1000)]


fill2()
t0 = time.time()
per_step = []
for k in range(3):
    o = d.step_fused_delete(1e-3, 1e-3, 1e-3, _hip.RNG_PHILOX, 11, k, plane, lazy=True)
    per_step.append((o["N"], o["removed"], list(o["sign"]), list(o["planes"])))
t1 = time.time() - t0
ta = tails()
fill2()
rows = [(o["N"], o["removed"], list(o["sign"]), list(o["planes"])) for o in d.step_fused_delete_multi(1e-3, 3, 1e-3, 1e-3, 11, 0, plane)]
tb = tails()
assert rows == per_step, (rows, per_step)
assert all(np.array_equal(a, b) for a, b in zip(ta, tb))
assert np.all(np.diff(ta[3]) > 0) and ta[3][-1] <= N2 - 1 and ta[3][-1] > N2 - 1000
print("ok: delete x3 at %d photons: %.3f s one pipeline per loop body; rows %s" % (N2, t1, per_step), flush=True)
sc2 = dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=H_LIT, n_expr=None, rng_mode=_hip.RNG_PHILOX, seed=11, step=0)
fill2()
single = []
for k in range(4):
    if k % 2 == 0:
        o = d.step_fused(1e-3, dict(sc2, step=k), plane, lazy=True)
        single.append((o["N"], o["hits"], list(o["sign"]), list(o["planes"])))
    else:
        o = d.step_fused_delete(1e-3, 1e-3, 1e-3, _hip.RNG_PHILOX, 11, k, plane, lazy=True)
        single.append((o["N"], o["removed"], list(o["sign"]), list(o["planes"])))
ta = tails()
fill2()
t0 = time.time()
rows = d.step_mixed_multi(1e-3, 2, ("iso", "delete"), sc2, (1e-3, 1e-3), plane, 11, 0)
t1 = time.time() - t0
mixed = [(o["N"], o["hits"] if o["phase"] == "iso" else o["removed"], list(o["sign"]), list(o["planes"])) for o in rows]
tb = tails()
assert mixed == single, (mixed, single)
assert all(np.array_equal(a, b) for a, b in zip(ta, tb))
print("ok: mixed [iso, delete] x2 at %d photons in one pass: %.3f s; rows %s" % (N2, t1, mixed))
d.close()
