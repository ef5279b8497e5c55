"""One hit queue per workgroup (PCL_MULTI_POOL=1) against the per-wave queues (=0: 256 / 192 photons per wave as the library picks)
on the bench workload: ms per step by block of K steps as the hit fraction decays, same box, alternating.
    python tools/ab_pool.py [K=20] [blocks=12]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip as hip
C, H = 299792458.0, 6.62607015e-34
N = 100_000_000
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
expr = sys.argv[3] if len(sys.argv) > 3 else "0.000000001 * exp(r0[gid] - 5)"
d = hip.Device(0); d.store_alloc(N)
for mode in ("0", "1", "0", "1"):
    hip.set_knob("PCL_MULTI_POOL", mode)
    d.fill_photons(N, 0, C, H * C / 700e-9, H * C / 200e-9, 1234)
    sc = dict(A=1e-15, n=1e-19, flags=3, c=C, h=H, n_expr=expr, rng_mode=hip.RNG_PHILOX, seed=1234)
    d.step_fused_multi(5e-3, 5, dict(sc, step=0)); k = 5
    out = []
    for b in range(blocks):
        d.sync(); t0 = time.perf_counter()
        rows = d.step_fused_multi(5e-3, K, dict(sc, step=k)); d.sync()
        el = time.perf_counter() - t0; k += K
        w = d.last_multi_work()
        out.append((round(el / K * 1e3, 4), round(sum(o["hits"] for o in rows) / N / K, 3), w[2], round(w[0] / max(1, w[1]), 3)))
    print("POOL=%s (ms per step, hit fraction, photons per wave, dense passes per wave-step)" % mode, out, flush=True)
