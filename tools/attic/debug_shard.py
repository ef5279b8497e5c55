"""debug: why do shards filled with id_base = g * 1e8 lose photons after an 8e8 store lived in the same context?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from physicl_amd import _hip as hip
C, H = 299792458.0, 6.62607015e-34
SHARD = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
BIG = int(float(sys.argv[2])) if len(sys.argv) > 2 else 0
K = 8
e_lo, e_hi = H * C / 700e-9, H * C / 200e-9
sc = dict(A=1e-15, n=1e-19, flags=3, c=C, h=H, n_expr="0.000000001 * exp(r0[gid] - 5)", rng_mode=hip.RNG_PHILOX, seed=1234, step=0)
with hip.Device(0) as d:
    if BIG:
        d.store_alloc(BIG)
        d.fill_photons(BIG, 0, C, e_lo, e_hi, 1234)
        rows = d.step_fused_multi(5e-3, K, sc)
        print("big rows", [(o["N"], o["hits"]) for o in rows], flush=True)
        print("big window", d.download(hip.R0, 4, BIG - 4), flush=True)
        d.store_free()
        import time
        time.sleep(float(os.environ.get("DBG_SLEEP", "0")))
    d.store_alloc(SHARD)
    print("alloc_info", d.alloc_info(), "layout", d.layout(), flush=True)
    for g in (0, 1):
        d.fill_photons(SHARD, g * SHARD, C, e_lo, e_hi, 1234)
        v0 = d.download(hip.V0)
        E = d.download(hip.E)
        print("shard", g, "filled: v0 != c:", int((v0 != C).sum()), "E out of range:", int(((E < e_lo) | (E > e_hi)).sum()), "count", d.count, flush=True)
        rows = d.step_fused_multi(5e-3, K, sc)
        print("   rows", [(o["N"], o["hits"]) for o in rows], "work", d.last_multi_work(), flush=True)
        v0 = d.download(hip.V0); r0 = d.download(hip.R0)
        bad = np.flatnonzero(r0 == 0.0)
        print("   after: r0 == 0:", len(bad), "first", bad[:5], "last", bad[-5:] if len(bad) else None, flush=True)
        if len(bad):
            dd = np.diff(bad)
            runs = np.flatnonzero(dd != 1)
            print("   runs of unmoved photons:", len(runs) + 1, "first run", bad[0], "..", bad[runs[0]] if len(runs) else bad[-1], "run starts", bad[np.r_[0, runs + 1]][:10])
