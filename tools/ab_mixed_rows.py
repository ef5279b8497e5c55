"""k_mixed (two rows of 64 particles per wave and trip) against k_mixed3 (three, velocities in LDS) on BASELINE configs[4]'s loop,
[Newton, ScatterIsotropic(A = n = 1e-3), Newton, ScatterDelete(A n = 2e-8: pcoll = 6e-3)] x 16 per launch, 1e8 photons, same box, alternating:
kernel time of the launches (the library's own events) and what is left alive.   python tools/ab_mixed_rows.py [f64|f32] [launches=6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip as hip
C, H = 299792458.0, 6.62607015e-34
N = 100_000_000
dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = hip.Device(0); d.store_alloc(N, dtype)
for mode in ("0", "1", "0", "1"):
    hip.set_knob("PCL_MIXED_NE3", mode)
    d.fill_photons(N, 0, C, 1.0, 1.0, 1234)
    sc = dict(A=1e-3, n=1e-3, flags=0, c=C, h=H, n_expr=None, rng_mode=hip.RNG_PHILOX, seed=1234, step=1)
    d.prof_enable(True)
    out, step = [], 1
    for b in range(launches):
        d.sync(); t0 = time.perf_counter()
        rows = d.step_mixed_multi(1e-3, 16, ("iso", "delete"), dict(sc, step=step), (2e-5, 1e-3), [], 1234, step)
        d.sync(); el = time.perf_counter() - t0; step += 32
        out.append((round(el * 1e3, 2), rows[0]["N"], round(rows[0]["hits"] / max(1, rows[0]["N"]), 3), d.last_mixed_rows()))
    prof = {hip.PROF_NAMES[k]: round(d.prof_read(k)["total_ms"], 3) for k in (hip.PROF_MULTI, hip.PROF_COMPACT)}
    d.prof_enable(False)
    print("NE3=%s %s (wall ms per launch incl. compaction, alive at its start, hit fraction of its first step, rows)" % (mode, dtype), out, prof, flush=True)
