"""Where the time of bench.py's mixed leg (tools/sweep_fp32.py, fp64, 16 iterations per launch) goes: wall time of each of its eight
launches (1, 9, 16 x 5, 10 iterations; compaction included) beside the library's own kernel times.  python tools/mixed_leg_split.py [f64|f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip as hip
C, H = 299792458.0, 6.62607015e-34
N = 100_000_000
dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
d = hip.Device(0)
for rep in range(3):
    d.store_alloc(N, dtype)
    sc = dict(A=1e-3, n=1e-3, flags=0, c=C, h=H, rng_mode=hip.RNG_PHILOX, seed=11)
    d.fill_photons(N, 0, C, 2.84e-19, 9.93e-19, 11)
    d.step_mixed_multi(1e-3, 1, ("iso", "delete"), sc, (2e-5, 1e-3), (), 11, 2)
    d.fill_photons(N, 0, C, 2.84e-19, 9.93e-19, 11)
    d.sync()
    d.prof_enable(rep == 2)                       # (the library's per-kernel events cost a little: last repetition only)
    walls, k = [], 0
    t_all = time.perf_counter()
    for n_it in (1, 9, 16, 16, 16, 16, 16, 10):
        t0 = time.perf_counter()
        rows = d.step_mixed_multi(1e-3, n_it, ("iso", "delete"), sc, (2e-5, 1e-3), (), 11, 2 * (k + 1))
        walls.append(round((time.perf_counter() - t0) * 1e3, 3)); k += n_it
    d.sync()
    total = (time.perf_counter() - t_all) * 1e3
    prof = {hip.PROF_NAMES[j]: (d.prof_read(j)["launches"], round(d.prof_read(j)["total_ms"], 3)) for j in (hip.PROF_MULTI, hip.PROF_COMPACT)} if rep == 2 else None
    print(dtype, "total %.2f ms" % total, "per launch", walls, "alive", rows[-1]["N"], "rows", d.last_mixed_rows(), prof, flush=True)
    d.prof_enable(False)
