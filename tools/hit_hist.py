#!/usr/bin/env python3
"""Distribution of the hits a wave queues per step in the K-step pass, block by block of the driver's command (debug build
of the kernels: PCL_RTC_EXTRA=PCL_HIT_HIST adds one global atomic per wave-step; the run is slower, its counts are not).

    PCL_RTC_EXTRA=PCL_HIT_HIST PCL_MULTI_HIST=1 python tools/hit_hist.py > profiles/r04_hit_histogram.md

A dense pass serves up to 64 queued hits: ceil(hits / 64) passes per wave-step.  What the table shows per block (20 steps,
1e8 photons) for each form: the mean hits, the share of wave-steps that need 0 / 1 / 2 / 3+ passes, lane use of the passes,
and the passes per 128 photons -- the figure the two forms compete on."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from physicl_amd import _hip  # noqa: E402


def main():
    if "PCL_HIT_HIST" not in os.environ.get("PCL_RTC_EXTRA", ""):
        raise SystemExit("run with PCL_RTC_EXTRA=PCL_HIT_HIST PCL_MULTI_HIST=1")
    N, steps, warm = int(float(os.environ.get("PHOTONS", "1e8"))), 20, 5
    p = bench.PROFILES["example"]
    print("# hits queued per wave and step in the K-step pass (%.0e photons, the driver's blocks: %d warm-up steps, then blocks of %d)\n" % (N, warm, steps))
    print("`tools/hit_hist.py`, debug build of the kernels (`PCL_RTC_EXTRA=PCL_HIT_HIST`).  128-photon form: hits of the wave's 128 photons "
          "in a step; 256-photon form: hits of a queue round (all 256 photons when they fit the 128-entry queue, else one 128-photon group).\n")
    print("| form | block | hit fraction | mean hits | 0 passes | 1 pass | 2 passes | 3+ | lanes used in the passes | passes per 128 photons and step |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for form, knobs in (("128", {"PCL_MULTI_NQ2": "0"}), ("256", {"PCL_MULTI_NQ2": "1"})):
        for k, v in knobs.items():
            _hip.set_knob(k, v)
        with _hip.Device(0) as d:
            d.store_alloc(N)
            d.fill_photons(N, 0, bench.C_LIT, bench.H_LIT * bench.C_LIT / 700e-9, bench.H_LIT * bench.C_LIT / 200e-9, 1234)
            sc = dict(A=p["A_kernel"], n=p["n_kernel"], flags=_hip.SCATTER_WAVELENGTH | _hip.SCATTER_VARIABLE_N, c=bench.C_LIT, h=bench.H_LIT,
                      n_expr=p["expr"], rng_mode=_hip.RNG_PHILOX, seed=1234)
            d.step_fused_multi(p["dt"], warm, dict(sc, step=0))
            k = warm
            for blk in range(5):
                rows = d.step_fused_multi(p["dt"], steps, dict(sc, step=k))
                k += steps
                h = d.last_multi_hist().astype(np.float64)
                hits = sum(o["hits"] for o in rows)
                bins = np.arange(129)
                tot = h.sum()
                passes = np.ceil(bins / 64.0)
                passes[128] = 3                                     # (">= 128": at least two full passes and a third)
                n_pass = (h * passes).sum()
                per128 = n_pass / (N / 128.0 * steps)
                print("| %s | %d | %.3f | %.1f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f |" % (
                    form, blk, hits / float(N * steps), (h * bins).sum() / tot, h[0] / tot, h[1:65].sum() / tot, h[65:128].sum() / tot,
                    h[128] / tot, hits / (64.0 * n_pass) if n_pass else 0.0, per128))
        for k in knobs:
            _hip.set_knob(k, None)


if __name__ == "__main__":
    main()
