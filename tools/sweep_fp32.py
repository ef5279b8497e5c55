#!/usr/bin/env python3
"""BASELINE.json configs[4]: mixed Newton + light steps, fp32 vs fp64, same photons, same Philox stream.
Step list per iteration: [Newton, ScatterIsotropic(base), Newton, ScatterDelete]  (SURVEY.md 8(d), config 5).
Reports, after K in {1, 10, 100} iterations: decision mismatches (hits, deletes), max/median relative
position error of photons with identical histories, and the per-step time of both precisions.

    python tools/sweep_fp32.py --photons 1e8
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip  # noqa: E402

C_LIT, H_LIT = 299792458.0, 6.62607015e-34


def run(dev, N, dtype, K, sample, per_launch):
    """per_launch > 1: that many whole iterations per pass over the store and one compaction (pcl_step_mixed_multi);
    1: one launch per light step (pcl_step_fused + pcl_step_fused_delete, dr/dv implicit).  Same photons, same rows."""
    dev.store_alloc(N, dtype)
    sc = dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=H_LIT, rng_mode=_hip.RNG_PHILOX, seed=11)
    # one untimed iteration first: the store's second slab (the compaction's destination) is allocated on first use, and
    # after a change of precision that is a multi-GB hipMalloc (0.19 s in some runs) -- set-up, not the path; then the
    # photons are created again
    dev.fill_photons(N, 0, C_LIT, 2.84e-19, 9.93e-19, 11)
    dev.step_mixed_multi(1e-3, 1, ("iso", "delete"), sc, (2e-5, 1e-3), (), 11, 2)
    dev.fill_photons(N, 0, C_LIT, 2.84e-19, 9.93e-19, 11)
    out = {}
    dev.sync()
    t0 = time.perf_counter()
    hits = deleted = work = 0
    alive = N
    k = 0
    for stop in (1, 10, 100, K):
        while k < min(stop, K):
            n_it = min(per_launch, min(stop, K) - k)
            if per_launch > 1:
                rows = dev.step_mixed_multi(1e-3, n_it, ("iso", "delete"), sc, (2e-5, 1e-3), (), 11, 2 * (k + 1))
                hits += sum(o["hits"] for o in rows if o["phase"] == "iso")
                deleted += sum(o["removed"] for o in rows if o["phase"] == "delete")
                for o in rows:                       # a particle-step = one photon alive at the start of a Newton + light step
                    work += alive
                    alive = o["N"]
            else:
                o = dev.step_fused(1e-3, dict(sc, step=2 * (k + 1)), (), lazy=True)
                d = dev.step_fused_delete(1e-3, 2e-5, 1e-3, _hip.RNG_PHILOX, 11, 2 * (k + 1) + 1, None, lazy=True)
                hits += o["hits"]
                deleted += d["removed"]
                work += 2 * alive
                alive = d["N"]
            k += n_it
        if k in (1, 10, 100) and k not in out:
            dev.sync()
            el = time.perf_counter() - t0
            m = min(sample, alive)
            out[k] = dict(hits=hits, deleted=deleted, alive=alive, seconds=el, work=work)
            if m > 0:
                out[k].update(ids=dev.download_ids(m),
                              r=np.stack([dev.download(_hip.R0 + j, m) for j in range(3)], 1).astype(np.float64),
                              v=np.stack([dev.download(_hip.V0 + j, m) for j in range(3)], 1).astype(np.float64))
            t0 = time.perf_counter() - el          # do not count the download
    return out


def sweep(dev, N, steps, sample, per_launch, timing_only=False):
    """Both precisions on ``dev``; the report of main() as a dict.  ``timing_only``: seconds of the whole run per
    precision, no state comparison (sample may be 0)."""
    r64 = run(dev, N, "f64", steps, sample, per_launch)
    r32 = run(dev, N, "f32", steps, sample, per_launch)
    dev.store_free()
    last = max(r64)
    if timing_only:
        return {"seconds_f64": r64[last]["seconds"], "seconds_f32": r32[last]["seconds"]}
    report = {"workload": "configs[4]: [Newton, ScatterIsotropic(A=n=1e-3), Newton, ScatterDelete(pcoll=6e-3)] x K, "
                          "%d photons, fp32 vs fp64, same Philox stream" % N,
              "iterations_per_launch": per_launch, "checkpoints": {},
              "particle_steps_f64": r64[last]["work"], "particle_steps_f32": r32[last]["work"]}
    tol_v = 4 * 4 * float(np.spacing(np.float32(C_LIT)))
    for k in sorted(r64):
        a64, a32 = r64[k], r32[k]
        common, i64, i32 = np.intersect1d(a64["ids"], a32["ids"], return_indices=True)
        same = np.max(np.abs(a64["v"][i64] - a32["v"][i32]), axis=1) <= tol_v
        x, y = a64["r"][i64][same], a32["r"][i32][same]
        rel = np.linalg.norm(x - y, axis=1) / np.linalg.norm(x, axis=1)
        report["checkpoints"][k] = {
            "hits_f64": a64["hits"], "hits_f32": a32["hits"], "deleted_f64": a64["deleted"], "deleted_f32": a32["deleted"],
            "decision_mismatch_rate": (abs(a64["hits"] - a32["hits"]) + abs(a64["deleted"] - a32["deleted"])) / (2.0 * k * N),
            "compared": int(len(common)), "identical_history_fraction": float(same.mean()),
            # photons whose velocities agree can still have taken one different decision in between (both are
            # re-scattered later from the same stream), hence the tail: quote quantiles, not only the max
            "median_rel_err_r": float(np.median(rel)), "p99_rel_err_r": float(np.quantile(rel, 0.99)),
            "p9999_rel_err_r": float(np.quantile(rel, 0.9999)), "max_rel_err_r": float(rel.max()),
            "fraction_above_1e-4": float((rel > 1e-4).mean()),
            "seconds_f64": a64["seconds"], "seconds_f32": a32["seconds"]}
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--photons", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--sample", type=int, default=2_000_000, help="photons (lowest ids alive) compared per checkpoint")
    ap.add_argument("--iterations-per-launch", type=int, default=16,
                    help="whole [Newton, ScatterIsotropic, Newton, ScatterDelete] iterations per pass over the store "
                         "(pcl_step_mixed_multi; 1 = one launch per light step); results are identical for every value")
    a = ap.parse_args()
    dev = _hip.Device(0)
    print(json.dumps(sweep(dev, int(a.photons), a.steps, a.sample, a.iterations_per_launch)))
    dev.close()


if __name__ == "__main__":
    main()
