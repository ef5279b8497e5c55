"""K-sweep of pcl_step_fused_multi on the bench workload: kernel time per launch and per particle-step.

    python tools/bench_multi.py [--photons 1e8] [--dtype f64] [--ks 1,2,4,8,16,32,64]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PROFILES, C_LIT, H_LIT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--photons", type=float, default=1e8)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--profile", default="example")
    ap.add_argument("--ks", default="1,2,4,8,16,32,64")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--plain-p", type=float, default=None,
                    help="no variable-n / wavelength: constant hit probability p for every photon (AOT kernel)")
    ap.add_argument("--expr", default=None, help="override the profile's variable_n_fn (e.g. 1.0E+300: every photon hits)")
    ap.add_argument("--a-kernel", type=float, default=None, help="override the kernel's A (0: no photon hits)")
    args = ap.parse_args()
    from physicl_amd import _hip
    N = int(args.photons)
    prof = dict(PROFILES[args.profile])
    if args.expr is not None:
        prof["expr"] = args.expr
    if args.a_kernel is not None:
        prof["A_kernel"] = args.a_kernel
    dev = _hip.Device(0)
    dev.store_alloc(N, args.dtype)
    fill = lambda: dev.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 1234)
    fill()
    flags = _hip.SCATTER_WAVELENGTH | _hip.SCATTER_VARIABLE_N
    if args.plain_p is not None:                   # pcoll = A * n * |v| dt = p
        flags = 0
        prof.update(A_kernel=args.plain_p / (C_LIT * prof["dt"]), n_kernel=1.0, expr=None)
    sc = lambda k: dict(A=prof["A_kernel"], n=prof["n_kernel"], flags=flags, c=C_LIT, h=H_LIT, n_expr=prof["expr"],
                        rng_mode=_hip.RNG_PHILOX, seed=1234, step=k)
    rows = []
    for K in [int(x) for x in args.ks.split(",")]:
        fill()                                                            # every K starts from the same state
        step = 0
        for _ in range(max(1, 5 // K)):
            dev.step_fused_multi(prof["dt"], K, sc(step)); step += K      # warm-up (hipRTC, clocks): >= 5 steps
        dev.prof_enable(True)
        hits = 0
        for _ in range(args.reps):
            out = dev.step_fused_multi(prof["dt"], K, sc(step)); step += K
            hits += sum(o["hits"] for o in out)
        p = dev.prof_read(_hip.PROF_MULTI)
        dev.prof_enable(False)
        row = {"K": K, "launch_ms": p["avg_ms"], "ms_per_step": p["avg_ms"] / K,
               "particle_steps_per_s": N * K / (p["avg_ms"] * 1e-3), "hit_fraction": hits / (N * K * args.reps)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
