#!/usr/bin/env python3
"""bench.py -- particle-steps/s and achieved HBM GB/s of the photon time-step hot path on MI355X.

    python bench.py --gpus 1 --steps 64 --warmup 32        (the defaults: warm-up and timed launches all 32 steps long)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the config the metric is quoted on; weak-scaled = configs[3]):
1e8 photons per GPU, r = 0, v = (c,0,0), E power-law between E(700 nm) and E(200 nm)
(generate_photons default sampler, physicl/light.py:112-128), dt = 5e-3 and
ScatterIsotropicStep(n=1e-15, A=1e-19, wavelength_dep_scattering=True, variable_n=True,
variable_n_fn="0.000000001 * exp(r0[gid] - 5)") exactly as examples/variable_n_scattering.ipynb:30,52-56.
One "step" = one pass of the Simulation loop body (physicl/__init__.py:512-516):
UpdateTimeStep -> NewtonianKinematicsStep -> ScatterIsotropicStep -> ScatterSignMeasureStep counters,
then (N > 1) an RCCL all-reduce of the counter vector [N, hits, xp, yp, zp].

Default mode "fused" with --steps-per-launch S > 1 runs S consecutive loop bodies per pass over the store
(pcl_step_fused_multi: photons do not interact, so a photon is loaded once, stepped S times in registers and
stored once -- bit-identical state and per-step counters, 128/S instead of 104 B of HBM traffic per
particle-step, which turns the step from HBM-bound into VALU-bound).  The K timed steps are ceil(K/S) launches.
--steps-per-launch 1 is the one-launch-per-step, HBM-bound path; with N=1 it is also measured in the same run
and reported under "single_step".

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel of the timed region,
timed with HIP event pairs recorded around every launch inside the timed region (pcl_prof_*).
`cpu_baseline` = the oracle's C/OpenMP port of the same step timed on this box's host cores on a
bounded sample of the same photons (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C_LIT = 299792458.0            # str(light.c)          physicl/light.py:14
H_LIT = 6.62607015e-34         # str(light.h).upper()  physicl/light.py:15
HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

PROFILES = {
    # examples/variable_n_scattering.ipynb:30,52-56.  User n=1e-15, A=1e-19; the reference swaps them
    # into the kernel (light.py:287) and in variable_n mode only the kernel's A (= user n) is live.
    "example": dict(expr="0.000000001 * exp(r0[gid] - 5)", A_kernel=1e-15, n_kernel=1e-19, dt=5e-3,
                    arrays_read=("d0", "d1", "d2", "E", "r0"), c_profile=(1, 0.000000001, 5.0)),
    # atmosphere-like regime of examples/presentation_example_2.ipynb:41 (cl_n2), kernel constant chosen so
    # that pcoll spans 0.03..1.8: the branchy partial-hit regime (SURVEY.md 8(d) "tame" profile).
    "tame": dict(expr="2.5E+25 * exp(r2[gid] / 8600.0)", A_kernel=4.08e-56, n_kernel=1.0, dt=1e-5,
                 arrays_read=("d0", "d1", "d2", "E", "r2"), c_profile=None),
}


def algorithmic_bytes_per_particle(profile, h, mode="separate", multi=False):
    """fp64 bytes the dominant kernel must move per particle-step (DESIGN.md 'Kernels').
    separate (k_scatter): reads dr (24) + E (8) + the position components the expression names (8 each),
        writes dv (24, always); a hit additionally reads v_old (24) and writes v' (24): 64 + 48h here.
    fused-eager (k_fused = Newton + scatter + counters): reads r (24) + v (24) + E (8); writes r (24) +
        dr (24) + dv (24); a hit additionally writes v' (24): 128 + 24h.
    fused (PCL_FUSED_LAZY): dr and dv stay implicit (derivable from the v double buffer): 104, hit or miss."""
    if mode == "fused" and multi:   # per LAUNCH of S steps: reads r (24) + v (24) + lam4 (8); writes r (24) + v (24) + v_prev (24)
        return 128.0
    if mode == "fused":      # lazy: reads r (24) + v (24) + E (8); writes r (24) + v (24, double buffer)
        return 104.0
    if mode == "fused-eager":
        return 128 + 24.0 * h
    return 8 * len(PROFILES[profile]["arrays_read"]) + 24 + 48.0 * h


VALU_PEAK_TLANE = 256 * 4 * 16 * 2.4e9 / 1e12    # CUs x SIMDs x lanes/clk x 2.4 GHz = 39.3 T lane-instructions/s


def valu_roofline(v, per_gpu_rate):
    """The K-step kernel is bound by vector-ALU issue, not by HBM: VALU instructions per particle-step (rocprofv3
    SQ_INSTS_VALU x 64 / particle-steps, profiles/) x the measured particle-steps/s of one GPU, against the chip's
    issue peak of one VALU instruction per lane per clock (fp64 FMA included: 78.6 TFLOP/s = 2 x 39.3)."""
    out = dict(v)
    ach = v["valu_insts_per_particle_step"] * per_gpu_rate / 1e12
    out.update({"bound": "valu", "achieved": ach, "peak": VALU_PEAK_TLANE, "unit": "T lane-instr/s",
                "frac": ach / VALU_PEAK_TLANE})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--photons", type=float, default=1e8, help="photons PER GPU (weak scaling)")
    ap.add_argument("--profile", choices=sorted(PROFILES), default="example")
    ap.add_argument("--mode", choices=("fused", "fused-eager", "separate"), default="fused",
                    help="fused: the loop body as ONE kernel with dr/dv left implicit (pcl_step_fused, "
                         "PCL_FUSED_LAZY); fused-eager: one kernel, dr/dv written every step; "
                         "separate: one kernel per Step")
    ap.add_argument("--steps-per-launch", type=int, default=32,
                    help="fused mode: loop bodies per pass over the store (1 = one launch per step, HBM-bound; "
                         "1..64; results are bit-identical for every value)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64",
                    help="f64 = the reference's precision (the headline number); f32 = precision-sweep build")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend for N > 1: nccl = RCCL over xGMI (default); gloo = rehearsal on CPU tensors")
    ap.add_argument("--device", type=int, default=None,
                    help="HIP device index for this rank (default LOCAL_RANK); rehearsals put every rank on device 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-photons", type=float, default=1e7)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU baseline duration")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when it
    # initialises): everything but the final line goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, world))

    from physicl_amd import _hip
    from physicl_amd.dist import CounterComm

    comm = CounterComm.from_env(backend=args.backend, device_index=args.device)   # no-op communicator when world == 1
    N = int(args.photons)
    prof = PROFILES[args.profile]
    flags = _hip.SCATTER_WAVELENGTH | _hip.SCATTER_VARIABLE_N
    e_lo, e_hi = H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9   # E_from_wavelength, light.py:39-43

    dev = _hip.Device(local_rank if args.device is None else args.device)
    dev.store_alloc(N, args.dtype)
    dev.fill_photons(N, rank * N, C_LIT, e_lo, e_hi, args.seed)      # ids are global: shard-independent RNG

    sim_t = 0.0
    totals = None
    S = max(1, min(64, args.steps_per_launch)) if args.mode == "fused" else 1

    fused_sc = lambda k: dict(A=prof["A_kernel"], n=prof["n_kernel"], flags=flags, c=C_LIT, h=H_LIT,
                              n_expr=prof["expr"], rng_mode=_hip.RNG_PHILOX, seed=args.seed, step=k)

    def launch(k):
        """Enqueue step k (no host synchronisation)."""
        nonlocal sim_t
        sim_t += prof["dt"]                                          # UpdateTimeStep   __init__.py:337-343
        if args.mode.startswith("fused"):
            # newton.py:10-16 + light.py:281-331 + light.py:414-431 in one pass over the particles
            dev.step_fused(prof["dt"], fused_sc(k), planes=(), sync=False, lazy=(args.mode == "fused"))
        else:
            dev.step_newton(prof["dt"])                                  # newton.py:10-16
            dev.step_scatter_isotropic(prof["A_kernel"], prof["n_kernel"], flags, C_LIT, H_LIT, prof["expr"],
                                       _hip.RNG_PHILOX, args.seed, k, want_hits=False)   # light.py:281-331

    def collect():
        """Wait for the enqueued step and fetch its local counters [N, hits, xp, yp, zp] (one sync)."""
        if args.mode.startswith("fused"):
            o = dev.step_fused_read(0)
            return np.array([o["N"], o["hits"], o["sign"][0], o["sign"][1], o["sign"][2]], dtype=np.int64)
        cnt = dev.step_counters()                                        # light.py:414-431
        return np.array([cnt[_hip.CNT_N], dev.last_scatter_hits(), cnt[_hip.CNT_XP], cnt[_hip.CNT_YP],
                         cnt[_hip.CNT_ZP]], dtype=np.int64)

    def run_multi(k0, k1):
        """Steps k0..k1-1 as ceil((k1-k0)/S) passes of S loop bodies each; one all-reduce of the S x 5 per-step
        counters per pass."""
        nonlocal totals, sim_t
        hits = 0
        k = k0
        while k < k1:
            ks = min(S, k1 - k)
            for _ in range(ks):
                sim_t += prof["dt"]                                      # UpdateTimeStep   __init__.py:337-343
            rows = dev.step_fused_multi(prof["dt"], ks, fused_sc(k))
            c = np.array([[o["N"], o["hits"], o["sign"][0], o["sign"][1], o["sign"][2]] for o in rows], dtype=np.int64)
            hits += int(c[:, 1].sum())
            totals = comm.allreduce_sum(c.reshape(-1)).reshape(-1, 5)[-1]
            k += ks
        return hits

    def run_steps(k0, k1):
        """Steps k0..k1-1, software-pipelined: step k+1 is enqueued BEFORE the host waits for step k's counters
        (two counter banks in the library), so the GPU never idles on Python; the all-reduce of step k's
        counters (RCCL over xGMI when world > 1) runs while the GPU computes step k+1 -- counters are consumed
        one step behind; the last one is read and reduced inside the timed region."""
        nonlocal totals
        if S > 1:
            return run_multi(k0, k1)
        hits = 0
        if not args.mode.startswith("fused") or os.environ.get("PCL_BENCH_NOPIPE"):
            for k in range(k0, k1):
                launch(k)
                c = collect()
                hits += int(c[1])
                totals = comm.allreduce_sum(c)
            return hits
        if k1 > k0:
            launch(k0)
        for k in range(k0 + 1, k1):
            launch(k)
            c = collect()                      # counters of step k-1
            hits += int(c[1])
            totals = comm.allreduce_sum(c)
        if k1 > k0:
            c = collect()
            hits += int(c[1])
            totals = comm.allreduce_sum(c)
        return hits

    dev.prof_enable(True)
    run_steps(0, args.warmup)
    warm = {name: dev.prof_read(kid) for kid, name in _hip.PROF_NAMES.items()}   # rocprofv3 --stats averages these in too

    dev.prof_enable(True)
    comm.barrier()
    dev.sync()
    comm.device_synchronize()
    t0 = time.perf_counter()
    hits_local = run_steps(args.warmup, args.warmup + args.steps)
    dev.sync()
    comm.device_synchronize()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = comm.allreduce_max(elapsed)

    kern = {name: dev.prof_read(kid) for kid, name in _hip.PROF_NAMES.items()}
    dev.prof_enable(False)
    h_mean = hits_local / float(N * args.steps)
    bpp = algorithmic_bytes_per_particle(args.profile, h_mean, args.mode, S > 1) * (0.5 if args.dtype == "f32" else 1.0)
    dominant = ("k_multi" if S > 1 else "k_fused") if args.mode.startswith("fused") else "k_scatter"
    sc = kern[dominant]
    achieved = N * bpp / (sc["avg_ms"] * 1e-3) / 1e9 if sc["launches"] else 0.0
    # what the same steps would have to move one launch per step (104 B per particle-step) over the time they took:
    # > HBM peak is possible for the K-step pass precisely because it does not move those bytes
    eff_steps = args.steps / max(1, sc["launches"]) if S > 1 else 1.0
    effective = N * 104.0 * (0.5 if args.dtype == "f32" else 1.0) * eff_steps / (sc["avg_ms"] * 1e-3) / 1e9 \
        if (S > 1 and sc["launches"]) else None

    single = None
    if S > 1 and world == 1:
        # the one-launch-per-step path on the same store, same run: the HBM-bound kernel's own roofline line
        S_keep, S = S, 1
        totals_main = totals
        k_next = args.warmup + args.steps
        run_steps(k_next, k_next + 3)
        dev.prof_enable(True)
        dev.sync()
        t1 = time.perf_counter()
        h1 = run_steps(k_next + 3, k_next + 3 + args.steps)
        dev.sync()
        el1 = time.perf_counter() - t1
        k1 = dev.prof_read(_hip.PROF_FUSED)
        dev.prof_enable(False)
        S, totals = S_keep, totals_main
        b1 = 104.0 * (0.5 if args.dtype == "f32" else 1.0)
        a1 = N * b1 / (k1["avg_ms"] * 1e-3) / 1e9 if k1["launches"] else 0.0
        single = {"value": N * args.steps / el1, "unit": "particle-steps/s", "ms_per_step": el1 / args.steps * 1e3,
                  "steps": args.steps,
                  "roofline": {"bound": "hbm", "kernel": "k_fused (one launch per step, dr/dv implicit)", "achieved": a1,
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a1 / HBM_PEAK_GBPS,
                               "algorithmic_bytes_per_particle": b1, "avg_launch_ms": k1["avg_ms"],
                               "launches": k1["launches"], "hit_fraction": h1 / float(N * args.steps)}}

    out = None
    if rank == 0:
        total_particles = N * world
        value = total_particles * args.steps / elapsed
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            t = json.load(open(tfile)).get("%s:%s%s:%d" % (args.profile, args.mode, "-f32" if args.dtype == "f32" else "", N))
            valu = None
            traffic = t.get(dominant + "_bytes_per_launch") if t else None
            valu = t.get(dominant + "_valu") if t else None
            if single is not None and t:
                single["roofline"]["traffic"] = t.get("k_fused_bytes_per_launch")
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]%s: %.0e photons/GPU, variable-n + wavelength isotropic "
                                   "scatter (variable_n_scattering example); step = UpdateTime + Newton + "
                                   "ScatterIsotropic (Philox) + sign counters%s"
                                   % ("/[3] weak-scaled" if world > 1 else "", N,
                                      " + %s all-reduce of the %s int64 counters" % ("RCCL" if comm.backend == "nccl" else "gloo",
                                                                                   "%d x 5 per-launch" % S if S > 1 else "5") if world > 1 else ""),
                       "photons_per_gpu": N, "profile": args.profile, "mode": args.mode, "steps_per_launch": S,
                       "variable_n_fn": prof["expr"], "dt": prof["dt"],
                       "rng": "philox4x32-10 keyed by global photon id", "parallelism": "index-sharded x%d" % world},
            "roofline": {"bound": "hbm",
                         "kernel": (("k_multi: %d x (Newton + ScatterIsotropic + counters) per pass over the store (hipRTC "
                                     "variable-n), dr/dv implicit; VALU-bound by construction -- see 'valu' and single_step" % S)
                                    if S > 1 else
                                    "k_fused: Newton + ScatterIsotropic + counters in one pass (hipRTC variable-n)%s"
                                    % (", dr/dv implicit" if args.mode == "fused" else "")
                                    if args.mode.startswith("fused") else
                                    "k_scatter: ScatterIsotropicStep kernel + write-back (hipRTC variable-n)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "algorithmic_bytes_per_particle": bpp, "hit_fraction": h_mean,
                         "avg_launch_ms": sc["avg_ms"], "launches": sc["launches"], "steps_per_launch": S,
                         "effective_GBps_at_104B_per_step": effective,
                         "warmup_launches": warm[dominant]["launches"], "warmup_avg_launch_ms": warm[dominant]["avg_ms"],
                         "avg_launch_ms_incl_warmup": ((sc["avg_ms"] * sc["launches"] + warm[dominant]["avg_ms"] * warm[dominant]["launches"])
                                                       / max(1, sc["launches"] + warm[dominant]["launches"])),
                         "valu": valu_roofline(valu, value / world) if valu else None},
            # north_star target (>= 60 % of the HBM roofline on the photon-scatter step at 1e8 photons): carried by the
            # one-launch-per-step kernel, measured in this same run (single_step); the K-step pass trades those bytes away
            "hbm_target": ({"kernel": "k_fused, one launch per step", "frac": single["roofline"]["frac"], "target": 0.6,
                            "met": single["roofline"]["frac"] >= 0.6} if single is not None else None),
            "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in kern.items() if v["launches"]},
            "kernels_GBps": {
                "k_newton": N * 96 / (kern["k_newton"]["avg_ms"] * 1e-3) / 1e9 if kern["k_newton"]["launches"] else None,
                "k_counters": N * 24 / (kern["k_counters"]["avg_ms"] * 1e-3) / 1e9 if kern["k_counters"]["launches"] else None,
            },
            "counters_last_step": {"N": int(totals[0]), "hits": int(totals[1]), "xp": int(totals[2]),
                                   "yp": int(totals[3]), "zp": int(totals[4])},
            "device": dev.info()["name"],
        }
        if single is not None:
            out["single_step"] = single
        if world == 1 and not args.no_cpu_baseline and args.dtype == "f64":
            out["cpu_baseline"] = cpu_baseline(dev, args, prof)
            out["cpu_baseline_python"] = cpu_baseline_python(dev, args, prof)
            out["cpu_baseline_numpy"] = cpu_baseline_numpy(dev, args, prof)

    dev.store_free()
    dev.close()
    comm.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


def cpu_baseline(dev, args, prof):
    """The oracle's C/OpenMP port of the same step (newton + fused scatter + sign counters) on the first
    `cpu_photons` photons of the SAME initial workload, all host cores.  Checker code timed as a
    baseline: never part of the GPU path."""
    from oracle import c_oracle as co
    if prof["c_profile"] is None:
        return {"value": None, "unit": "particle-steps/s", "cores": 0, "kind": "port",
                "sample": "C port implements the example profile only"}
    n = int(min(args.cpu_photons, args.photons))
    # same photons as the GPU run: regenerate the initial state of ids [0, n) on the device and download it
    dev.fill_photons(n, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, args.seed)
    st = {g: [dev.download(f, n) for f in fids] for g, fids in
          (("r", (0, 1, 2)), ("v", (3, 4, 5)), ("dr", (6, 7, 8)), ("dv", (9, 10, 11)))}
    st["E"] = dev.download(12, n)
    co.set_threads(co.usable_cores())          # the box's CPU share, not every core of the host
    cores = co.threads()
    profile, pk, poff = prof["c_profile"]

    def step(k):
        co.newton(st, prof["dt"])
        co.scatter_isotropic(st, prof["A_kernel"], prof["n_kernel"], C_LIT, H_LIT, 1, profile, pk, poff, args.seed, k,
                             ids=None, id_base=0)
        co.counters(st)

    step(0)                                   # warm-up + calibration
    t0 = time.perf_counter()
    step(1)
    one = time.perf_counter() - t0
    steps = int(max(2, min(2000, args.cpu_seconds / max(one, 1e-4))))
    t0 = time.perf_counter()
    for k in range(2, 2 + steps):
        step(k)
    el = time.perf_counter() - t0
    return {"value": n * steps / el, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": "%d photons x %d steps of the same workload (oracle/c/physicl_oracle.c, OpenMP, %d threads, "
                      "%.1f s)" % (n, steps, cores, el)}


def cpu_baseline_numpy(dev, args, prof):
    """Third CPU figure: the numpy-vectorised oracle (oracle/physicl_oracle.py) on the first 1e6 photons, 1 core
    (BASELINE.md section 4, item 2)."""
    from oracle import physicl_oracle as orc
    n = int(min(1_000_000, args.photons))
    dev.fill_photons(n, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, args.seed)
    st = {g: [dev.download(f, n) for f in fids] for g, fids in
          (("r", (0, 1, 2)), ("v", (3, 4, 5)), ("dr", (6, 7, 8)), ("dv", (9, 10, 11)))}
    st["E"], st["id"] = dev.download(12, n), np.arange(n, dtype=np.int64)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 4.0:
        orc.step_newton(st, prof["dt"])
        orc.step_scatter_isotropic(st, orc.philox_draws(args.seed, steps, st["id"]), prof["A_kernel"], prof["n_kernel"],
                                   C_LIT, h=H_LIT, use_E=True, n_expr=prof["expr"])
        orc.sign_counts(st["v"])
        steps += 1
    el = time.perf_counter() - t0
    return {"value": n * steps / el, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": "%d photons x %d steps, numpy-vectorised oracle, %.1f s" % (n, steps, el)}


def cpu_baseline_python(dev, args, prof):
    """Second CPU figure, for scale: the reference-SHAPED path (one Python object per photon, a Python loop
    per step, oracle/pyloop.py) on the first 1e4 photons of the same workload, 1 core -- the cost model of
    the reference's own CPU path (BASELINE.md section 2 measured 1.5e4..2.4e4 particle-steps/s for it)."""
    from oracle import pyloop
    if prof["c_profile"] is None:
        return None
    n = int(min(10000, args.photons))
    dev.fill_photons(n, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, args.seed)
    E = dev.download(12, n)
    value, steps, el = pyloop.time_steps(E, prof["dt"], prof["A_kernel"], prof["n_kernel"], True, prof["c_profile"], 3.0)
    return {"value": value, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": "%d photons x %d steps, per-object Python loops (oracle/pyloop.py), %.1f s" % (n, steps, el)}


if __name__ == "__main__":
    main()
