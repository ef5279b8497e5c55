#!/usr/bin/env python3
"""bench.py -- particle-steps/s and achieved HBM GB/s of the photon time-step hot path on MI355X.

    python bench.py                                        (N=1; 64 steps per block, 5 blocks, 32 warm-up steps)
    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (same thing, ranks started by torchrun)

Workload (BASELINE.json configs[2], the config the metric is quoted on; weak-scaled = configs[3]):
1e8 photons per GPU, r = 0, v = (c,0,0), E power-law between E(700 nm) and E(200 nm)
(generate_photons default sampler, physicl/light.py:112-128), dt = 5e-3 and
ScatterIsotropicStep(n=1e-15, A=1e-19, wavelength_dep_scattering=True, variable_n=True,
variable_n_fn="0.000000001 * exp(r0[gid] - 5)") exactly as examples/variable_n_scattering.ipynb:30,52-56.
One "step" = one pass of the Simulation loop body (physicl/__init__.py:512-516):
UpdateTimeStep -> NewtonianKinematicsStep -> ScatterIsotropicStep -> ScatterSignMeasureStep counters,
then (N > 1) an RCCL all-reduce of the counter vector [N, hits, xp, yp, zp].

Timing: W warm-up steps, then the block of exactly K steps is timed R times (--repeats, default 5), each block
bracketed by barrier + device synchronise on both sides and reduced with MAX over ranks; `value` uses the MEDIAN
block (every block's time is in `repeat_ms_per_step`).  The simulation simply keeps running from block to block.

Default mode "fused" with --steps-per-launch S > 1 runs S consecutive loop bodies per pass over the store
(pcl_step_fused_multi: photons do not interact, so a photon is loaded once, stepped S times in registers and
stored once -- bit-identical state and per-step counters, 128/S instead of 104 B of HBM traffic per
particle-step, which turns the step from HBM-bound into VALU-bound).  The K timed steps are ceil(K/S) launches.
--steps-per-launch 1 is the one-launch-per-step, HBM-bound path; with N=1 it is also measured in the same run
and reported under "single_step", and the delete/compaction path (BASELINE configs[1](ii)) under "delete".

Prints ONE JSON line on rank 0.  Every number in `roofline`, `single_step`, `delete`, `iso_1e7`, `mixed` and `api` is
measured in THIS run (wall clock, and HIP event pairs recorded on the library's stream around every launch,
pcl_prof_*); PMC-counter figures from committed rocprofv3 profiles appear only under `static_profile`, labelled with
their source.  `roofline` refers to the MEDIAN block, the one `value` is taken from.  With N=1 the other BASELINE
configurations ride along: `delete` = configs[1](ii) at 1e7 and 1e8 (one call per loop body on the store's alive mask,
and K bodies per launch), `iso_1e7` = configs[1](i), `mixed` = configs[4] in its 1-GPU form (fp64 and fp32, with the
fp32-vs-fp64 error figures), `api` = configs[2]'s 500 passes through physicl_amd.Simulation with the constructor exactly
as a script written against the reference calls it.  `--gpus N --dry-run` brings the ranks and the collective up and
stops.
`cpu_baseline` = the oracle's C/OpenMP port of the same step timed on this box's host cores on a
bounded sample of the same photons (rank 0, N=1 only).
"""
import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before anything can initialise HIP (RCCL needs dmabuf IPC)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C_LIT = 299792458.0            # str(light.c)          physicl/light.py:14
H_LIT = 6.62607015e-34         # str(light.h).upper()  physicl/light.py:15
HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; the data sheet's engine clock (MI355X_MICROARCH.md) -- only where no launch measured its own


def valu_peak(clock_ghz):
    """The ceiling of a kernel bound by VALU issue: SIMD-cycles per second at the clock the chip HELD under the launch
    (measured in the kernel: pcl_store_last_multi_clock / pcl_store_ahead_clock), not at the data sheet's 2.4 GHz.  What an
    instruction costs of those cycles is a property of its class (2 / 4 / 8 / 16: tools/valu_issue_probe.hip,
    profiles/r05_valu_issue_probe.txt; tools/isa_count.py prices a kernel's instruction mix)."""
    return N_SIMD * (clock_ghz if clock_ghz and clock_ghz > 0 else CLOCK_GHZ) * 1e9

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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--repeats", type=int, default=5, help="how many times the block of --steps steps is timed (median reported)")
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
                    help="collective backend for N > 1: nccl = RCCL over xGMI (default; failing to bring it up is an "
                         "error); gloo = rehearsal on CPU tensors, must be asked for")
    ap.add_argument("--device", type=int, default=None,
                    help="HIP device index for EVERY rank (default LOCAL_RANK); rehearsals put all ranks on device 0")
    ap.add_argument("--dry-run", action="store_true",
                    help="bring up the ranks, the collective backend and its start-up all-reduce, report every rank's "
                         "device and stop: nothing is allocated, nothing is timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the single_step, delete and api legs (N=1)")
    ap.add_argument("--cpu-photons", type=float, default=1e8)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target duration of the OpenMP CPU baseline")
    ap.add_argument("--delete-photons", type=str, default="1e7,1e8", help="sizes of the delete leg (comma separated)")
    ap.add_argument("--api-steps", type=int, default=0,
                    help="passes of the api leg's simulations (0: 500 -- BASELINE configs[2]'s run length -- at >= 1e7 photons, "
                         "else --steps)")
    ap.add_argument("--iso-photons", type=float, default=1e7, help="size of the iso_1e7 leg (BASELINE configs[1](i))")
    ap.add_argument("--mixed-photons", type=float, default=1e8, help="size of the mixed leg (BASELINE configs[4], 1-GPU form)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched directly: start the ranks ourselves.  Nothing above has touched the GPU (no torch, no
        # libphysicl_hip), the children are fresh interpreters, and this process only waits for them.
        from physicl_amd.launch import spawn_ranks
        rc, out = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        lines = [ln for ln in out.splitlines() if ln.strip()]
        if rc == 0 and len(lines) != 1:
            sys.stderr.write("bench.py: rank 0 printed %d lines instead of one JSON line\n" % len(lines))
            rc = 1
        if rc == 0:
            sys.stdout.write(lines[0] + "\n")
            sys.stdout.flush()
        else:
            sys.stderr.write("bench.py: a rank failed (exit code %d); no result line\n" % rc)
        sys.exit(rc)
    run_rank(args)


class Bench:
    """One rank's store, step functions and timing helpers."""

    def __init__(self, args):
        from physicl_amd import _hip
        from physicl_amd.dist import CounterComm
        self.args, self.hip = args, _hip
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        self.comm = CounterComm.from_env(backend=args.backend, device_index=args.device)   # no-op when world == 1
        self.N = int(args.photons)
        self.prof = PROFILES[args.profile]
        self.flags = _hip.SCATTER_WAVELENGTH | _hip.SCATTER_VARIABLE_N
        self.e_lo, self.e_hi = H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9   # E_from_wavelength, light.py:39-43
        self.dev_index = self.local_rank if args.device is None else args.device
        self.dev = _hip.Device(self.dev_index)
        self.S = max(1, min(64, args.steps_per_launch)) if args.mode == "fused" else 1
        self.sim_t = 0.0
        self.totals = None
        self.local_block_s = []        # this rank's own wall time of every timed block (the line reports min / max over ranks)
        self.work_log = []             # K-step launches: what each kernel tallied of its own work (pcl_store_last_multi_work)
        self.rows_log = {}             # launch index k -> this rank's row [N, hits, xp, yp, zp] of step k of the MAIN run (checked against the CPU port)
        self.log_rows = True

    def fill(self):
        self.dev.store_alloc(self.N, self.args.dtype)
        self.dev.fill_photons(self.N, self.rank * self.N, C_LIT, self.e_lo, self.e_hi, self.args.seed)   # global ids

    def sc(self, k):
        p, a = self.prof, self.args
        return dict(A=p["A_kernel"], n=p["n_kernel"], flags=self.flags, c=C_LIT, h=H_LIT, n_expr=p["expr"],
                    rng_mode=self.hip.RNG_PHILOX, seed=a.seed, step=k)

    def launch(self, k):
        """Enqueue step k (no host synchronisation)."""
        a, dev, p = self.args, self.dev, self.prof
        self.sim_t += p["dt"]                                          # UpdateTimeStep   __init__.py:337-343
        if a.mode.startswith("fused"):
            # newton.py:10-16 + light.py:281-331 + light.py:414-431 in one pass over the particles
            dev.step_fused(p["dt"], self.sc(k), planes=(), sync=False, lazy=(a.mode == "fused"))
        else:
            dev.step_newton(p["dt"])                                  # newton.py:10-16
            dev.step_scatter_isotropic(p["A_kernel"], p["n_kernel"], self.flags, C_LIT, H_LIT, p["expr"],
                                       self.hip.RNG_PHILOX, a.seed, k, want_hits=False)   # light.py:281-331

    def collect(self):
        """Wait for the enqueued step and fetch its local counters [N, hits, xp, yp, zp] (one sync)."""
        dev, hip = self.dev, self.hip
        if self.args.mode.startswith("fused"):
            o = dev.step_fused_read(0)
            return np.array([o["N"], o["hits"], o["sign"][0], o["sign"][1], o["sign"][2]], dtype=np.int64)
        cnt = dev.step_counters()                                        # light.py:414-431
        return np.array([cnt[hip.CNT_N], dev.last_scatter_hits(), cnt[hip.CNT_XP], cnt[hip.CNT_YP],
                         cnt[hip.CNT_ZP]], dtype=np.int64)

    def run_multi(self, k0, k1, S):
        """Steps k0..k1-1 as ceil((k1-k0)/S) passes of S loop bodies each; one all-reduce of the S x 5 per-step
        counters per pass."""
        hits, k = 0, k0
        while k < k1:
            ks = min(S, k1 - k)
            self.sim_t += ks * self.prof["dt"]                           # UpdateTimeStep   __init__.py:337-343
            rows = self.dev.step_fused_multi(self.prof["dt"], ks, self.sc(k))
            c = np.array([[o["N"], o["hits"], o["sign"][0], o["sign"][1], o["sign"][2]] for o in rows], dtype=np.int64)
            hits += int(c[:, 1].sum())
            if self.log_rows:
                for j in range(ks):
                    self.rows_log.setdefault(k + j, c[j].tolist())
            # steps, hits, dense passes, wave-steps, photons per wave, wave-steps on exp's shortcut (-1: no probe), GHz held under the launch
            self.work_log.append((ks, int(c[:, 1].sum())) + self.dev.last_multi_work() + (round(self.dev.last_multi_clock(), 4),))
            self.totals = self.comm.allreduce_sum(c.reshape(-1)).reshape(-1, 5)[-1]
            k += ks
        return hits

    def run_steps(self, k0, k1, S=None):
        """Steps k0..k1-1.  S == 1 in fused mode is software-pipelined: step k+1 is enqueued BEFORE the host waits
        for step k's counters (two counter banks in the library), so the GPU never idles on Python; the all-reduce
        of step k's counters (RCCL over xGMI when world > 1) runs while the GPU computes step k+1 -- counters are
        consumed one step behind; the last one is read and reduced inside the timed region."""
        S = self.S if S is None else S
        if S > 1:
            return self.run_multi(k0, k1, S)
        hits = 0
        if not self.args.mode.startswith("fused") or os.environ.get("PCL_BENCH_NOPIPE"):
            for k in range(k0, k1):
                self.launch(k)
                c = self.collect()
                hits += int(c[1])
                if self.log_rows:
                    self.rows_log.setdefault(k, c.tolist())
                self.totals = self.comm.allreduce_sum(c)
            return hits
        if k1 > k0:
            self.launch(k0)
        for k in range(k0 + 1, k1):
            self.launch(k)
            c = self.collect()                      # counters of step k-1
            hits += int(c[1])
            if self.log_rows:
                self.rows_log.setdefault(k - 1, c.tolist())
            self.totals = self.comm.allreduce_sum(c)
        if k1 > k0:
            c = self.collect()
            hits += int(c[1])
            if self.log_rows:
                self.rows_log.setdefault(k1 - 1, c.tolist())
            self.totals = self.comm.allreduce_sum(c)
        return hits

    def timed_blocks(self, k_start, steps, repeats, S=None):
        """``repeats`` consecutive blocks of ``steps`` steps, each bracketed by barrier + synchronise on both sides.
        Returns per-block wall seconds (MAX over ranks), local hit counts and per-kernel HIP-event samples."""
        comm, dev = self.comm, self.dev
        el, hits, kern = [], [], []
        self.block_work = []
        k = k_start
        for _ in range(repeats):
            w0 = len(self.work_log)
            dev.prof_enable(True)                   # clears the samples; synchronises (outside the timed region)
            comm.barrier()
            dev.sync()
            comm.device_synchronize()
            t0 = time.perf_counter()
            h = self.run_steps(k, k + steps, S)
            dev.sync()
            comm.device_synchronize()
            comm.barrier()
            dt = time.perf_counter() - t0
            el.append(comm.allreduce_max(dt))
            self.local_block_s.append(dt)
            hits.append(h)
            kern.append({name: dev.prof_read(kid) for kid, name in self.hip.PROF_NAMES.items()})
            self.block_work.append(self.work_log[w0:])
            k += steps
        dev.prof_enable(False)
        return el, hits, kern, k


def tame_leg(b, args, R):
    """SURVEY.md 8(d) config 3's SECOND profile: the same photons under ``2.5E+25 * exp(r2[gid] / 8600.0)``
    (examples/presentation_example_2.ipynb:41's atmosphere, kernel constant chosen so that pcoll spans 0.03 .. 1.8: exp never
    saturates, the hit fraction is position dependent and falls slowly) -- the branchy regime, where the example's own
    constants give inf / 0 after the first step.  Same store, filled again; same K-step pass, R blocks of --steps steps."""
    saved = b.prof
    b.prof = PROFILES["tame"]
    try:
        N, prof = b.N, b.prof
        b.dev.fill_photons(N, b.rank * N, C_LIT, b.e_lo, b.e_hi, args.seed)
        b.run_steps(0, args.warmup)                     # (includes the hipRTC lookup of the second expression)
        el, hits, kern, _ = b.timed_blocks(args.warmup, args.steps, R)
        mi = median_index(el)
        kb = kern[mi]["k_multi"]
        valu = valu_roofline(b.block_work[mi], kb["total_ms"], prof["expr"], False)
        hbm = N * 128.0 / (kb["avg_ms"] * 1e-3) / 1e9 if kb["launches"] else 0.0
        rec = {"workload": "SURVEY 8(d) config 3, tame profile: %.0e photons, variable_n_fn = %s, dt = %g, A (kernel) = %g"
                           % (N, prof["expr"], prof["dt"], prof["A_kernel"]),
               "value": N * args.steps / el[mi], "unit": "particle-steps/s", "ms_per_step": el[mi] / args.steps * 1e3, "steps": args.steps,
               "repeats": R, "repeat_ms_per_step": [round(e / args.steps * 1e3, 5) for e in el],
               "repeat_hit_fraction": [round(h / float(N * args.steps), 6) for h in hits],
               "roofline": (dict(valu, bound="valu", kernel="k_multi (hipRTC specialisation of the tame expression)",
                                 valu_busy=None,   # (the committed counter records are keyed by code object, not by expression: the example's)
                                 traffic=pmc_traffic(max(valu["kernel_forms"], key=valu["kernel_forms"].get), N), avg_launch_ms=kb["avg_ms"], launches=kb["launches"],
                                 hbm={"achieved": hbm, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBPS,
                                      "algorithmic_bytes_per_particle": 128.0})
                            if valu is not None else
                            {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBPS,
                             "traffic": pmc_traffic("pcl_rtc_multi_e1", N), "algorithmic_bytes_per_particle": 128.0,
                             "avg_launch_ms": kb["avg_ms"], "launches": kb["launches"]})}
        return rec
    finally:
        b.prof = saved


def kernel_summary(kern_blocks, name):
    """All timed launches of one kernel over the blocks: launches, average / min / max launch ms."""
    n = sum(b[name]["launches"] for b in kern_blocks)
    tot = sum(b[name]["total_ms"] for b in kern_blocks)
    live = [b[name] for b in kern_blocks if b[name]["launches"]]
    return {"launches": n, "avg_ms": tot / n if n else 0.0, "min_ms": min((b["min_ms"] for b in live), default=0.0),
            "max_ms": max((b["max_ms"] for b in live), default=0.0),
            "per_block_avg_ms": [round(b[name]["avg_ms"], 5) for b in kern_blocks]}


def median_index(xs):
    """Index of the median element (the lower one for even lengths): the block every per-block figure refers to."""
    order = sorted(range(len(xs)), key=lambda i: xs[i])
    return order[(len(xs) - 1) // 2]


def static_profile(profile, mode, dtype, N, S, steps):
    """PMC-counter figures of a COMMITTED rocprofv3 run of this configuration (profiles/pmc_traffic.json): HBM bytes
    per launch and the SQ (VALU) counters.  Not measured by this process -- nested here with their source, the K and
    the hit fraction they were taken at; nothing in the line is derived from them."""
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tfile):
        return None
    try:
        table = json.load(open(tfile))
    except ValueError:
        return None
    for key in ("%s:%s%s:%d:K%d:steps%d" % (profile, mode, "-f32" if dtype == "f32" else "", N, S, steps),
                "%s:%s%s:%d" % (profile, mode, "-f32" if dtype == "f32" else "", N)):
        if key in table and table[key].get("csrc_sha") == csrc_sha():          # (a run of other device sources is not quoted)
            return dict(table[key], key=key, file="profiles/pmc_traffic.json",
                        note="from a committed rocprofv3 --pmc run of this configuration, not from this process")
    return None


def _json_file(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return {}


_CSRC_SHA = None


def csrc_sha():
    global _CSRC_SHA
    if _CSRC_SHA is None:
        from physicl_amd import build
        _CSRC_SHA = build.csrc_sha()
    return _CSRC_SHA


def pmc_record(kernel):
    """The committed rocprofv3 --pmc record of one kernel (profiles/pmc_traffic.json, section "kernels"), or None when there is
    none OR when it was measured on other device sources than the ones this process runs (``csrc_sha``): a counter figure
    of a kernel that has changed since is not quoted."""
    rec = _json_file("pmc_traffic.json").get("kernels", {}).get(kernel)
    if not rec or rec.get("csrc_sha") != csrc_sha():
        return None
    return rec


def pmc_traffic(kernel, units):
    """HBM bytes of one launch of ``kernel`` from the COMMITTED rocprofv3 --pmc run (profiles/pmc_traffic.json, section
    "kernels": bytes per photon or per slot = FETCH_SIZE x 2 + WRITE_SIZE per MI355X_MICROARCH.md, separate passes, and
    the commit / profile file they were measured at) x the units this launch processed.  PMC counters cannot be read
    from inside the process: the record says where the figure comes from."""
    rec = pmc_record(kernel)
    if not rec or "bytes_per_unit" not in rec:
        return None
    return {"bytes": rec["bytes_per_unit"] * units, "bytes_per_unit": rec["bytes_per_unit"], "unit": rec["unit"],
            "source": rec["source"], "measured_at_commit": rec.get("commit"), "csrc_sha": rec.get("csrc_sha"),
            "note": "committed rocprofv3 --pmc run, not this process"}


def ahead_valu_roofline(work, kern_ms, clock_ghz=0.0, ids=False):
    """VALU-issue roofline of k_delete_ahead_live over the launches of one run.  Issue work = the kernel's own tally (groups
    of 128 slots loaded -- their first pass decided on the spot, two bodies or one --, rounds of 64 listed photons deciding
    two bodies / one body: pcl_store_ahead_work) x the four instruction counts of profiles/isa_counts.json (least squares of
    SQ_INSTS_VALU on that tally, profiles/r04_calib_ahead.md) x the mean price of an instruction of the kernel's mix
    (tools/isa_count.py --aot; classes priced by tools/valu_issue_probe.hip) = SIMD-cycles; ceiling = 1024 SIMDs x the clock
    the launches held (pcl_store_ahead_clock)."""
    tab = _json_file("isa_counts.json")
    c = tab.get("k_delete_ahead_live<double>")
    mix = tab.get("aot", {}).get("k_delete_ahead_live<double, %s>" % ("true" if ids else "false"))
    if not c or not mix or not kern_ms or not sum(work[:4]):
        return None
    g2, g1, r2, r1 = work[:4]
    groups = g2 + g1
    names = ("valu_per_group_first_pass_two_bodies", "valu_per_group_first_pass_one_body", "valu_per_round_two_bodies", "valu_per_round_one_body")
    instr = sum(c[nm] * w for nm, w in zip(names, work))
    peak = valu_peak(clock_ghz)
    ach = instr * mix["cycles_per_valu"] / (kern_ms * 1e-3)
    rec = pmc_record("k_delete_ahead_live<double>") or {}
    return {"bound": "valu", "kernel": "k_delete_ahead_live (the run's loop bodies worked out a launch at a time for the photons still alive)",
            "achieved": ach, "peak": peak, "unit": "SIMD-cycles/s", "frac": ach / peak,
            "frac_at_4_waves_per_simd": instr * mix["cycles_per_valu_at_4_waves"] / (kern_ms * 1e-3) / peak,
            "clock_GHz": clock_ghz if clock_ghz > 0 else None, "valu_busy": rec.get("valu_busy"), "wave_instructions": instr,
            "cycles_per_wave_instruction": mix["cycles_per_valu"],
            "work": {"groups_of_128_slots_first_pass_two_bodies": g2, "groups_of_128_slots_first_pass_one_body": g1, "rounds_two_bodies": r2,
                     "rounds_one_body": r1},
            "instruction_counts": {nm: c[nm] for nm in names},
            # (False: the kernel has changed since its four counts were fitted -- tools/prof_calib_ahead.sh)
            "instruction_counts_source": c["source"], "instruction_counts_current": c.get("csrc_sha") == csrc_sha(), "total_ms": kern_ms,
            "peak_note": "1024 SIMDs x the clock measured in the kernel (s_memtime / s_memrealtime); an instruction costs 2 / 4 / 8 / 16 cycles by class",
            "traffic": pmc_traffic("k_delete_ahead_live<double>", groups * 128),
            "hbm": {"achieved": 33.0 * groups * 128 / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": 33.0 * groups * 128 / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_slot": 33.0}}


def valu_roofline(work, kern_ms, expr, f32):
    """VALU-issue roofline of the K-step pass over a set of launches.  ``work``: per launch (steps, hits, dense passes,
    wave-steps, photons per wave, wave-steps on exp's shortcut, GHz the chip held) as the kernel tallied them; ``kern_ms``: the
    sum of those launches' durations (HIP events).
    Issue work in SIMD-cycles = decision instructions x wave-steps (+ the launch's fixed work per wave and grid-stride trip)
    x the mean price of the decision part's instruction mix + the dense pass's cycles x dense passes -- counts and prices per
    code object from profiles/isa_counts.json (tools/isa_count.py: the hipRTC translation unit's gfx950 assembly; the decision
    count is the dynamic one, calibrated against SQ_INSTS_VALU of the committed PMC run; the dense pass is a straight-line loop
    body; an instruction costs 2 / 4 / 8 / 16 cycles by class, tools/valu_issue_probe.hip).
    Ceiling = 1024 SIMDs x the clock the launches held.  ``frac`` prices instructions at what a SIMD with many waves pays (a
    true ceiling); ``frac_at_4_waves_per_simd`` at what the probe measured with the four waves these kernels keep per SIMD."""
    table = _json_file("isa_counts.json").get(expr, {}).get("kernels", {})
    instr = cycles = cycles4 = useful = 0.0
    clk_w = clk_n = 0.0
    forms = {}
    for w in work:
        steps, hits, passes, wsteps, ppw, sat = w[:6]
        ghz = w[6] if len(w) > 6 else 0.0
        name = "pcl_rtc_multi_f_e1" if f32 else ("pcl_rtc_multi%s%s_e1" % ({256: "2", 192: "3"}.get(ppw, ""), "s" if sat >= 0 else ""))
        c = table.get(name)
        if c is None or "dense_pass_cycles" not in c:
            return None
        A, B = c["decision_valu_per_wave_step"], c["dense_pass_valu"]
        trips = wsteps / float(steps)                  # waves x grid-stride trips: the launch's fixed work per wave
        # the variant with the saturation probe: wave-steps on exp's shortcut run the shorter decision part
        if sat < 0:
            dec = A * wsteps + c.get("decision_valu_per_wave_trip", 0.0) * trips
        else:
            f = sat / float(wsteps)
            dec = (c.get("decision_valu_per_wave_step_shortcut", A) * sat + A * (wsteps - sat) +
                   (c.get("decision_valu_per_wave_trip_shortcut", 0.0) * f + c.get("decision_valu_per_wave_trip", 0.0) * (1.0 - f)) * trips)
        instr += dec + B * passes
        cycles += dec * c["decision_cycles_per_valu"] + c["dense_pass_cycles"] * passes
        cycles4 += dec * c["decision_cycles_per_valu_at_4_waves"] + c["dense_pass_cycles_at_4_waves"] * passes
        useful += dec * c["decision_cycles_per_valu"] + c["dense_pass_cycles"] * hits / 64.0
        forms[name] = forms.get(name, 0) + 1
        if ghz > 0:
            clk_w += ghz * wsteps
            clk_n += wsteps
    if not cycles or not kern_ms:
        return None
    clock = clk_w / clk_n if clk_n else 0.0
    peak = valu_peak(clock)
    achieved = cycles / (kern_ms * 1e-3)
    return {"achieved": achieved, "peak": peak, "unit": "SIMD-cycles/s", "frac": achieved / peak,
            # the same against the data sheet's 2.4 GHz (rounds 2-4 priced against that; the chip holds 2.1-2.4 under this kernel):
            # the figure to compare ACROSS rounds and boxes, ``frac`` (at the clock the launch held) the one that says how much of
            # what the chip offered the kernel used
            "frac_at_spec_clock": achieved / valu_peak(0.0), "spec_clock_GHz": CLOCK_GHZ,
            "frac_at_4_waves_per_simd": cycles4 / (kern_ms * 1e-3) / peak, "clock_GHz": clock if clock > 0 else None,
            "lane_util": useful / cycles, "wave_instructions": instr, "issue_cycles": cycles, "dense_passes": sum(w[2] for w in work),
            "wave_steps": sum(w[3] for w in work), "dense_passes_per_wave_step": sum(w[2] for w in work) / float(sum(w[3] for w in work)),
            "saturated_wave_steps": sum(max(w[5], 0) for w in work),
            "kernel_forms": forms,
            # (the 192-photon form keeps FIVE waves per SIMD: its instructions cost ~1.5 % less than the at-4-waves price)
            "waves_per_simd": {k: (5 if "multi3" in k else 4) for k in forms},
            "instruction_counts": {k: {key: table[k][key] for key in ("decision_valu_per_wave_step", "decision_valu_per_wave_step_shortcut",
                                                                                           "decision_valu_per_wave_trip", "decision_valu_per_wave_trip_shortcut",
                                                                                           "dense_pass_valu", "dense_pass_cycles", "decision_cycles_per_valu",
                                                                                           "dense_pass_cycles_at_4_waves", "decision_cycles_per_valu_at_4_waves",
                                                                                           "dense_pass_classes", "decision_classes_static") if key in table[k]} for k in forms},
            "instruction_counts_source": "profiles/isa_counts.json (tools/isa_count.py)",
            "peak_note": "%d SIMDs x the clock measured in the kernel (s_memtime / s_memrealtime; %.1f GHz where a launch has none); an instruction "
                         "costs 2 / 4 / 8 / 16 cycles by class (profiles/r05_valu_issue_probe.txt)" % (N_SIMD, CLOCK_GHZ)}


def static_valu(kernel):
    """valu_busy / lane utilisation of ``kernel`` from the committed SQ-counter pass (None when the device sources changed since)."""
    rec = pmc_record(kernel)
    if not rec or "valu_busy" not in rec:
        return None, None
    return rec["valu_busy"], rec.get("source")


def run_rank(args):
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when it
    # initialises): everything but the final line goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    b = Bench(args)
    comm, dev, N, S, prof, world, rank = b.comm, b.dev, b.N, b.S, b.prof, b.world, b.rank
    hip = b.hip
    f32 = args.dtype == "f32"
    bscale = 0.5 if f32 else 1.0
    if args.dry_run:
        # the collective is up (Bench() raised otherwise): one all-reduce of a per-rank vector as a last check, then out
        devices = comm.allgather_object({"rank": rank, "device": b.dev_index, "pci": dev.info().get("pci_bus_id")})
        probe = comm.allreduce_sum(np.array([1, rank], dtype=np.int64))
        dev.close()
        comm.close()
        if rank == 0:
            ok = int(probe[0]) == world and int(probe[1]) == world * (world - 1) // 2
            os.write(json_fd, (json.dumps({"dry_run": True, "n_gpus": world, "ok": ok, "collective": dict(comm.info(), devices=devices),
                                           "distinct_pci": len({d["pci"] for d in devices})}) + "\n").encode())
            if not ok:
                sys.exit(1)
        return
    b.fill()
    slab_info = dev.alloc_info()

    # every rank reports which physical device it drives (N ranks on N distinct devices is what RCCL needs) and which
    # memory its store got: an imbalance between ranks is visible in the line
    devices = comm.allgather_object({"rank": rank, "device": b.dev_index, "pci": dev.info().get("pci_bus_id"),
                                     "slab_selection": slab_info})

    dev.prof_enable(True)
    b.run_steps(0, args.warmup)
    warm = {name: dev.prof_read(kid) for kid, name in hip.PROF_NAMES.items()}

    R = max(1, args.repeats)
    el, hits, kern, k_next = b.timed_blocks(args.warmup, args.steps, R)
    b.log_rows = False                 # (the legs below run other formulations / profiles on the same store)
    b.block_work_main = b.block_work
    totals_main = b.totals
    rank_blocks = comm.allgather_object([round(x / args.steps * 1e3, 5) for x in b.local_block_s[:R]])
    mi = median_index(el)
    elapsed = el[mi]
    dominant = ("k_multi" if S > 1 else "k_fused") if args.mode.startswith("fused") else "k_scatter"
    ks = kernel_summary(kern, dominant)
    h_blocks = [h / float(N * args.steps) for h in hits]
    bpp = algorithmic_bytes_per_particle(args.profile, h_blocks[mi], args.mode, S > 1) * bscale
    # the roofline record refers to the SAME block as ``value`` (the median one): its launches, its hit fraction
    kb = kern[mi][dominant]
    achieved = N * bpp / (kb["avg_ms"] * 1e-3) / 1e9 if kb["launches"] else 0.0
    steps_per_timed_launch = args.steps / max(1, kb["launches"]) if S > 1 else 1.0

    single = delete = api = None
    extra = world == 1 and not args.no_extra
    if extra and S > 1:
        # the one-launch-per-step path on the same store, same run: the HBM-bound kernel's own roofline line
        b.run_steps(k_next, k_next + 3, S=1)
        el1, hits1, kern1, k_next = b.timed_blocks(k_next + 3, args.steps, R, S=1)
        m1 = median_index(el1)
        k1 = kernel_summary(kern1, "k_fused")
        b1 = 104.0 * bscale
        k1b = kern1[m1]["k_fused"]                  # the median block's launches, as for ``value``
        a1 = N * b1 / (k1b["avg_ms"] * 1e-3) / 1e9 if k1b["launches"] else 0.0
        single = {"value": N * args.steps / el1[m1], "unit": "particle-steps/s", "ms_per_step": el1[m1] / args.steps * 1e3,
                  "steps": args.steps, "repeats": R, "repeat_ms_per_step": [round(e / args.steps * 1e3, 5) for e in el1],
                  "roofline": {"bound": "hbm", "kernel": "k_fast (pcl_rtc_fast_e1): one launch per step, dr/dv implicit",
                               "achieved": a1, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a1 / HBM_PEAK_GBPS,
                               "traffic": pmc_traffic("pcl_rtc_fast_f_e1" if f32 else "pcl_rtc_fast_e1", N),
                               "algorithmic_bytes_per_particle": b1, "avg_launch_ms": k1b["avg_ms"],
                               "min_launch_ms": k1b["min_ms"], "max_launch_ms": k1b["max_ms"], "launches": k1b["launches"],
                               "all_blocks_avg_launch_ms": k1["avg_ms"], "per_block_avg_launch_ms": k1["per_block_avg_ms"],
                               "hit_fraction": hits1[m1] / float(N * args.steps)}}
        b.totals = totals_main

    tame = None
    if extra and S > 1 and args.profile == "example" and not f32:
        tame = tame_leg(b, args, R)
        b.totals = totals_main

    info = dev.info()
    dev.store_free()
    iso = mixed = None
    if extra and not f32:
        delete = delete_leg(dev, hip, [int(float(x)) for x in args.delete_photons.split(",") if x.strip()], args.seed)
        iso = iso_leg(dev, hip, int(args.iso_photons), args.seed)
        mixed = mixed_leg(dev, int(args.mixed_photons))
    dev.close()
    if extra and not f32:
        api = api_leg(args, prof)

    valu = valu_roofline(b.block_work_main[mi], kb["total_ms"], prof["expr"], f32) if S > 1 else None
    # the code object the median block's launches ran (the K-step pass has several instantiations; the kernel's tally says which)
    dom_rtc = (max(valu["kernel_forms"], key=valu["kernel_forms"].get) if valu is not None else
               {"k_multi": "pcl_rtc_multi_%se1" % ("f_" if f32 else ""), "k_fused": "pcl_rtc_fast_%se1" % ("f_" if f32 else "")}.get(dominant))
    hbm_rec = {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
               "traffic": pmc_traffic(dom_rtc, N) if (dom_rtc and args.mode == "fused") else None, "algorithmic_bytes_per_particle": bpp,
               "algorithmic_bytes_note": "per LAUNCH of the K-step pass (r, v, lam4 read; r, v, vprev written)" if S > 1 else "per particle-step"}
    common = {"hit_fraction": h_blocks[mi], "avg_launch_ms": kb["avg_ms"], "min_launch_ms": kb["min_ms"], "max_launch_ms": kb["max_ms"],
              "launches": kb["launches"], "block": "median block (index %d), as value" % mi, "all_blocks_avg_launch_ms": ks["avg_ms"],
              "all_blocks_launches": ks["launches"], "per_block_avg_launch_ms": ks["per_block_avg_ms"], "steps_per_launch_max": S,
              "steps_per_timed_launch": steps_per_timed_launch, "warmup_launches": warm[dominant]["launches"],
              "warmup_avg_launch_ms": warm[dominant]["avg_ms"]}
    if valu is not None:
        # The K-step pass moves 128 B per photon per LAUNCH and is bound by VALU issue: the record
        # says so.  ``traffic`` (HBM bytes of one launch, committed PMC run) and the HBM form of the same launch ride along.
        form = max(valu["kernel_forms"], key=valu["kernel_forms"].get)
        busy, busy_src = static_valu(form)
        roofline = dict(valu, bound="valu",
                        kernel="k_multi (%s): %d x (Newton + ScatterIsotropic + counters) per pass over the "
                               "store, dr/dv implicit; bound by VALU issue -- the HBM-bound formulation is roofline_hbm" % (form, S),
                        # what the SQ counters of the committed rocprofv3 run of this code object say (None: the device sources
                        # have changed since that run): the share of SIMD-cycles with a vector instruction in flight, and that
                        # share x the lanes that did useful work
                        valu_busy=busy, valu_busy_source=busy_src, useful=(busy * valu["lane_util"] if busy else None),
                        traffic=hbm_rec["traffic"], hbm=hbm_rec,
                        # SURVEY 8(d)'s per-step form: what the same particle-steps would have had to move one launch per step
                        # (104 B each).  > peak is possible precisely because the pass does not move those bytes.
                        per_step_form={"bytes_per_particle_step": 104.0 * bscale,
                                       "GBps": N * 104.0 * bscale * steps_per_timed_launch / (kb["avg_ms"] * 1e-3) / 1e9,
                                       "frac_of_peak": N * 104.0 * bscale * steps_per_timed_launch / (kb["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                        per_block=[(lambda v: None if v is None else {"frac": round(v["frac"], 4), "frac_at_4_waves_per_simd": round(v["frac_at_4_waves_per_simd"], 4),
                                                                         "clock_GHz": v["clock_GHz"], "lane_util": round(v["lane_util"], 4),
                                                                         "dense_passes_per_wave_step": round(v["dense_passes_per_wave_step"], 4),
                                                                         "forms": v["kernel_forms"]})(
                            valu_roofline(b.block_work_main[i], kern[i][dominant]["total_ms"], prof["expr"], f32)) for i in range(R)],
                        **common)
    else:
        roofline = dict(hbm_rec, bound="hbm",
                        kernel=(("k_multi: %d x (Newton + ScatterIsotropic + counters) per pass over the store, dr/dv implicit; "
                                 "arithmetic-bound by construction (no instruction counts committed for this code object: HBM form only)" % S)
                                if S > 1 else
                                "k_fast/k_fused: Newton + ScatterIsotropic + counters in one pass (hipRTC variable-n)%s"
                                % (", dr/dv implicit" if args.mode == "fused" else "")
                                if args.mode.startswith("fused") else
                                "k_scatter: ScatterIsotropicStep kernel + write-back (hipRTC variable-n)"),
                        **common)

    out = None
    if rank == 0:
        value = N * world * args.steps / elapsed
        if world > 1:
            coll_txt = " + %s all-reduce of the %s int64 counters" % (
                "RCCL" if comm.backend == "nccl" else "gloo", "%d x 5 per-launch" % S if S > 1 else "5")
        else:
            coll_txt = ""
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "repeats": R, "repeat_ms_per_step": [round(e / args.steps * 1e3, 5) for e in el],
            "repeat_hit_fraction": [round(h, 6) for h in h_blocks], "timing": "median of %d blocks of %d steps" % (R, args.steps),
            "config": {"workload": "BASELINE configs[2]%s: %.0e photons/GPU, variable-n + wavelength isotropic "
                                   "scatter (variable_n_scattering example); step = UpdateTime + Newton + "
                                   "ScatterIsotropic (Philox) + sign counters%s"
                                   % ("/[3] weak-scaled" if world > 1 else "", N, coll_txt),
                       "photons_per_gpu": N, "profile": args.profile, "mode": args.mode,
                       # S = the most loop bodies one launch may carry; a timed block of --steps steps is ceil(steps / S)
                       # launches, so the launches that were timed carried steps_per_timed_launch each
                       "steps_per_launch_max": S, "steps_per_timed_launch": steps_per_timed_launch,
                       "variable_n_fn": prof["expr"], "dt": prof["dt"],
                       "rng": "philox4x32-10 keyed by global photon id", "parallelism": "index-sharded x%d" % world,
                       # which memory the store's slab got (outside the timed region): the library measures a few candidate
                       # blocks with a write sweep when a big store is created and keeps the fastest ("Placement")
                       "slab_selection": slab_info},
            "collective": (dict(comm.info(), devices=devices,
                                # every rank's own time of every block (ms per step, barrier to barrier): min = max means the
                                # ranks ran in step, a straggler shows as one rank's row
                                per_rank_block_ms_per_step=rank_blocks,
                                block_min_ms_per_step=[min(r[i] for r in rank_blocks) for i in range(R)],
                                block_max_ms_per_step=[max(r[i] for r in rank_blocks) for i in range(R)])
                           if world > 1 else None),
            "roofline": roofline,
            # the north_star's HBM target (>= 60 % of the HBM roofline on the photon-scatter step at 1e8 photons) is about the
            # one-launch-per-step kernel: the same workload run that way in this same process (the single_step leg)
            "roofline_hbm": (dict(single["roofline"], value=single["value"], ms_per_step=single["ms_per_step"]) if single is not None else None),
            # every K-step launch of this process in order (warm-up, the timed blocks, the tame leg): steps, hits, dense passes,
            # wave-steps, photons per wave, wave-steps on exp's saturation shortcut (-1: the variant without the probe) -- what
            # tools/summarize_driver_prof.py lines up with the profiler's dispatches
            "k_step_launch_work": [list(w) for w in b.work_log],
            "static_profile": static_profile(args.profile, args.mode, args.dtype, N, S, args.steps),
            # north_star target (>= 60 % of the HBM roofline on the photon-scatter step at 1e8 photons): carried by the
            # one-launch-per-step kernel, measured in this same run (single_step); the K-step pass trades those bytes away
            "hbm_target": ({"kernel": "k_fast, one launch per step", "frac": single["roofline"]["frac"], "target": 0.6,
                            "met": single["roofline"]["frac"] >= 0.6} if single is not None else None),
            "kernels_ms": {k: round(kernel_summary(kern, k)["avg_ms"], 4) for k in kern[0] if kernel_summary(kern, k)["launches"]},
            "counters_last_step": {"N": int(totals_main[0]), "hits": int(totals_main[1]), "xp": int(totals_main[2]),
                                   "yp": int(totals_main[3]), "zp": int(totals_main[4])},
            "device": info["name"],
            # the device sources this process ran (physicl_amd.build.csrc_sha): what a profile summary of this run tags its
            # counter records with -- the record is about THIS build, whatever the tree looks like when it is summarised
            "csrc_sha": csrc_sha(),
        }
        if single is not None:
            out["single_step"] = single
        if tame is not None:
            out["tame"] = tame
        if delete is not None:
            out["delete"] = delete
        if iso is not None:
            out["iso_1e7"] = iso
        if mixed is not None:
            out["mixed"] = mixed
        if api is not None:
            out["api"] = api
        if world == 1 and not args.no_cpu_baseline and not f32:
            dev2 = hip.Device(b.dev_index)
            try:
                out["cpu_baseline"] = cpu_baseline(dev2, args, prof, b.rows_log)
                out["cpu_baseline_python"] = cpu_baseline_python(dev2, args, prof, 10_000)        # 1e4 and 1e5
                out["cpu_baseline_python_1e5"] = cpu_baseline_python(dev2, args, prof, 100_000)
                out["cpu_baseline_numpy"] = cpu_baseline_numpy(dev2, args, prof)
                out["cpu_baseline_python_units"] = cpu_baseline_python_units(args, prof)
            finally:
                dev2.close()

    comm.close()
    if rank == 0:
        # The contract is ONE short JSON line, the last thing on stdout.  Everything the run measured goes to
        # bench_detail.json next to this file (and, as one line, to stderr BEFORE the contract line is written).
        detail_file = write_detail(out)
        sys.stdout.flush()
        sys.stderr.write(json.dumps(out) + "\n")
        sys.stderr.flush()
        os.write(json_fd, (json.dumps(contract_line(out, detail_file)) + "\n").encode())


DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")
LINE_LIMIT = 4000              # characters: the driver keeps the last ~8 000 of stdout, the line must fit with room to spare


def write_detail(out):
    """Everything the run measured (per-block tables, every leg, the instruction-count dictionaries, the other CPU
    legs) as indented JSON next to bench.py.  Returns the path relative to the repo root, or None when the directory
    cannot be written (the contract line then says so: the line itself never depends on the file)."""
    path = os.environ.get("PCL_BENCH_DETAIL", DETAIL_FILE)
    try:
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        os.replace(tmp, path)
    except OSError as e:
        sys.stderr.write("bench.py: could not write %s: %s\n" % (path, e))
        return None
    return os.path.relpath(path, ROOT)


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _traffic_number(t):
    """The contract's ``traffic``: HBM bytes of one launch from the PMC counters, a number (or null).  Where the figure
    comes from (profile file, commit) is in ``traffic_source`` beside it."""
    return (None, None) if not t else (t["bytes"], "%s @ %s" % (t.get("source"), t.get("measured_at_commit")))


def contract_line(out, detail_file):
    """The one stdout line: the bench contract's keys and nothing else.  ``out`` is the full record of the run."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data", "repeats", "repeat_ms_per_step"))
    line["config"] = _pick(out["config"], ("workload", "photons_per_gpu", "profile", "mode", "steps_per_launch_max",
                                           "steps_per_timed_launch", "variable_n_fn", "dt", "rng", "parallelism"))
    r = out["roofline"]
    rl = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_at_spec_clock", "frac_at_4_waves_per_simd", "valu_busy", "lane_util", "useful", "clock_GHz",
                   "avg_launch_ms", "launches", "hit_fraction"))
    rl["kernel"] = rl.get("kernel", "").split(":")[0].split(";")[0][:120]
    rl["traffic"], rl["traffic_source"] = _traffic_number(r.get("traffic"))
    line["roofline"] = rl
    h = out.get("roofline_hbm")
    if h is not None:
        hl = _pick(h, ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches", "value", "ms_per_step"))
        hl["traffic"], hl["traffic_source"] = _traffic_number(h.get("traffic"))
        hl["algorithmic_bytes_per_particle"] = h.get("algorithmic_bytes_per_particle")
        line["roofline_hbm"] = hl
    else:
        line["roofline_hbm"] = None
    c = out.get("cpu_baseline")
    line["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "nproc", "kind", "sample", "rows_match", "rows_compared")) if c else None
    coll = out.get("collective")
    if coll is not None:
        line["collective"] = dict(_pick(coll, ("backend", "ranks_seen", "world", "rccl_version", "block_min_ms_per_step",
                                               "block_max_ms_per_step")),
                                  devices=[_pick(d, ("rank", "device", "pci")) for d in coll.get("devices", [])])
    else:
        line["collective"] = None
    line["counters_last_step"] = out.get("counters_last_step")
    # the other BASELINE configurations that rode along (N = 1): one figure each, particle-steps/s; the records are in the detail file
    legs = {}
    for name, path in (("iso_1e7_per_step", ("iso_1e7", "per_step", "value")), ("iso_1e7_multi", ("iso_1e7", "multi", "value")),
                       ("mixed_f64", ("mixed", "value_f64")), ("mixed_f32", ("mixed", "value_f32")), ("tame", ("tame", "value")),
                       ("api_default", ("api", "default", "value")), ("api_trace", ("api", "trace_default", "value")),
                       ("api_delete_default", ("api", "delete_default", "value"))):
        v = out
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
        if v is not None:
            legs[name] = float("%.4g" % v)
    for size, rec in (out.get("delete") or {}).get("sizes", {}).items():
        for mode in ("per_step", "multi"):
            if mode in rec:
                legs["delete_%s_%s" % (size.replace("+", ""), mode)] = float("%.4g" % rec[mode]["value"])
    if legs:
        line["legs"] = legs
    # ``value`` is the median timed block of a run whose hit fraction falls from block to block; the same workload's WHOLE run
    # (500 passes through the plugin API, start() .. join()) is the steady figure: beside it, one key away
    if "api_default" in legs:
        line["whole_run_value"] = legs["api_default"]
    line["device"] = out.get("device")
    line["detail_file"] = detail_file
    s = json.dumps(line)
    if len(s) >= LINE_LIMIT:                 # never let the line outgrow the driver's window: drop the optional parts first
        for k in ("legs", "counters_last_step", "repeat_ms_per_step"):
            line.pop(k, None)
            if len(json.dumps(line)) < LINE_LIMIT:
                break
    return line


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[1](ii): Newton + ScatterDeleteStep(A = n = 1e-3) + plane counter until no photon is left
# (test/test_light.py:52-59; physicl/light.py:231-260, physicl/__init__.py:455-459)
# ---------------------------------------------------------------------------------------------------------------------
def delete_leg(dev, hip, sizes, seed, repeats=3, K=16):
    """Delete-until-empty at each size, two formulations: "per_step" = one call per loop body (pcl_step_fused_delete: the
    body runs on the store's alive mask -- ONE kernel, k_delete_alive, nothing moves -- and the store is compacted, flag
    kernel -> scan -> k_compact_*, only in the bodies that start with fewer than half of the slots alive; from a population's
    first delete body on the library works the next bodies out in one sweep, k_delete_ahead, and answers the calls from
    those rows -- "bodies_answered_by" says how many launches the run really took); "multi" = K
    loop bodies per pass and ONE compaction (pcl_step_fused_delete_multi), the form Simulation runs by itself when the exit
    test allows it.  A "particle-step" is one photon alive at the start of one loop body.  The roofline records count the
    bytes of the SLOTS a kernel sweeps, dead ones included (fp64): k_delete_alive reads v (24) and, for the plane
    counter, r (24) of every slot, the id (8) once ids are explicit, and reads + writes the alive bit; the compaction
    reads a bit per slot and moves r, v, E, id of the survivors (64 B each way; dv, all +0.0 in a run that never
    scatters, travels in the first compaction only)."""
    plane = np.array([[1.0 / (1e-3 * 1e-3), np.nan, np.nan]])          # test/test_light.py:58
    dt, A, n = 1e-3, 1e-3, 1e-3
    out = {"workload": "BASELINE configs[1](ii): Newton + ScatterDelete(A=n=1e-3) + plane counter until empty, E = 1, "
                       "v = (c,0,0), dt = 1e-3, Philox", "repeats": repeats, "sizes": {}}
    for N in sizes:
        dev.store_alloc(N)
        rec = {}
        for mode in ("per_step", "per_step_no_ahead", "multi"):
            # per_step_no_ahead: the same calls with PCL_AHEAD=0 -- every body is its own launch (round 3's path + this round's
            # narrower reads): what the bodies worked out ahead of their calls are worth, in the same line
            hip.set_knob("PCL_AHEAD", "0" if mode == "per_step_no_ahead" else None)
            runs = []
            # rep 0 = warm-up (allocations of the second slab, first touch); reps 1..repeats are timed WITHOUT the HIP-event
            # pairs around the kernels (a loop body of a small store is a 10 us kernel: the two event records per kernel
            # cost 8 % of the 1e7 run); one more rep runs with them for the per-kernel times of the roofline records
            for rep in range(repeats + 2):
                dev.fill_photons(N, 0, C_LIT, 1.0, 1.0, seed)
                dev.prof_enable(rep == repeats + 1)
                dev.sync()
                aw0 = dev.ahead_work()
                t0 = time.perf_counter()
                work, per_step, k = 0, [], 0
                nb = sb = N                                   # alive photons / slots of the store (dense after the fill)
                track = rep == repeats + 1 and mode.startswith("per_step")   # the instrumented run also notes how each body was answered
                if track:
                    how, st_prev = [], dev.ahead_stats()
                while nb > 0 and k < 4096:
                    if mode == "multi":
                        for o in dev.step_fused_delete_multi(dt, K, A, n, seed, k, plane):
                            work += nb
                            per_step.append((nb, o["N"], nb, o["N"]))
                            nb = o["N"]
                        k += K
                    else:
                        o = dev.step_fused_delete(dt, A, n, hip.RNG_PHILOX, seed, k, plane, lazy=True)
                        work += nb
                        sa = dev.slots
                        per_step.append((nb, o["N"], sb, sa))             # alive before / after, slots before / after
                        if track:                                         # "kernel": k_delete_alive ran; "ahead_launch": k_delete_ahead
                            st_now = dev.ahead_stats()                    # ran (this body + the next ones); "ahead": no launch at all
                            how.append("ahead_launch" if st_now[0] > st_prev[0] else ("ahead" if st_now[1] > st_prev[1] else "kernel"))
                            st_prev = st_now
                        nb, sb = o["N"], sa
                        k += 1
                dev.sync()
                el = time.perf_counter() - t0
                if mode == "multi":
                    how = []
                if rep == repeats + 1:
                    kern = {name: dev.prof_read(kid) for kid, name in hip.PROF_NAMES.items()}
                    ahead_work = [b - a for a, b in zip(aw0, dev.ahead_work())]
                    ahead_clock = dev.ahead_clock()          # (GHz held under this context's k_delete_ahead_live launches so far)
                    instrumented_ms = el * 1e3
                    dev.prof_enable(False)
                elif rep:
                    runs.append((el, work, per_step))
            runs.sort(key=lambda r: r[0])
            el, work, per_step = runs[(len(runs) - 1) // 2]
            tot = sum(b[0] for b in per_step if b[0])
            surv = sum(b[1] for b in per_step if b[0])
            r = {"value": work / el, "unit": "particle-steps/s", "ms_total": el * 1e3, "loop_bodies": len([1 for b in per_step if b[0]]),
                 "particle_steps": work, "survivor_fraction": surv / float(tot) if tot else 0.0,
                 "run_ms": [round(x[0] * 1e3, 4) for x in runs],
                 "timing": "median of %d runs without per-kernel events; kernel times from one more run with them (%.4f ms)"
                           % (len(runs), instrumented_ms),
                 "kernels_total_ms": {kname: round(v["total_ms"], 4) for kname, v in kern.items() if v["launches"]},
                 "kernel_launches": {kname: v["launches"] for kname, v in kern.items() if v["launches"]}}
            if mode.startswith("per_step"):
                b1 = b2 = b3 = 0.0
                slots_swept = slots_ahead = compactions = 0
                explicit_ids = False
                big = []                                          # the compactions of >= 1e7 slots, one by one
                for (nb, na, sb, sa), h in zip(per_step, how + ["kernel"] * len(per_step)):
                    if not nb:
                        continue
                    compacting = sa < sb                           # the body ended on a smaller extent: it compacted
                    # k_delete_alive: alive bit read + written, v, the id once explicit; r ALONG THE PLANE'S AXIS (8 of its 24
                    # bytes) only when it counts the plane crossings itself (a compacting body leaves the counters to the
                    # compaction).  k_delete_ahead (K bodies of a small store in one launch): the same reads once + a byte written.
                    per_slot = 0.125 + 24.0 + (8.0 if explicit_ids else 0.0) + (0.0 if compacting else 8.0)
                    if h == "kernel":
                        slots_swept += sb
                        b1 += sb * (per_slot + 0.125)
                    elif h == "ahead_launch":
                        slots_ahead += sb
                        b2 += sb * (per_slot + 1.0)
                    if compacting:
                        compactions += 1
                        cb = sb * 0.125 + na * 2.0 * 64.0
                        b3 += cb
                        if sb >= 10_000_000:
                            big.append({"slots": sb, "survivors": na, "algorithmic_bytes": cb,
                                        "source_bytes_at_line_granularity": sb * 56.0 + (sb * 8.0 if explicit_ids else 0.0) + na * 64.0})
                        explicit_ids = True
                p1_ms, p2_ms, p3_ms = kern["k_delete_mask"]["total_ms"], kern["k_delete_ahead"]["total_ms"], kern["k_compact"]["total_ms"]
                g1 = b1 / (p1_ms * 1e-3) / 1e9 if p1_ms else 0.0
                g3 = b3 / (p3_ms * 1e-3) / 1e9 if p3_ms else 0.0
                answered = {h: how.count(h) for h in ("kernel", "ahead_launch", "ahead")}
                r["bodies_answered_by"] = dict(answered, note="kernel: one k_delete_alive launch (or flag + scan + compaction); ahead_launch: "
                                               "one k_delete_ahead launch worked out this body and the next ones; ahead: answered from those "
                                               "rows, no launch (the commit behind the last of them also runs the compaction that has become due)")
                alive_rec = {"bound": "hbm", "kernel": "k_delete_alive (one loop body on the alive mask: Newton + delete flag + counters, nothing moves)",
                                 "achieved": g1, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": g1 / HBM_PEAK_GBPS,
                                 "traffic": pmc_traffic("k_delete_alive<double, true>", slots_swept),
                                 "algorithmic_bytes": b1, "total_ms": p1_ms, "slots_swept": slots_swept, "alive_particle_steps": tot,
                                 "bytes_per_alive_particle_step": (b1 + b2 + b3) / tot if tot else 0.0}
                # the record of the kernel that does the run's work: k_delete_ahead_live when the bodies are worked out ahead (its
                # VALU roofline, from the kernel's own tally; k_delete_alive then sweeps next to nothing and keeps its figures as
                # "roofline_alive"), k_delete_alive otherwise
                va = ahead_valu_roofline(ahead_work, kern["k_delete_ahead"]["total_ms"], ahead_clock)
                if va:
                    r["roofline"], r["roofline_alive"] = va, alive_rec
                else:
                    r["roofline"] = alive_rec
                r["ahead"] = {"kernel": "k_delete_ahead_live (the next loop bodies worked out in ONE sweep of the extent -- 24 for stores of <= 2^22 "
                                        "slots, 16 up to 2^25, 12 above --, answered call by call from the rows; the store is only written at the "
                                        "commit; the kernel lists the photons still alive per 256 slots, so a body costs what they cost)",
                              "launches": kern["k_delete_ahead"]["launches"], "total_ms": p2_ms, "slots_swept": slots_ahead,
                              "algorithmic_bytes": b2, "bodies": answered["ahead_launch"] + answered["ahead"],
                              "traffic": pmc_traffic("k_delete_ahead_live<double>", slots_ahead),
                              # its HBM rate on the slots it sweeps: well below the streaming kernels' -- the bodies' Philox blocks
                              # (twenty quarter-rate 32 x 32 -> 64 multiplies each), compares and ballots bind it, not bytes
                              "achieved_GBps": b2 / (p2_ms * 1e-3) / 1e9 if p2_ms else 0.0,
                              "frac_of_hbm_peak": b2 / (p2_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if p2_ms else 0.0, "bound": "valu"}
                r["roofline_compaction"] = {"bound": "hbm", "kernel": "k_compact_* (stable compaction of the survivors, %d of %d bodies)"
                                                                      % (compactions, r["loop_bodies"]),
                                            "achieved": g3, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": g3 / HBM_PEAK_GBPS,
                                            # (since round 4 the compactions of big extents start from the masks of six bodies worked
                                            # out ahead: ~8 % survivors, which the scan hands to the direct kernel)
                                            "traffic": pmc_traffic("k_compact_count<double, unsigned long, 7>" if mode == "per_step" else
                                                                   "k_compact_lds<double, unsigned long, 7>", sum(c["slots"] for c in big if c["slots"] >= 90_000_000))
                                                       if any(c["slots"] >= 90_000_000 for c in big) else None,
                                            "algorithmic_bytes": b3, "total_ms": p3_ms, "compactions": compactions,
                                            # the bytes HBM cannot avoid serving: survivors are scattered at random, so every 128-byte
                                            # line of the seven source rows holds one (P(16 neighbours all removed) < 0.2 % at 34 %
                                            # survivors) and is read whole -- 56 B per SLOT, not 56 B per survivor
                                            "compactions_of_1e7_slots_or_more": big}
            else:
                r["steps_per_launch"] = K
                va = ahead_valu_roofline(ahead_work, kern["k_delete_ahead"]["total_ms"], ahead_clock)   # (K-body calls on an all-photon store
                if va:                                                                      # take the single calls' path)
                    r["roofline"] = va
            rec[mode] = r
        hip.set_knob("PCL_AHEAD", None)
        out["sizes"]["%.0e" % N] = rec
        dev.store_free()
    return out


def iso_leg(dev, hip, N, seed, steps=100, repeats=3):
    """BASELINE configs[1](i): ``N`` photons (1e7), r = 0, v = (c,0,0), E = 1, dt = 1e-3, 100 passes of
    [UpdateTime, Newton, ScatterIsotropicStep(A = n = 1e-3), sign rows] (test/test_light.py:27-45; pcoll = A n |v dt| =
    0.2998).  Two formulations on the same photons: one launch per step (k_fast: r, v read and written, 96 B per
    particle-step -- the HBM-bound form, with its roofline record) and up to 32 steps per launch (k_multi)."""
    sc = dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=H_LIT, rng_mode=hip.RNG_PHILOX, seed=seed)
    out = {"workload": "BASELINE configs[1](i): %.0e photons, E = 1, v = (c,0,0), dt = 1e-3, 100 x [UpdateTime, Newton, "
                       "ScatterIsotropic(A=n=1e-3), sign counters], Philox" % N, "steps": steps, "repeats": repeats}
    dev.store_alloc(N)
    for mode in ("per_step", "multi"):
        runs = []
        for rep in range(repeats + 1):                        # rep 0 = warm-up
            dev.fill_photons(N, 0, C_LIT, 1.0, 1.0, seed)
            dev.prof_enable(True)
            dev.sync()
            t0 = time.perf_counter()
            hits, k = 0, 0
            if mode == "multi":
                while k < steps:
                    ks = min(32, steps - k)
                    hits += sum(o["hits"] for o in dev.step_fused_multi(1e-3, ks, dict(sc, step=k)))
                    k += ks
            else:                                             # software-pipelined like Bench.run_steps: counters one step behind
                dev.step_fused(1e-3, dict(sc, step=0), planes=(), sync=False, lazy=True)
                for k in range(1, steps):
                    dev.step_fused(1e-3, dict(sc, step=k), planes=(), sync=False, lazy=True)
                    hits += dev.step_fused_read(0)["hits"]
                hits += dev.step_fused_read(0)["hits"]
            dev.sync()
            el = time.perf_counter() - t0
            kern = {name: dev.prof_read(kid) for kid, name in hip.PROF_NAMES.items()}
            dev.prof_enable(False)
            if rep:
                runs.append((el, hits, kern))
        runs.sort(key=lambda r: r[0])
        el, hits, kern = runs[(len(runs) - 1) // 2]
        r = {"value": N * steps / el, "unit": "particle-steps/s", "ms_per_step": el / steps * 1e3, "hit_fraction": hits / float(N * steps),
             "run_ms": [round(x[0] * 1e3, 4) for x in runs],
             "kernels_total_ms": {kname: round(v["total_ms"], 4) for kname, v in kern.items() if v["launches"]},
             "kernel_launches": {kname: v["launches"] for kname, v in kern.items() if v["launches"]}}
        if mode == "per_step":
            kf = kern["k_fused"]
            g = N * 96.0 / (kf["avg_ms"] * 1e-3) / 1e9 if kf["launches"] else 0.0
            r["roofline"] = {"bound": "hbm", "kernel": "k_fast<double, no wavelength term, constant n>: one launch per step, dr/dv implicit",
                             "achieved": g, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": g / HBM_PEAK_GBPS, "traffic": None,
                             "algorithmic_bytes_per_particle": 96.0, "avg_launch_ms": kf["avg_ms"], "launches": kf["launches"]}
        else:
            r["steps_per_launch_max"] = 32
        out[mode] = r
    dev.store_free()
    return out


def mixed_valu_record():
    """What binds k_mixed (the K-pass kernel of configs[4]): VALU issue.  Its instruction count per launch depends on the
    photons' histories, and the kernel keeps no tally of its own, so this record is NOT computed in this process: it is the
    committed rocprofv3 run of the same command (profiles/pmc_traffic.json: SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU and
    GRBM_GUI_ACTIVE of the fp64 16-iteration launches, durations from the trace pass).  ``frac`` = wave-instructions x the
    mean price of the kernel's instruction mix / (1024 SIMDs x the clock GRBM_GUI_ACTIVE gives); the counters' own ratio
    (``valu_busy``: SQ_ACTIVE_INST_VALU x 4 / available SIMD-cycles) rides along -- it counts the cycles waves have a vector
    instruction IN FLIGHT, which overlap between waves (k_delete_ahead_live reads 1.04-1.07), so it is an indicator, not a
    ceiling fraction.  None when the device sources have changed since that run."""
    rec = pmc_record("k_mixed valu f64")
    if not rec:
        return None
    kname = rec.get("kernel", "k_mixed<double, false, 0>")     # (k_mixed3<double, false> for configs[4]'s loop: three rows per wave and trip)
    mix = _json_file("isa_counts.json").get("aot", {}).get(kname, {})
    clock = rec.get("clock_GHz")
    peak = valu_peak(clock)
    priced = rec["wave_instructions"] * mix.get("cycles_per_valu", 4.0) / rec["seconds"]
    return {"bound": "valu", "kernel": "%s (16 iterations of [Newton, ScatterIsotropic, Newton, ScatterDelete] per launch)" % kname,
            "achieved": priced, "peak": peak, "unit": "SIMD-cycles/s", "frac": priced / peak,
            "frac_at_4_waves_per_simd": rec["wave_instructions"] * mix.get("cycles_per_valu_at_4_waves", 4.3) / rec["seconds"] / peak,
            "cycles_per_wave_instruction": mix.get("cycles_per_valu"), "clock_GHz": clock,
            "valu_busy": rec["valu_busy"], "lane_util": rec["lane_utilisation"], "useful": rec["valu_busy"] * rec["lane_utilisation"],
            "wave_instructions_per_s": rec["wave_instructions_per_s"], "launches": rec["launches"],
            "source": rec["source"], "measured_at_commit": rec.get("commit"), "csrc_sha": rec.get("csrc_sha"),
            "note": "committed rocprofv3 --pmc run, not this process"}


def mixed_leg(dev, N, iterations=100, sample=2_000_000):
    """BASELINE configs[4], 1-GPU form: [Newton, ScatterIsotropic(A=n=1e-3), Newton, ScatterDelete] x 100 on ``N`` photons
    (1e8) in fp64 and in fp32, same Philox stream (tools/sweep_fp32.py): seconds per precision (16 iterations per launch,
    k_mixed, and one launch per light step) and the error figures of the fp32 run against the fp64 run."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sweep_fp32
    rep = sweep_fp32.sweep(dev, N, iterations, sample, 16)
    # The timed run takes no state samples: the checkpoints' downloads of the run above are outside its clock, but what they leave
    # behind is not -- they make the implicit dr / dv rows real, and every later compaction of that run moves them (fp64:
    # 0.096 s against 0.078 s for the same 100 iterations, tools/mixed_leg_split.py).  Same launches, same rows.
    fast = sweep_fp32.sweep(dev, N, iterations, 0, 16, timing_only=True)
    one = sweep_fp32.sweep(dev, N, iterations, 0, 1, timing_only=True)
    last = rep["checkpoints"][max(rep["checkpoints"])]
    work = rep["particle_steps_f64"]
    return {"workload": rep["workload"], "iterations": iterations, "iterations_per_launch": 16,
            "seconds_f64": fast["seconds_f64"], "seconds_f32": fast["seconds_f32"],
            "seconds_with_state_samples": {"f64": last["seconds_f64"], "f32": last["seconds_f32"],
                                           "note": "the run the fp32-vs-fp64 figures come from: state sampled after 1, 10 and 100 iterations"},
            "particle_steps": work, "value_f64": work / fast["seconds_f64"], "value_f32": rep["particle_steps_f32"] / fast["seconds_f32"],
            "unit": "particle-steps/s (a particle-step = one photon alive at the start of one Newton + light step)",
            "one_launch_per_light_step": {"seconds_f64": one["seconds_f64"], "seconds_f32": one["seconds_f32"]},
            "roofline": mixed_valu_record(),
            "fp32_vs_fp64": {str(k): {key: v[key] for key in ("decision_mismatch_rate", "identical_history_fraction", "median_rel_err_r",
                                                                "p99_rel_err_r", "p9999_rel_err_r", "max_rel_err_r", "fraction_above_1e-4",
                                                                "compared")}
                             for k, v in rep["checkpoints"].items()}}


def api_leg(args, prof):
    """The same workload through the public API (physicl_amd.Simulation + the reference's step classes), so that the
    Simulation / ObjectList / UpdateTimeStep / measure-step overhead is part of a measured number: particle-steps/s
    over sim.run_time (start() .. join()).  "default" = the constructor exactly as a script written against the
    reference calls it (physicl/__init__.py:405-418: no steps_per_launch keyword) -- the simulation decides by itself
    that ``exit`` can be evaluated ahead of a launch; then steps_per_launch 32 and 1 spelled out.  The runs are
    BASELINE configs[2]'s 500 passes (examples/variable_n_scattering.ipynb) unless --api-steps says otherwise.
    "delete_default" = BASELINE configs[1](ii) (test/test_light.py:52-59) at the delete leg's first size, default constructor."""
    import physicl_amd as phys
    import physicl_amd.light as light
    import physicl_amd.newton as newton
    N = int(args.photons)
    steps = args.api_steps if args.api_steps > 0 else (500 if N >= 10_000_000 else args.steps)
    out = {"what": "Simulation(...) with [UpdateTimeStep, NewtonianKinematicsStep, ScatterIsotropicStep, ScatterSignMeasureStep], "
                   "exit after %d passes; whole run incl. hipRTC lookup and terminate (creation of the photons is set-up)" % steps,
           "steps": steps}
    dt = np.double(prof["dt"])
    for name, kw in (("default", {}), ("steps_per_launch_32", {"steps_per_launch": 32}), ("steps_per_launch_1", {"steps_per_launch": 1})):
        sim = phys.Simulation(exit=lambda s: len(s.ts) >= steps, seed=args.seed, **kw)
        sim.add_objs(light.generate_photons_bulk(N, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9),
                                                 seed=args.seed))
        sim.add_step(0, phys.UpdateTimeStep(lambda s: dt))
        sim.add_step(1, newton.NewtonianKinematicsStep())
        sim.add_step(2, light.ScatterIsotropicStep(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True,
                                                   variable_n=True, variable_n_fn=prof["expr"]))
        m = light.ScatterSignMeasureStep(None, True)
        sim.add_step(3, m)
        sim._to_device()                       # creation of the photons is set-up, not stepping
        sim._dev.sync()
        sim.start()
        sim.join()
        if sim.error is not None:
            raise sim.error
        out[name] = {"value": N * len(sim.ts) / sim.run_time, "unit": "particle-steps/s", "steps": len(sim.ts),
                     "run_time_s": sim.run_time, "rows": len(m.data), "schedule": dict(sim.schedule), "note": sim.launch_note}
        sim.close(download=False)
    # the example's own four steps (examples/variable_n_scattering.ipynb:52-60): UpdateTimeStep, NewtonianKinematicsStep,
    # ScatterSphericalStep and TracePathMeasureStep -- the first 1000 photons traced (the notebook's population), worked out on
    # the device ahead of every launch (pcl_store_trace_ahead), default constructor
    sim = phys.Simulation(exit=lambda s: len(s.ts) >= steps, seed=args.seed)
    sim.add_objs(light.generate_photons_bulk(N, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=args.seed))
    sim.add_step(2, phys.UpdateTimeStep(lambda s: dt))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(3, light.ScatterSphericalStep(0.000000000000001, 0.0000000000000000001, wavelength_dep_scattering=True,
                                               variable_n=True, variable_n_fn=prof["expr"]))
    tp = light.TracePathMeasureStep(None)
    sim.add_step(0, tp)
    sim._to_device()
    sim._dev.sync()
    sim.start()
    sim.join()
    if sim.error is not None:
        raise sim.error
    n_tr = len(tp.data) - 1
    # the last traced position of every tracked photon is where the store has it (the trace is worked out ahead of the launches)
    last = np.array([np.asarray(tp.data[1 + j][-1], dtype=np.float64) for j in range(n_tr)])
    here = np.stack([sim._dev.download(f, n_tr, 0) for f in (0, 1, 2)], 1) if n_tr else last
    out["trace_default"] = {"value": N * len(sim.ts) / sim.run_time, "unit": "particle-steps/s", "steps": len(sim.ts),
                            "run_time_s": sim.run_time, "tracked": n_tr, "schedule": dict(sim.schedule), "note": sim.launch_note,
                            "trace_matches_store": bool(np.array_equal(last, here)),
                            "what": "[UpdateTimeStep, NewtonianKinematicsStep, ScatterSphericalStep, TracePathMeasureStep]: the "
                                    "step list of examples/variable_n_scattering.ipynb:52-60"}
    if not out["trace_default"]["trace_matches_store"]:
        raise RuntimeError("bench: the traced positions differ from the store's")
    sim.close(download=False)
    # the delete loop of test/test_light.py:52-59 with the constructor's defaults (exit: no objects left).  The whole run is
    # under a millisecond at 1e7 photons: five simulations, the median run time (every run listed; the first one of a
    # process also pays for Python's own first pass through the host layer and the library's first allocations of the path)
    Nd = int(float(args.delete_photons.split(",")[0]))
    runs = []
    for _ in range(5):
        sim = phys.Simulation(seed=args.seed)
        sim.add_objs(light.generate_photons_bulk(Nd, min=1.0, max=1.0, seed=args.seed))
        sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
        sim.add_step(1, newton.NewtonianKinematicsStep())
        sim.add_step(2, light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
        m = light.ScatterMeasureStep(None, True, [[1.0 / (1e-3 * 1e-3), np.nan, np.nan]])
        sim.add_step(3, m)
        sim._to_device()
        sim._dev.sync()
        sim.start()
        sim.join()
        if sim.error is not None:
            raise sim.error
        work = int(sum(int(np.asarray(row)[1]) for row in m.data)) + Nd - (int(np.asarray(m.data[-1])[1]) if m.data else 0)
        runs.append((sim.run_time, work, len(sim.ts), dict(sim.schedule), sim.launch_note))
        sim.close(download=False)
    rt, work, passes, schedule, note = sorted(runs, key=lambda r: r[0])[2]
    out["delete_default"] = {"photons": Nd, "value": work / rt, "unit": "particle-steps/s", "particle_steps": work,
                             "passes": passes, "run_time_s": rt, "run_times_s": [r[0] for r in runs], "timing": "median of 5 simulations",
                             "schedule": schedule, "note": note}
    return out


def cpu_baseline(dev, args, prof, gpu_rows=None):
    """The oracle's C/OpenMP port of the same step (newton + fused scatter + sign counters) on the first
    `cpu_photons` photons of the SAME initial workload, all host cores this process may use (BASELINE.md section 4
    item 3: 1e8 photons).  Checker code timed as a baseline: never part of the GPU path.
    When the port runs ALL of the run's photons (the default: 1e8 of 1e8) its rows [N, hits, xp, yp, zp] of steps 0, 1, 2 ...
    are the rows the GPU produced for the same launch indices in this very run (``gpu_rows``): compared, ``rows_match`` in
    the record, and a mismatch fails the bench -- whole-store parity at full size, every run."""
    from oracle import c_oracle as co
    if prof["c_profile"] is None:
        return {"value": None, "unit": "particle-steps/s", "cores": 0, "kind": "port",
                "sample": "C port implements the example profile only"}
    n = int(min(args.cpu_photons, args.photons))
    # same photons as the GPU run: regenerate the initial energies of ids [0, n) on the device and download them;
    # every other field of the initial state is a constant (r = dr = dv = 0, v = (c, 0, 0))
    dev.store_alloc(n)
    dev.fill_photons(n, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, args.seed)
    st = {"r": [np.zeros(n) for _ in range(3)], "v": [np.full(n, C_LIT), np.zeros(n), np.zeros(n)],
          "dr": [np.zeros(n) for _ in range(3)], "dv": [np.zeros(n) for _ in range(3)], "E": dev.download(12, n)}
    dev.store_free()
    co.set_threads(co.usable_cores())          # the box's CPU share, not every core of the host
    cores = co.threads()
    profile, pk, poff = prof["c_profile"]

    cpu_rows = []

    def step(k):
        co.newton(st, prof["dt"])
        hits = co.scatter_isotropic(st, prof["A_kernel"], prof["n_kernel"], C_LIT, H_LIT, 1, profile, pk, poff, args.seed, k,
                                    ids=None, id_base=0)
        sign = co.counters(st)
        cpu_rows.append([n, int(hits), int(sign[0]), int(sign[1]), int(sign[2])])

    step(0)                                   # warm-up (first touch of the pages) + calibration
    t0 = time.perf_counter()
    step(1)
    one = time.perf_counter() - t0
    steps = int(max(2, min(2000, args.cpu_seconds / max(one, 1e-4))))
    t0 = time.perf_counter()
    for k in range(2, 2 + steps):
        step(k)
    el = time.perf_counter() - t0
    rec = {"value": n * steps / el, "unit": "particle-steps/s", "cores": cores, "nproc": os.cpu_count(), "kind": "port",
           "sample": "%d photons x %d steps of the same workload (oracle/c/physicl_oracle.c, OpenMP, %d threads of %d "
                     "CPUs on the box, %.1f s)" % (n, steps, cores, os.cpu_count() or 0, el)}
    if gpu_rows is not None and n == int(args.photons):
        both = [k for k in range(len(cpu_rows)) if k in gpu_rows]
        bad = [k for k in both if list(gpu_rows[k]) != cpu_rows[k]]
        rec["rows_compared"] = len(both)
        rec["rows_match"] = (not bad) if both else None
        rec["rows_note"] = ("rows [N, hits, xp, yp, zp] of launch indices 0 .. %d: the C port over ALL %d photons against the rows the GPU "
                            "produced in this run" % (len(both) - 1, n))
        if bad:
            k = bad[0]
            raise RuntimeError("bench: the GPU's row of step %d %r differs from the CPU port's %r" % (k, list(gpu_rows[k]), cpu_rows[k]))
    return rec


def cpu_baseline_numpy(dev, args, prof):
    """Third CPU figure: the numpy-vectorised oracle (oracle/physicl_oracle.py) on the first 1e7 photons, 1 core
    (--cpu-photons bounds it for the small test runs)."""
    from oracle import physicl_oracle as orc
    n = int(min(10_000_000, args.photons, max(args.cpu_photons, 1000)))
    dev.store_alloc(n)
    dev.fill_photons(n, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, args.seed)
    st = {g: [dev.download(f, n) for f in fids] for g, fids in
          (("r", (0, 1, 2)), ("v", (3, 4, 5)), ("dr", (6, 7, 8)), ("dv", (9, 10, 11)))}
    st["E"], st["id"] = dev.download(12, n), np.arange(n, dtype=np.int64)
    dev.store_free()
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


def cpu_baseline_python(dev, args, prof, photons=10000):
    """Second CPU figure, for scale: the reference-SHAPED path (one Python object per photon, a Python loop
    per step, oracle/pyloop.py) on the first 1e4 / 1e5 photons of the same workload, 1 core -- the cost model of
    the reference's own CPU path (BASELINE.md section 2 measured 1.5e4..2.4e4 particle-steps/s for it)."""
    from oracle import pyloop
    if prof["c_profile"] is None:
        return None
    n = int(min(photons, args.photons, max(args.cpu_photons, 1000)))
    dev.store_alloc(n)
    dev.fill_photons(n, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, args.seed)
    E = dev.download(12, n)
    dev.store_free()
    value, steps, el = pyloop.time_steps(E, prof["dt"], prof["A_kernel"], prof["n_kernel"], True, prof["c_profile"], 3.0)
    return {"value": value, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": "%d photons x %d steps, per-object Python loops (oracle/pyloop.py), %.1f s" % (n, steps, el)}


def cpu_baseline_python_units(args, prof, photons=2000, seconds=3.0):
    """Fourth CPU figure: the reference's CPU path at its REAL cost -- per-object Python loops over objects whose fields are
    ``Measurement``s (the code-units ndarray subclass, physicl/__init__.py:11-291: every ``obj.v * sim.dt`` and ``obj.r += obj.dr``
    goes through its ``__array_ufunc__`` unit algebra), with the constants h and c as Measurements in the wavelength term
    (physicl/newton.py:14-16, physicl/light.py:335-350).  oracle/pyloop.py runs the same loops on bare ndarrays and is ~10x
    faster than the reference measured in the survey container (1.5e4 .. 2.4e4, 2.2e3 with the wavelength term: BASELINE.md
    section 2); this leg uses the build's own ``Measurement`` (bug-compatible with the reference's, tests/test_units_parity.py),
    so the figure quoted beside the GPU's is what a PhysiCL user's CPU run costs.  Host code only: nothing of it is on the GPU path."""
    import physicl_amd.light as light
    from physicl_amd.units import Measurement
    n = int(min(photons, args.photons))
    rs = np.random.RandomState(args.seed)
    e_lo, e_hi = light.E_from_wavelength(700e-9), light.E_from_wavelength(200e-9)
    objs = [light.PhotonObject(E=e_lo + (e_hi - e_lo) * rs.power(3), v=Measurement([light.c, 0, 0], "m**1 s**-1")) for _ in range(n)]
    dt = Measurement(np.double(prof["dt"]), "s**1")
    A, nn = prof["A_kernel"], prof["n_kernel"]
    np.random.seed(0)
    steps, t0 = 0, time.perf_counter()
    with np.errstate(all="ignore"):
        while True:
            for o in objs:                                     # NewtonianKinematicsStep.run          newton.py:14-16
                o.dr = o.v * dt
                o.r += o.dr
            for o in objs:                                     # ScatterIsotropicStep.__run_py        light.py:336-350
                norm = np.linalg.norm(o.dr)
                pcoll = nn * A * norm
                pcoll *= ((light.h * light.c) / o.E) ** -4
                if pcoll >= np.random.random():
                    phi, theta = np.random.random() * np.pi, np.random.random() * np.pi * 2
                    vold = o.v
                    o.v = np.array([light.c * np.sin(theta) * np.cos(phi), light.c * np.sin(theta) * np.sin(phi), light.c * np.cos(theta)])
                    o.dv = vold
                else:
                    o.dv = np.array([0, 0, 0])
            xp = sum(int(o.v[0] > 0) for o in objs)           # ScatterSignMeasureStep.run            light.py:423-426
            steps += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                break
    return {"value": n * steps / el, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": "%d PhotonObjects with Measurement fields x %d steps of [Newton, ScatterIsotropic (constant n, wavelength term), sign "
                      "count], per-object Python loops as physicl/newton.py:14-16 and physicl/light.py:336-350, %.1f s" % (n, steps, el)}


if __name__ == "__main__":
    main()
