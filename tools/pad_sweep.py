"""Row-pitch experiment: fill rate and one-step kernel rate for several PCL_ROW_PAD values, with physically contiguous
slabs (PCL_SLAB_CONTIG=1: the slow placement) and with default allocations (three stores side by side each)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
from bench import PROFILES, C_LIT, H_LIT
from physicl_amd import _hip
N = 100_000_000
prof = PROFILES["example"]
sc = lambda k: dict(A=prof["A_kernel"], n=prof["n_kernel"], flags=3, c=C_LIT, h=H_LIT, n_expr=prof["expr"], rng_mode=_hip.RNG_PHILOX, seed=7, step=k)
devs = [_hip.Device(0) for _ in range(3)]
out = []
for d in devs:
    d.store_alloc(N)
for d in devs:
    d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 7)
    d.timer_start()
    for k in range(4):
        d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, 7)
    f = 104.0 * N / (d.timer_stop() / 4) / 1e9
    for k in range(3):
        d.step_fused(prof["dt"], sc(k), (), lazy=True)
    d.timer_start()
    for k in range(10):
        d.step_fused(prof["dt"], sc(3 + k), None, sync=False, lazy=True)
    ms = d.timer_stop() / 10
    out.append("%%.2f/%%.3f" %% (f, 104.0 * N / (ms * 1e-3) / 8e12))
print(" ".join(out))
''' % ROOT
pads = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,256,512,1024,2048,4096,8192".split(","))]
for contig in (1, 0):
    for pad in pads:
        env = dict(os.environ, PCL_ROW_PAD=str(pad), PCL_POOL_GB="0")
        if contig:
            env["PCL_SLAB_CONTIG"] = "1"
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        print("contig=%d pad=%5d B: fill TB/s / one-step frac of 3 stores: %s %s" % (contig, pad, out.stdout.decode().strip(), out.stderr.decode().strip()[-200:] if out.returncode else ""), flush=True)
