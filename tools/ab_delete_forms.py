"""Same-process-sequence A/B of the K-step delete pass formulations (PCL_MULTI_FORM is read once per process, so each case
is a child process): prints pass-1 kernel ms and particle-steps/s per (K, form), two rounds."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
forms = sys.argv[1].split(",") if len(sys.argv) > 1 else ["ring", "ring1", "ring2"]
Ks = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["8", "16", "48"])]
for rnd in range(2):
    for K in Ks:
        row = []
        for f in forms:
            env = dict(os.environ, PCL_MULTI_FORM=f)
            out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "bench_delete.py"), "--photons", "1e8", "--steps", "48",
                                           "--mode", "multi", "--steps-per-launch", str(K)], env=env, timeout=300)
            d = json.loads(out.decode().strip().splitlines()[-1])
            row.append("%s %.3f ms %.3g" % (f, d["kernels_total_ms"]["k_delete_mask"], d["particle_steps_per_s"]))
        print("K=%d: %s" % (K, " | ".join(row)), flush=True)
