#!/bin/bash
# VALU instructions of the K-step kernel at constant hit probability p (timing experiment)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_multi_p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for p in ${PS:-0 0.1 0.3 0.5 1}; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/p$p -o pmc -- python3 $REPO/tools/bench_multi.py --ks 16 --reps 2 --plain-p $p > $OUT/p$p.json 2> $OUT/p$p.err || { tail -5 $OUT/p$p.err; exit 1; }
  python3 - $OUT/p$p $p <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.Counter()
for r in csv.DictReader(open(f)):
    if "multi" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
print("p", sys.argv[2], "VALU per wave-iteration", agg["SQ_INSTS_VALU"] / (3 * 16 * 1e8 / 128), dict(agg))
PY
done
