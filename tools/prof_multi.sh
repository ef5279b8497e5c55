#!/bin/bash
# SQ counters of the K-step kernel (one pass, 8 SQ slots + GRBM): where its wave cycles go
set -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_multi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -o pmc -- python3 $REPO/tools/bench_multi.py --ks ${1:-16} --reps 2 "${@:2}" > $OUT/sq1.json 2> $OUT/sq1.err || { tail -5 $OUT/sq1.err; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/bench_multi.py --ks ${1:-16} --reps 2 "${@:2}" > $OUT/sq2.json 2> $OUT/sq2.err || { tail -5 $OUT/sq2.err; exit 1; }
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for d in ("sq1", "sq2"):
    f = glob.glob("gpurun_out/prof_multi/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f:
        print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "multi" in k:
            print(d, k, {a: b for a, b in v.items()})
PY
