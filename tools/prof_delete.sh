#!/bin/bash
# counters of the K-step delete pass (timing experiment)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_delete; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/tools/bench_delete.py --photons 1e8 --steps 8 --mode multi "$@" > $OUT/sq.json 2> $OUT/sq.err || { tail -5 $OUT/sq.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o pmc -- python3 $REPO/tools/bench_delete.py --photons 1e8 --steps 8 --mode multi "$@" > $OUT/f.json 2> $OUT/f.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o pmc -- python3 $REPO/tools/bench_delete.py --photons 1e8 --steps 8 --mode multi "$@" > $OUT/w.json 2> $OUT/w.err || exit 1
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for d in ("sq", "f", "w"):
    f = glob.glob("gpurun_out/prof_delete/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "mask_multi" in k or "flag_mask" in k or "compact_count" in k or "tile_scan" in k:
            print(d, k, {a: b[-1] for a, b in v.items()})
PY
