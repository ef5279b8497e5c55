#!/bin/bash
# kernel trace of delete-until-empty, one call per loop body (alive-mask path): every dispatch with its duration
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_delbodies; mkdir -p $OUT
N=${1:-1e8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/tools/bench_delete_bodies.py --photons $N --reps 1 > $OUT/kt.json 2> $OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
cd $REPO
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_delbodies/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run of the tool: dispatches after the last k_fill_photons
last = max(i for i, r in enumerate(rows) if "k_fill_photons" in r["Kernel_Name"])
prev_end = None
out = []
for r in rows[last + 1:last + 60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    out.append("%-44s grid %8s  dur %9.1f us  gap %7.1f us" % (r["Kernel_Name"][:44], r.get("Grid_Size_X", r.get("Grid_Size", "?")), (e - s) / 1e3, gap))
open("gpurun_out/prof_delbodies/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
PY
