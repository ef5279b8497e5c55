#!/bin/bash
# per-dispatch evidence for the compactions of delete-until-empty, one call per loop body: a kernel-trace pass and two PMC
# passes (FETCH_SIZE, WRITE_SIZE; separate, --kernel-trace only) of tools/bench_delete_bodies.py; summarise with
# tools/summarize_compact_prof.py.   usage: tools/prof_compact.sh [photons]
set -o pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_compact; mkdir -p $OUT
N=${1:-1e8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/bench_delete_bodies.py --photons $N --reps 1 --no-prof > $OUT/trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
echo trace done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/tools/bench_delete_bodies.py --photons $N --reps 1 --no-prof > $OUT/fetch.json 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/tools/bench_delete_bodies.py --photons $N --reps 1 --no-prof > $OUT/write.json 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
echo write done
cd $REPO
find $OUT -name "*.csv" | head -20
