#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]
# rocprofv3 kernel-trace stats + two PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950) of bench.py
set -o pipefail
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/trace_bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
echo done $TAG
