#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]
# rocprofv3 kernel-trace stats + three PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950; SQ counters) of bench.py
set -o pipefail
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/trace_bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
# SQ counters (one pass, 8 SQ slots + GRBM): VALU instructions / busy cycles / lane utilisation of the K-step kernel
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || { tail -5 $OUT/pmc_sq.err; exit 1; }
echo done $TAG
