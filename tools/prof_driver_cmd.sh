#!/bin/bash
# usage: tools/prof_driver_cmd.sh <tag> [bench args...]      (default args: the driver's command  --gpus 1 --steps 20 --warmup 5)
# rocprofv3 of EXACTLY the command the driver runs at round end: one --kernel-trace --stats pass, then three PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; SQ counters), each with --kernel-trace only (the pool refuses
# --pmc together with the sys/hip/hsa trace domains).  Output under gpurun_out/prof_<tag>/; summarise with
# tools/summarize_driver_prof.py <tag>.  bench.py prints the contract line only; each pass leaves the run's full record
# (PCL_BENCH_DETAIL) beside its CSVs, which is what the summary reads.
set -o pipefail
TAG=$1; shift
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then ARGS=(--gpus 1 --steps 20 --warmup 5); fi
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
echo "${ARGS[@]}" > $OUT/bench_args.txt
cd /tmp && export TMPDIR=/tmp
PCL_BENCH_DETAIL=$OUT/trace_detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py "${ARGS[@]}" > $OUT/trace_bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
echo "trace pass done"
PCL_BENCH_DETAIL=$OUT/pmc_fetch_detail.json rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "${ARGS[@]}" > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
echo "fetch pass done"
PCL_BENCH_DETAIL=$OUT/pmc_write_detail.json rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "${ARGS[@]}" > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
echo "write pass done"
PCL_BENCH_DETAIL=$OUT/pmc_sq_detail.json rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "${ARGS[@]}" > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || { tail -5 $OUT/pmc_sq.err; exit 1; }
echo done $TAG
