#!/bin/bash
# The round's closing GPU runs, in the order the records depend on each other (run from the repo root on the GPU box):
#   tools/final_run.sh profiles    the -m gpu suite, then rocprofv3 of the driver's command and of bench.py's default command and
#                                  the calibration launches of k_delete_ahead_live -> summarise HERE afterwards
#                                  (tools/summarize_driver_prof.py r06_driver_cmd / r06_default_cmd, tools/summarize_calib_ahead.py)
#   tools/final_run.sh lines       the three bench lines (driver's command, default, two ranks over gloo on one card) with their
#                                  full records, quoting the counter records summarised above
O=gpurun_out
case "$1" in
profiles)
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r06_t_all.log 2>&1; rc=$?; tail -3 $O/r06_t_all.log
  [ $rc -eq 0 ] || exit $rc
  bash tools/prof_driver_cmd.sh r06_driver_cmd && bash tools/prof_driver_cmd.sh r06_default_cmd --gpus 1 && bash tools/prof_calib_ahead.sh > $O/r06_calib.log 2>&1
  ;;
lines)
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_line_driver.json 2>/dev/null && cp bench_detail.json $O/r06_detail_driver.json &&
  python bench.py > $O/r06_line_default.json 2>/dev/null && cp bench_detail.json $O/r06_detail_default.json &&
  python bench.py --gpus 2 --backend gloo --device 0 --photons 5e7 --steps 20 --warmup 5 > $O/r06_line_2rank.json 2>/dev/null && cp bench_detail.json $O/r06_detail_2rank.json
  rc=$?; wc -c $O/r06_line_*.json; exit $rc
  ;;
*) echo "usage: tools/final_run.sh profiles|lines"; exit 2;;
esac
