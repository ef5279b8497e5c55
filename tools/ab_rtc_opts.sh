# same-box A/B of extra hipRTC compile options on the driver's command (the K-step pass alone, no other legs), twice each:
#   tools/ab_rtc_opts.sh "<options A>" "<options B>" ...      ("" = as shipped)   -> prints value, blocks, forms
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
i=0
for rep in 1 2; do
  i=0
  for opts in "$@"; do
    i=$((i+1))
    PCL_RTC_DEFINE="$opts" PCL_BENCH_DETAIL=$O/rtcopt_${i}_$rep.json timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/rtcopt_${i}_$rep.line 2> /dev/null || { echo "failed: $opts"; continue; }
    python - "$opts" $O/rtcopt_${i}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(repr(sys.argv[1]), "%.4g" % d["value"], d["repeat_ms_per_step"], [list(b["forms"])[0][8:-3] for b in d["roofline"]["per_block"]], "tame %.4g" % d["tame"]["value"] if "tame" in d else "", flush=True)
PY
  done
done
