# same-box A/B of the K-step forms' waves per SIMD (amdgpu_waves_per_eu), the driver's command without the other legs, twice each:
#   tools/ab_occupancy.sh        -> gpurun_out/occ_<tag>_<rep>.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
run() { PCL_BENCH_DETAIL=$O/occ_$1.json timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/occ_$1.line 2> /dev/null; }
for rep in 1 2; do
  run A_$rep &&
  PCL_RTC_DEFINE="-DPCL_MULTI2_ATTR=__attribute__((amdgpu_waves_per_eu(5,5)))" run B_$rep &&
  PCL_RTC_DEFINE="-DPCL_MULTI2_ATTR=__attribute__((amdgpu_waves_per_eu(6,6)))" run C_$rep || exit 1
done
python - <<'PY'
import json
for t in "ABC":
    for rep in (1, 2):
        d = json.load(open("gpurun_out/occ_%s_%d.json" % (t, rep)))
        print(t, rep, "%.4g" % d["value"], d["repeat_ms_per_step"], [list(b["forms"])[0][8:-3] for b in d["roofline"]["per_block"]])
PY
