# same-box A/B of two builds of the library (physicl_amd/_lib/libphysicl_hip_old.so from tools/build_old_lib.sh against the tree's) on the
# driver's command, K-step pass and the tame expression only: tools/ab_lib_bench.sh   -> value, blocks, forms, tame
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=physicl_amd/_lib
cp $L/libphysicl_hip.so $L/new.so
# whatever ends this script (a timeout, Ctrl-C, a failing run), the tree's own build is what stays installed: later tests
# and benches must never measure the old library under the new sources' hash
trap 'cp $L/new.so $L/libphysicl_hip.so' EXIT
for rep in 1 2; do
  for which in old new; do
    cp $L/${which/old/libphysicl_hip_old}.so $L/libphysicl_hip.so 2>/dev/null || cp $L/new.so $L/libphysicl_hip.so
    PCL_BENCH_DETAIL=gpurun_out/ablib_${which}_$rep.json timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --delete-photons 1e7 > /dev/null 2>&1
    python - $which gpurun_out/ablib_${which}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "%.4g" % d["value"], d["repeat_ms_per_step"], [list(b["forms"])[0][8:-3] for b in d["roofline"]["per_block"]],
      "tame %.4g %s" % (d["tame"]["value"], d["tame"]["repeat_ms_per_step"]), "api %.4g" % d["api"]["default"]["value"], "mixed f64 %.4g f32 %.4g" % (d["mixed"]["value_f64"], d["mixed"]["value_f32"]), flush=True)
PY
  done
done
cp $L/new.so $L/libphysicl_hip.so
