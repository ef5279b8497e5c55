# SQ_INSTS_VALU of the constant-n K-step launches of tools/bench_iso.py (BASELINE configs[1](i), 1e7 photons), once on the
# 128-photons-per-wave form with |v| dt thresholds as doubles (PCL_MULTI_NQ3=0: k_multi<double, false, 0>) and once on this round's
# default (k_multi3_e0: 192 per wave, integer thresholds):
#   bash tools/prof_constant_n.sh ; python tools/summarize_constant_n.py   -> profiles/r06_constant_n_pmc.md
cd /tmp && export TMPDIR=/tmp
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$GRAFT_REPO_ROOT/gpurun_out
for q in 0 default; do
  rm -rf $O/prof_constant_n_$q
  if [ $q = 0 ]; then export PCL_MULTI_NQ3=0; else unset PCL_MULTI_NQ3; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/prof_constant_n_$q -o pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_iso.py 1e7 > $O/constant_n_$q.json 2> $O/constant_n_$q.err || { tail -5 $O/constant_n_$q.err; exit 1; }
  cat $O/constant_n_$q.json
done
