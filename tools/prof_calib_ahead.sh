# SQ_INSTS_VALU of the k_delete_ahead_live launches of tools/calib_ahead.py (one per case, in order):
#   bash tools/prof_calib_ahead.sh ; python tools/summarize_calib_ahead.py
cd /tmp && export TMPDIR=/tmp
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/prof_calib_ahead
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/prof_calib_ahead -o pmc -- python3 $GRAFT_REPO_ROOT/tools/calib_ahead.py > $O/calib_ahead.jsonl 2> $O/calib_ahead.err || { tail -5 $O/calib_ahead.err; exit 1; }
cat $O/calib_ahead.jsonl
