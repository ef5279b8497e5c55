# SQ counters of the delete-until-empty run's kernels at 1e8 photons (two passes: the counters do not fit one):
#   bash tools/prof_ahead.sh ; python tools/summarize_ahead_prof.py
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_ahead -o pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_delete_bodies.py --photons 1e8 --reps 1 --no-prof > $O/prof_ahead.json 2> $O/prof_ahead.err || { tail -5 $O/prof_ahead.err; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/prof_ahead2 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_delete_bodies.py --photons 1e8 --reps 1 --no-prof > $O/prof_ahead2.json 2> $O/prof_ahead2.err || { tail -5 $O/prof_ahead2.err; exit 1; }
echo done
