# same-box A/B of -mllvm -disable-machine-licm: (A) as built, (B) the ahead-of-time library built with it
# (physicl_amd/_lib/libphysicl_hip_nolicm.so, built by hand), (C) the hipRTC specialisations compiled with it (PCL_RTC_DEFINE)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
L=physicl_amd/_lib
cp $L/libphysicl_hip.so $L/keep.so
trap 'cp $L/keep.so $L/libphysicl_hip.so' EXIT    # (a timeout or a failing run must not leave the other build installed)
run() { PCL_BENCH_DETAIL=$O/licm_$1.json timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/licm_$1.line 2> $O/licm_$1.err; }
run A && cp $L/libphysicl_hip_nolicm.so $L/libphysicl_hip.so && run B && cp $L/keep.so $L/libphysicl_hip.so && PCL_RTC_DEFINE="-mllvm -disable-machine-licm" run C
rc=$?
cp $L/keep.so $L/libphysicl_hip.so
tail -3 $O/licm_C.err
exit $rc
