#!/bin/bash
# timing experiment: what the K-step kernel's compute is made of (results are WRONG under an ablation)
for d in "" "-DPCL_ABLATE_PHILOX" "-DPCL_ABLATE_TRIG" "-DPCL_ABLATE_POW"; do
  echo "== ${d:-baseline}"
  PCL_RTC_DEFINE="$d" timeout -k 10 120 python tools/bench_multi.py --ks 16 --reps 3 || exit 1
done
