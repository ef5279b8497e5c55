#!/bin/bash
# same-box A/B of two builds of the library: physicl_amd/_lib/libphysicl_hip_old.so (built by hand from another commit)
# against the current one.  Default workload: the delete-until-empty run; or  tools/ab_lib.sh <command ...>  (its output is shown)
set -e
L=physicl_amd/_lib
cp $L/libphysicl_hip.so $L/new.so
run() {
  if [ $# -gt 1 ]; then shift; echo "== $TAG"; "$@"; return; fi
  for n in 1e8 1e7; do
    echo "== $1 photons $n"
    python tools/bench_delete_bodies.py --photons $n --reps 3 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_total'], '%.3g' % d['value'], d['kernels_ms'])
"
  done
}
for rep in 1 2; do
  cp $L/libphysicl_hip_old.so $L/libphysicl_hip.so; TAG=old run old "$@"
  cp $L/new.so $L/libphysicl_hip.so; TAG=new run new "$@"
done
