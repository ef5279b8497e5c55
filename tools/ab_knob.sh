#!/bin/bash
# same-box A/B of one knob on the delete-until-empty run:  tools/ab_knob.sh PCL_AHEAD_LIVE 0 1
set -e
K=$1; shift
for rep in 1 2; do
 for v in "$@"; do
  for n in 1e8 1e7; do
    echo "== $K=$v photons $n"
    env $K=$v python tools/bench_delete_bodies.py --photons $n --reps 3 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_total'], '%.3g' % d['value'], d['kernels_ms'])
"
  done
 done
done
