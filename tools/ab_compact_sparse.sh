set -e
for n in 1e8 1e7; do
 for sp in 0 256 0 256; do
  echo "== photons $n PCL_COMPACT_SPARSE=$sp"
  PCL_COMPACT_SPARSE=$sp python tools/bench_delete_bodies.py --photons $n --reps 3 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_total'], '%.3g' % d['value'], d['kernels_ms'])
"
 done
done
