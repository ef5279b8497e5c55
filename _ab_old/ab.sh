#!/bin/bash
# A/B: old per-field SoA (this dir) vs tiled slab (repo root), alternating, same box
for i in 1 2 3; do
  (cd /root/repo/_ab_old && timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('old  ', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['counters_last_step'])") || exit 1
  (cd /root/repo && timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tiled', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['counters_last_step'])") || exit 1
done
