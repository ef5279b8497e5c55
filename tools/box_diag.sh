#!/bin/bash
# Box diagnostics next to a bench run: clocks / partition modes / bench line. Output -> gpurun_out/box_diag_<tag>.log
tag=${1:-x}
out=gpurun_out/box_diag_$tag.log
mkdir -p gpurun_out
{
  rocm-smi --showcomputepartition --showmemorypartition --showclocks --showperflevel 2>&1 | grep -v "^$" | head -40
  python bench.py --no-cpu-baseline --steps 30 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  rocm-smi --showclocks 2>&1 | grep -i "mclk\|sclk\|fclk"
} > $out 2>&1
cat $out
