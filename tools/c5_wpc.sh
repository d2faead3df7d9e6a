#!/bin/bash
# level-0 block Gauss-Seidel of C5 for several sizes and resident-wave caps: tools/c5_wpc.sh "345 360" "2 4 8"
export C5_ORACLE=0 C5_STEPS=3
for g in $1; do for w in $2; do
  echo "== grid $g wpc $w"
  C5_GRID=$g AMG_FLOW_WPC=$w timeout -k 10 400 python tools/bench_c5.py 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('ms_per_step','level0_smoother_ms','level0_smoother_GBs','coarse_smoother_ms')})"
done; done
