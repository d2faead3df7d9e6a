#!/bin/bash
# same-box A/B of the C5 block smoothers: tools/c5_ab.sh GRID "variant wpc [xcd]" ...  (variant d = the regular library;
# xcd = AMG_FLOW_XCD_BLOCK, consecutive tasks on one XCD)
GRID=$1; shift
export C5_GRID=$GRID C5_ORACLE=0 C5_STEPS=5
for v in "$@"; do
  set -- $v
  if [ "$1" != d ]; then export AMGCORE_HIP_LIB=$PWD/tools/_bin/libamg_$1.so; else unset AMGCORE_HIP_LIB; fi
  if [ -n "$3" ]; then export AMG_FLOW_XCD_BLOCK=$3; else unset AMG_FLOW_XCD_BLOCK; fi
  echo "== $v"
  AMG_FLOW_WPC=$2 timeout -k 10 300 python tools/bench_c5.py 2>gpurun_out/c5_ab.err | head -1 > gpurun_out/c5_ab.json
  python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/c5_ab.json").readline())
    print({k: d[k] for k in ("ms_per_step", "level0_smoother_ms", "level0_smoother_GBs", "coarse_smoother_ms")})
except Exception as e:
    print("failed:", e); print(open("gpurun_out/c5_ab.err").read()[-1500:])
PY
done
