#!/bin/bash
# A/B the CSR-stream kernel builds on the level-0 operator (GPU box): tools/variant_sweep.sh GRID name1 name2 ...
g=${1:-500}; shift
for v in base "$@"; do
  if [ $v = base ]; then unset AMGCORE_HIP_LIB; else export AMGCORE_HIP_LIB=$PWD/pyamg_amd/lib/variants/libamgcore_hip_$v.so; fi
  echo "== $v"; python tools/spmv_sweep.py $g 2>&1 | grep -E "variant 1 chunk +(0|16|64) "
done
