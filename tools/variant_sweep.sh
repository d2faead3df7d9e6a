#!/bin/bash
# A/B the CSR-stream kernel builds on the level-0 500^3 operator (GPU box)
for v in base nt t4k nt_t4k t1k; do
  if [ $v = base ]; then unset AMGCORE_HIP_LIB; else export AMGCORE_HIP_LIB=$PWD/pyamg_amd/lib/variants/libamgcore_hip_$v.so; fi
  echo "== $v"; python tools/spmv_sweep.py ${1:-500} 2>&1 | grep -E "variant 1 chunk +(0|16|64) "
done
