#!/bin/bash
# Round-3 Gauss-Seidel measurement pass on the GPU box (run from the repo root): per-level costs and cycle times of the
# reference's default 3-D setup, rocprofv3 kernel tables of the 128^3 cycle and of the C5 block Gauss-Seidel cycle,
# the randomised cross-check of all sweep paths.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
{ python tools/gs_level_cost.py 128; python tools/gs_level_cost.py 64; python tools/gs3d_trace.py 64; python tools/gs3d_trace.py 128; python tools/gs3d_trace.py 200; } 2>&1 | grep -v amdgpu.ids > $O/r03_gs3d.txt && \
rocprofv3 --kernel-trace --output-format csv -d $O/prof_gs3d_r03 -- python3 tools/gs3d_trace.py 128 > $O/r03_gs3d_rocprof.log 2>&1 && \
python tools/kernel_table.py $O/prof_gs3d_r03 12 > $O/r03_gs3d_kernels.txt && \
{ for s in 0 1 2; do python tools/stress_gs.py $s 60; done; } 2>&1 | grep -v amdgpu.ids > $O/r03_stress_gs.txt && \
C5_GRID=345 C5_ORACLE=0 C5_STEPS=5 rocprofv3 --kernel-trace --output-format csv -d $O/prof_c5_r03 -- python3 tools/bench_c5.py > $O/r03_c5_under_rocprof.json 2> $O/r03_c5_rocprof.err && \
python tools/kernel_table.py $O/prof_c5_r03 12 > $O/r03_c5_kernels.txt && \
cat $O/r03_gs3d.txt && head -16 $O/r03_gs3d_kernels.txt && tail -1 $O/r03_stress_gs.txt && head -16 $O/r03_c5_kernels.txt
