#!/bin/bash
# Round-2 measurement pass on the GPU box (run from the repo root): the bench line, the rocprofv3 kernel trace of the
# same command, and the FETCH_SIZE / WRITE_SIZE PMC passes of the level-0 kernel (separate runs, counters only).
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python bench.py > $O/r02_bench500.json 2> $O/r02_bench500.err && tail -c 600 $O/r02_bench500.json && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-value-index > $O/r02_bench500_under_rocprof.json 2> $O/r02_rocprof.err && \
python tools/trace_cycle.py $(ls $O/prof_r02/*/*kernel_trace.csv | head -1) 10 > $O/r02_bench500_timed_region.txt && \
cp $(ls $O/prof_r02/*/*kernel_stats.csv | head -1) $O/r02_bench500_kernel_stats.csv && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_r02 -- python3 tools/pmc_spmv.py 500 1 32 1 > $O/r02_pmc_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_r02 -- python3 tools/pmc_spmv.py 500 1 32 1 > $O/r02_pmc_write.log 2>&1 && \
cp $(ls $O/pmc_fetch_r02/*/*counter_collection.csv | head -1) $O/r02_pmc_fetch_stencil_500_counter_collection.csv && \
cp $(ls $O/pmc_write_r02/*/*counter_collection.csv | head -1) $O/r02_pmc_write_stencil_500_counter_collection.csv && \
python tools/pmc_summary.py $O/r02_pmc_fetch_stencil_500_counter_collection.csv $O/r02_pmc_write_stencil_500_counter_collection.csv 500 $O/r02_pmc_summary.json $O/r02_bench500.json > /dev/null && \
head -30 $O/r02_bench500_timed_region.txt
