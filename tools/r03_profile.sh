#!/bin/bash
# Round-3 measurement pass on the GPU box (run from the repo root): the bench line of the metric configuration, the
# rocprofv3 kernel trace of the same command, the FETCH_SIZE / WRITE_SIZE PMC passes of the level-0, level-1 and transfer kernels (tools/pmc_ops.py;
# separate runs, counters only; first, so that the bench line can cite them), and the bench lines of the other BASELINE configurations.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_r03 -- python3 tools/pmc_ops.py 500 > $O/r03_pmc_ops_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_r03 -- python3 tools/pmc_ops.py 500 > $O/r03_pmc_ops_write.log 2>&1 && \
{ F=$(ls $O/pmc_fetch_r03/*/*counter_collection.csv | head -1); head -1 $F; grep 'stencil\|sell_kernel<\|csr_stream_kernel<0, 1>\|sumsq_stage1' $F; } > $O/r03_pmc_fetch_ops_500_counter_collection.csv && \
{ F=$(ls $O/pmc_write_r03/*/*counter_collection.csv | head -1); head -1 $F; grep 'stencil\|sell_kernel<\|csr_stream_kernel<0, 1>\|sumsq_stage1' $F; } > $O/r03_pmc_write_ops_500_counter_collection.csv && \
python tools/pmc_ops.py --summary $O/r03_pmc_fetch_ops_500_counter_collection.csv $O/r03_pmc_write_ops_500_counter_collection.csv $O/r03_pmc_ops_fetch.log $O/r03_pmc_summary.json > /dev/null && \
grep -v amdgpu.ids $O/r03_pmc_ops_fetch.log | grep -v '^{' > $O/r03_pmc_ops.txt && \
cp $O/r03_pmc_summary.json profiles/r03_pmc_summary.json && \
python bench.py > $O/r03_bench500.json 2> $O/r03_bench500.err && tail -c 400 $O/r03_bench500.json && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-value-index > $O/r03_bench500_under_rocprof.json 2> $O/r03_rocprof.err && \
python tools/trace_cycle.py $(ls $O/prof_r03/*/*kernel_trace.csv | head -1) 10 > $O/r03_bench500_timed_region.txt && \
cp $(ls $O/prof_r03/*/*kernel_stats.csv | head -1) $O/r03_bench500_kernel_stats.csv && \
head -24 $O/r03_bench500_timed_region.txt && \
python bench.py --config C1 > $O/r03_bench_C1.json 2> $O/r03_bench_C1.err && \
python bench.py --config C2 > $O/r03_bench_C2.json 2> $O/r03_bench_C2.err && \
python bench.py --config C5 --steps 20 > $O/r03_bench_C5.json 2> $O/r03_bench_C5.err && \
python -c "
import json
for c in ('C1','C2','C5'):
    d=json.load(open('$O/r03_bench_%s.json'%c)); r=d['roofline']; print(c, d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['ms_per_launch'], d['cpu_baseline']['first_step_iterate_bit_identical_to_gpu'])
"
