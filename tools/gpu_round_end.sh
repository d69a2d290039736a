#!/bin/bash
# round-end evidence: full GPU test suite, smoke, bench line, rocprofv3 kernel stats of the bench run
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round_end.sh r02'
tag=${1:-rXX}
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/${tag}_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/${tag}_gpu_tests.log | tail -2
timeout 200 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; tail -1 gpurun_out/${tag}_smoke.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; head -c 400 gpurun_out/${tag}_bench.json; echo
cd /tmp; rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_prof_bench.log 2>&1
cd $R
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
python tools/summarize_trace.py $(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1) gpurun_out/${tag}_bench_kernel_by_grid.csv
head -8 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-160
