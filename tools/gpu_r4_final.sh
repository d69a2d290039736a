#!/bin/bash
# round-end evidence: GPU suite, the default bench line, the headline-only kernel trace, smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final6
export MMREC_TEST_OBSERVED=$PWD/gpurun_out/final6/observed.tsv
rm -f $MMREC_TEST_OBSERVED
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final6/gpu_suite.log 2>&1; echo rc=$? >> gpurun_out/final6/gpu_suite.log)
(timeout 600 python bench.py > gpurun_out/final6/bench_line.json 2> gpurun_out/final6/bench.err; echo bench rc=$?)
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final6/headline_trace -- python bench.py --headline-only > gpurun_out/final6/bench_headline_only_line.json 2> gpurun_out/final6/bench_headline.err; echo headline rc=$?)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final6/smoke.log 2>&1; echo smoke rc=$?; tail -1 gpurun_out/final6/smoke.log)
f=$(find gpurun_out/final6/headline_trace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/final6/headline_kernel_stats.csv; head -5 gpurun_out/final6/headline_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/final6/headline_trace
grep -n "passed\|failed" gpurun_out/final6/gpu_suite.log | tail -2; head -c 300 gpurun_out/final6/bench_line.json
