#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export MMREC_TEST_OBSERVED=$PWD/gpurun_out/r04_observed_h.tsv
rm -f $MMREC_TEST_OBSERVED
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > gpurun_out/r04_gpu_suite.log 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_b.json ) 2> gpurun_out/r04_bench_b.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1
tail -5 gpurun_out/r04_gpu_suite.log; tail -3 gpurun_out/r04_bench_b.err; tail -2 gpurun_out/r04_smoke.log; head -c 400 gpurun_out/r04_bench_b.json
