# round-end evidence: GPU suite, the default bench line, the headline-only kernel trace, top-K counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/gpu_suite.log 2>&1; echo rc=$? >> gpurun_out/final/gpu_suite.log)
(timeout 600 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err; echo bench rc=$?)
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/headline_trace -- python bench.py --headline-only > gpurun_out/final/bench_headline_only_line.json 2> gpurun_out/final/bench_headline.err; echo headline rc=$?)
(timeout 800 python tools/pmc_kernels.py topk gpurun_out/final/topk_pmc > gpurun_out/final/topk_pmc.log 2>&1; echo pmc rc=$?)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; echo smoke rc=$?; tail -1 gpurun_out/final/smoke.log)
grep -n "passed\|failed" gpurun_out/final/gpu_suite.log | tail -2
