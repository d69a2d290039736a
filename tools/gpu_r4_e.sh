#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python tools/dslice_probe.py --out gpurun_out/r04_dslice_probe2.json ) > gpurun_out/r04_dslice_probe2.log 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_a.json ) 2> gpurun_out/r04_bench_a.err
( time timeout 600 python bench.py --steps 20 --warmup 5 --force-dist --layout dslice --no-cpu-baseline > gpurun_out/r04_bench_fd_dslice.json ) 2> gpurun_out/r04_bench_fd_dslice.err
tail -6 gpurun_out/r04_dslice_probe2.log; tail -3 gpurun_out/r04_bench_a.err; head -c 1500 gpurun_out/r04_bench_a.json; echo; tail -3 gpurun_out/r04_bench_fd_dslice.err; head -c 600 gpurun_out/r04_bench_fd_dslice.json
