#!/bin/bash
# round 4, run y: row-lazy Adam catch-up with the settled-parameter path: bitwise tests + the config-5 step in steady state
# (A/B against a build without the path: tools/probe_libs/libmmrec_adam_nosettled.so, -DMMREC_ADAM_NO_SETTLED=1)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_hip_parity.py tests/test_models_gpu.py -q -x -k "lazy" ) > gpurun_out/r04_y_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^E  " gpurun_out/r04_y_tests.log | head
export MMREC_C5_ROOT=/tmp/mmrec_c5_root MMREC_C5_PLAIN_ONLY=1 MMREC_C5_LATE_STEPS=630
mkdir -p $MMREC_C5_ROOT
( time timeout 900 python tools/run_c5_plugin.py 40 ) > gpurun_out/r04_y_c5_steady_settled.log 2>&1
grep -n "ms/step" gpurun_out/r04_y_c5_steady_settled.log
( time MMREC_HIP_LIB=$PWD/tools/probe_libs/libmmrec_adam_nosettled.so timeout 900 python tools/run_c5_plugin.py 40 ) > gpurun_out/r04_y_c5_steady_full_replay.log 2>&1
grep -n "ms/step" gpurun_out/r04_y_c5_steady_full_replay.log
