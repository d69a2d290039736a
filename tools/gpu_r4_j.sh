#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python tools/c5_sliced_step.py 40 ) > gpurun_out/r04_c5_sliced_step.log 2>&1
grep "c5-sliced\|Error\|error\|failed" gpurun_out/r04_c5_sliced_step.log | tail -12
