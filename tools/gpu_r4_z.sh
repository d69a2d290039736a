#!/bin/bash
# round 4, run z: the settled-parameter replay later in a run (2400 steps in: closer to the steady state of the gaps)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp MMREC_C5_ROOT=/tmp/mmrec_c5_root MMREC_C5_PLAIN_ONLY=1 MMREC_C5_LATE_STEPS=2400
mkdir -p $MMREC_C5_ROOT
( timeout 170 python tools/run_c5_plugin.py 40 ) 2>&1 | grep "ms/step" > gpurun_out/r04_z_settled.log
( MMREC_HIP_LIB=$PWD/tools/probe_libs/libmmrec_adam_nosettled.so timeout 120 python tools/run_c5_plugin.py 40 ) 2>&1 | grep "ms/step" > gpurun_out/r04_z_full_replay.log
cat gpurun_out/r04_z_settled.log; echo ----; cat gpurun_out/r04_z_full_replay.log
