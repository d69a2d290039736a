#!/bin/bash
# round 4, run q: sampled scoring on column slices (mmrec_bpr_dots_f32 / _loss_from_dots_f32) + the sliced plugin's step cost
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "sliced_shared_user_bpr or sharded_freedom_plugin_rccl or shared_user_bpr or bpr_variants" ) > gpurun_out/r04_q_tests.log 2>&1
tail -5 gpurun_out/r04_q_tests.log
( time MMREC_C5S_ONLY=Sliced timeout 600 python tools/c5_sliced_step.py 40 ) > gpurun_out/r04_q_c5_sliced_step.log 2>&1
grep "c5-sliced" gpurun_out/r04_q_c5_sliced_step.log
