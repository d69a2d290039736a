#!/bin/bash
# round 4, run u: listed-rows SpMM (pull / push, ABI 10) + FREEDOM's item-item layer at the batch rows only
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "listed_rows or pulled_item_rows or shared_user_bpr or sharded_freedom_plugin" ) > gpurun_out/r04_u_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^E  " gpurun_out/r04_u_tests.log | head -20
( time timeout 900 python tools/c5_sliced_step.py 40 ) > gpurun_out/r04_u_c5_step.log 2>&1
grep "c5-sliced" gpurun_out/r04_u_c5_step.log
