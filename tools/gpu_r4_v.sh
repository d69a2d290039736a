#!/bin/bash
# round 4, run v: the propagated tables at the batch rows only (lightgcn_mean_parts_rows + spmm_rows): tests + the config-5 step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "listed_rows or batch_rows or pulled_item_rows or shared_user_bpr or sharded_freedom_plugin" ) > gpurun_out/r04_v_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^E  " gpurun_out/r04_v_tests.log | head -20
( time timeout 900 python tools/c5_sliced_step.py 40 ) > gpurun_out/r04_v_c5_step.log 2>&1
grep "c5-sliced" gpurun_out/r04_v_c5_step.log
( time timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_c5_e2e_gpu.py tests/test_config_shapes_gpu.py -q -x -k "freedom or FREEDOM or c5" ) > gpurun_out/r04_v_freedom_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^E  " gpurun_out/r04_v_freedom_tests.log | head -20
