#!/bin/bash
# round 4, run s: top-K tests after the wide-row crossover rule + C5 kNN pieces (wide path still serves them)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -k "topk" ) > gpurun_out/r04_s_topk_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_s_topk_tests.log
( time timeout 900 python -m pytest tests/test_c5_pieces_gpu.py -q -k "knn" ) > gpurun_out/r04_s_c5_knn.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_s_c5_knn.log
