#!/bin/bash
# round 4, run t: the rescue GEMM as a persistent 1-D grid over live tiles (gemm_nt_live_kernel): parity + the C5 kNN build time
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -k "topk or knn or linear or gemm" ) > gpurun_out/r04_t_tests.log 2>&1
grep -n "passed\|failed\|^FAILED" gpurun_out/r04_t_tests.log
( time MMREC_TEST_OBSERVED=gpurun_out/r04_observed_t.tsv timeout 900 python -m pytest tests/test_c5_pieces_gpu.py -q -s -k "knn" ) > gpurun_out/r04_t_c5_knn.log 2>&1
grep -n "passed\|failed\|kNN\|knn" gpurun_out/r04_t_c5_knn.log | head -20
