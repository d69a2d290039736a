#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export MMREC_TEST_OBSERVED=$PWD/gpurun_out/r04_observed_f.tsv
rm -f $MMREC_TEST_OBSERVED
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "wide_rows or general_kd or knn_shape or sharded_freedom_plugin_rccl" ) > gpurun_out/r04_f_wide.log 2>&1
( time timeout 900 python -m pytest tests/test_config_shapes_gpu.py -q -x -m gpu -k "knn_graph" ) > gpurun_out/r04_f_knn_sports.log 2>&1
( time timeout 900 python -m pytest tests/test_c5_pieces_gpu.py -q -m gpu -s -k knn ) > gpurun_out/r04_f_pieces.log 2>&1
( time timeout 900 python -m pytest tests/test_topk_fuzz_gpu.py -q -m gpu ) > gpurun_out/r04_f_fuzz.log 2>&1
tail -4 gpurun_out/r04_f_wide.log; tail -4 gpurun_out/r04_f_knn_sports.log; grep "c5-pieces\|passed\|failed" gpurun_out/r04_f_pieces.log; tail -4 gpurun_out/r04_f_fuzz.log
