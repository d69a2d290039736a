#!/bin/bash
# round 4, GPU call A: the new parity tests + the feature-slice probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export MMREC_TEST_OBSERVED=$PWD/gpurun_out/r04_observed_a.tsv
rm -f $MMREC_TEST_OBSERVED
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "feature_slices or on_feature_slices or padding_steps or rows_of_128" ) > gpurun_out/r04_a_slices.log 2>&1
( time timeout 900 python tools/dslice_probe.py --out gpurun_out/r04_dslice_probe.json ) > gpurun_out/r04_dslice_probe.log 2>&1
( time timeout 900 python -m pytest tests/test_topk_fuzz_gpu.py -q -m gpu ) > gpurun_out/r04_a_fuzz.log 2>&1
( time timeout 900 python -m pytest tests/test_c5_pieces_gpu.py -q -m gpu -s ) > gpurun_out/r04_a_pieces.log 2>&1
tail -3 gpurun_out/r04_a_slices.log; cat gpurun_out/r04_dslice_probe.log | tail -8; tail -3 gpurun_out/r04_a_fuzz.log; tail -5 gpurun_out/r04_a_pieces.log
