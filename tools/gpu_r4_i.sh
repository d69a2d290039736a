#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "feature_slices or on_feature_slices or relabelled" ) > gpurun_out/r04_i_slices.log 2>&1
( time timeout 900 python tools/dslice_probe.py --out gpurun_out/r04_dslice_probe3.json ) > gpurun_out/r04_dslice_probe3.log 2>&1
( time timeout 600 python -m pytest tests/test_topk_fuzz_gpu.py -q -m gpu ) > gpurun_out/r04_i_fuzz.log 2>&1
tail -4 gpurun_out/r04_i_slices.log; grep -v "^$\|amdgpu.ids" gpurun_out/r04_dslice_probe3.log | tail -8; tail -4 gpurun_out/r04_i_fuzz.log
