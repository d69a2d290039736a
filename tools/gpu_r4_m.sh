#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "feature_slices or on_feature_slices" ) > gpurun_out/r04_m_slices.log 2>&1
( time timeout 900 python tools/dslice_probe.py --out gpurun_out/r04_dslice_probe4.json ) > gpurun_out/r04_dslice_probe4.log 2>&1
tail -4 gpurun_out/r04_m_slices.log; grep -v "^$\|amdgpu.ids" gpurun_out/r04_dslice_probe4.log | tail -8
