#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "relabelled" ) > gpurun_out/r04_g_relabel.log 2>&1
( time timeout 900 python tools/spmm_locality_probe.py ) > gpurun_out/r04_locality_probe.log 2>&1
tail -3 gpurun_out/r04_g_relabel.log; grep -v "^$\|amdgpu.ids" gpurun_out/r04_locality_probe.log | tail -12
