#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 600 python tools/spmm_rows_lab.py run 30 ) > gpurun_out/r04_spmm_rows_lab3.log 2>&1
grep -v "^\[c\|Warn\|warn\|amdgpu" gpurun_out/r04_spmm_rows_lab3.log | tail -6
