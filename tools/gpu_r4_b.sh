#!/bin/bash
# round 4, GPU call B: short-row SpMM variants, hip_deterministic A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 600 python tools/spmm_rows_lab.py run 20 ) > gpurun_out/r04_spmm_rows_lab.log 2>&1
( time timeout 900 python tools/det_ab.py c2 c3 c4 c5 ) > gpurun_out/r04_det_ab.log 2>&1
grep -v "^\[c\|Warn\|warn" gpurun_out/r04_spmm_rows_lab.log | tail -8; grep "det-ab\|Error\|error" gpurun_out/r04_det_ab.log | tail
