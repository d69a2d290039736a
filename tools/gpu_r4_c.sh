#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 600 python tools/spmm_rows_lab.py run 20 ) > gpurun_out/r04_spmm_rows_lab2.log 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 -L ) > gpurun_out/r04_counters_list.txt 2>&1
grep -v "^\[c\|Warn\|warn" gpurun_out/r04_spmm_rows_lab2.log | tail -8; wc -l gpurun_out/r04_counters_list.txt
