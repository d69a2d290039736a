#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 1200 python tools/pmc_kernels.py spmm gpurun_out/r04_spmm_pmc ) > gpurun_out/r04_spmm_pmc.log 2>&1
tail -5 gpurun_out/r04_spmm_pmc.log
