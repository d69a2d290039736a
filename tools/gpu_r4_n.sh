#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "feature_slices" ) > gpurun_out/r04_n_slices.log 2>&1
for mb in 2 1 0.5 4; do
( timeout 600 python tools/dslice_probe.py --window-mb $mb ) 2>&1 | grep "c5_full_20M \|c5_pruned" | sed "s/^/[window $mb MB] /" >> gpurun_out/r04_dslice_probe5.log
done
tail -3 gpurun_out/r04_n_slices.log; cut -c1-420 gpurun_out/r04_dslice_probe5.log
