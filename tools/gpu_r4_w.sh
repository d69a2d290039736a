#!/bin/bash
# round 4, run w: hip_pull_batch_rows on / off at the small configurations (FREEDOM at Amazon-Sports and Amazon-Baby shape)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for c in c3 freedom_baby; do
  for f in "" "--no-batch-rows"; do
    echo "== $c $f"
    timeout 600 python tools/run_config.py $c --epochs 3 $f 2>&1 | grep "epoch"
  done
done > gpurun_out/r04_w_batch_rows_small_ab.log 2>&1
cat gpurun_out/r04_w_batch_rows_small_ab.log
