#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "linear" ) > gpurun_out/r04_k_linear.log 2>&1
( time timeout 600 python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
from mmrec_amd import hip_ops
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev).manual_seed(0)
for n in (7050, 18357, 23033, 500000):
    X = torch.rand(n, 4096, device=dev, generator=gen)
    W = torch.rand(64, 4096, device=dev, generator=gen) - 0.5
    b = torch.zeros(64, device=dev)
    res = []
    for split in (False, True):
        hip_ops.LINEAR_F16X3 = split
        for _ in range(10):
            hip_ops.linear(X, W, b)
        torch.cuda.synchronize()
        reps = 100 if n < 100000 else 10
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            hip_ops.linear(X, W, b)
        e.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(e) / reps * 1e3)
    fl = 2.0 * n * 4096 * 64
    print("[linear-ab] n %d x 4096: fp32 MFMA %.1f us (%.1f TF) | split fp16 x3 %.1f us (%.1f TF useful, X stream %.2f TB/s)" %
          (n, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, 4.0 * n * 4096 / res[1] / 1e6), flush=True)
    del X
PY
) > gpurun_out/r04_linear_split_ab.log 2>&1
tail -4 gpurun_out/r04_k_linear.log; grep "linear-ab\|Error" gpurun_out/r04_linear_split_ab.log
