#!/bin/bash
# round 4, run r: k <= 128 on the fp32 block path (select_topk_kernel's second half); where the wide-row fp16 pass beats the fp32 block path
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "topk" ) > gpurun_out/r04_r_topk_tests.log 2>&1
tail -4 gpurun_out/r04_r_topk_tests.log
( time timeout 900 python -m pytest tests/test_topk_fuzz_gpu.py -x -q ) > gpurun_out/r04_r_fuzz.log 2>&1
tail -4 gpurun_out/r04_r_fuzz.log
timeout 300 python - > gpurun_out/r04_r_grcn_eval_ab.log 2>&1 <<'PY'
import time, torch, numpy as np
from mmrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(19445, 7050, 192, 50), (35598, 18357, 192, 50), (19445, 7050, 192, 100), (4096, 3000, 64, 100)]
for kd in (192, 384, 1024, 4096):
    for n in (7050, 18357, 65536):
        if kd * n <= 4096 * 18357:
            shapes.append((n, n, kd, 10))            # kNN-shaped: the items against themselves
    shapes.append((19445, 7050, kd, 20))             # evaluation-shaped
for (nq, nc, kd, k) in shapes:
    Q = torch.randn(nq, kd, device=dev, generator=g) * 0.1 + 0.05
    C = torch.randn(nc, kd, device=dev, generator=g) * 0.1 + 0.05
    rows = torch.randint(0, nq, (8 * nq,), device=dev, generator=g)
    cols = torch.randint(0, nc, (8 * nq,), device=dev, generator=g)
    key = torch.unique(rows * nc + cols)
    rp, col = hip_ops.mask_to_csr(torch.stack((key // nc, key % nc)), nq, dev)
    def t(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.time() - t0) / reps * 1e3
    a = hip_ops.score_topk(Q, C, k, rp, col)
    b = hip_ops.score_topk(Q, C, k, rp, col, use_filter=False)
    same = (a == b).float().mean().item()
    print("%6d x %6d kd %d k %d: wide fp16 pass %.3f ms | fp32 block path %.3f ms | same ids %.5f" % (
        nq, nc, kd, k, t(lambda: hip_ops.score_topk(Q, C, k, rp, col)), t(lambda: hip_ops.score_topk(Q, C, k, rp, col, use_filter=False)), same), flush=True)
PY
cat gpurun_out/r04_r_grcn_eval_ab.log
