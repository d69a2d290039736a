#!/usr/bin/env python3
"""The dense fused Adam (adam_multi_kernel) over config 5's id tables -- [1,000,000, 64] + [500,000, 64] fp32, 28 bytes moved
per element -- as a hipGraph replay: ms per step and TB/s.   python tools/prof_adam_dense.py   (MMREC_HIP_LIB picks a variant
library: tools/prof_adam_dense.py ab libA.so libB.so)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    import torch
    from mmrec_amd.common.optim import HipAdam
    dev = torch.device("cuda:0")
    ps = [torch.nn.Parameter(torch.randn(n, 64, device=dev) * 0.01) for n in (1_000_000, 500_000)]
    for p in ps:
        p.grad = torch.randn_like(p) * 1e-3
    opt = HipAdam(ps, lr=1e-3, capturable=True)
    opt.init_state()
    for _ in range(3):
        opt.step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    per = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(50):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        per.append(a.elapsed_time(b) / 50)
    ms = float(np.median(per))
    n = sum(p.numel() for p in ps)
    print("%-28s dense Adam of %d elements: median %.3f ms  min %.3f  max %.3f  -> %.2f TB/s of 28 B per element" %
          (os.path.basename(os.environ.get("MMREC_HIP_LIB", "libmmrec_hip.so")), n, ms, min(per), max(per), 28.0 * n / ms / 1e9))


if __name__ == "__main__":
    if sys.argv[1:2] == ["ab"]:
        for rnd in range(3):
            for lib in sys.argv[2:]:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, MMREC_HIP_LIB=os.path.abspath(lib)),
                                   capture_output=True, text=True)
                print(r.stdout.strip() or r.stderr[-300:], flush=True)
    else:
        one()
