#!/usr/bin/env python3
"""Forward + backward of the 4096 -> 64 projection as a hipGraph replay (what bench.py reports as baby_linear4096_fwd_bwd_us), for
A/B runs of whole-library variants in ONE lease:

    python tools/prof_linear_replay.py [n=7050] [F=4096]                 # the library in the tree (or MMREC_HIP_LIB)
    python tools/prof_linear_replay.py ab libA.so libB.so ... [n] [F]    # alternating, three rounds, each in its own process
    python tools/prof_linear_replay.py fork [n] [F]                      # hip_ops.LINEAR_BWD_FORK off / on, alternating
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(n, F):
    import numpy as np
    import torch
    from mmrec_amd import hip_ops
    if os.environ.get("MMREC_LINEAR_BWD_FORK") is not None and hasattr(hip_ops, "LINEAR_BWD_FORK"):
        # (A/B of a two-stream backward that was built and removed in round 5: profiles/r05_linear_bwd_fork_ab.log)
        hip_ops.LINEAR_BWD_FORK = os.environ["MMREC_LINEAR_BWD_FORK"] == "1"
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand(n, F, device=dev, generator=gen).requires_grad_()
    W = (torch.rand(64, F, device=dev, generator=gen) - 0.5).requires_grad_()
    b = torch.zeros(64, device=dev, requires_grad=True)
    G = torch.rand(n, 64, device=dev, generator=gen) - 0.5

    def fb():
        X.grad = W.grad = b.grad = None
        hip_ops.linear(X, W, b).backward(G)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5):
            fb()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            fb()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(10):
        graph.replay()
    per = []
    for _ in range(7):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(200):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        per.append(a.elapsed_time(e) / 200 * 1e3)
    print("n=%d F=%d fwd+bwd replay: median %.2f us  min %.2f  max %.2f" % (n, F, float(np.median(per)), min(per), max(per)), flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["ab"]:
        libs = [a for a in sys.argv[2:] if a.endswith(".so")]
        rest = [a for a in sys.argv[2:] if not a.endswith(".so")]
        for rnd in range(3):
            for lib in libs:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + rest, env=dict(os.environ, MMREC_HIP_LIB=os.path.abspath(lib)),
                                   capture_output=True, text=True)
                print("%-40s %s" % (os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:]), flush=True)
    elif sys.argv[1:2] == ["fork"]:
        for rnd in range(3):
            for flag in ("0", "1"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[2:], env=dict(os.environ, MMREC_LINEAR_BWD_FORK=flag),
                                   capture_output=True, text=True)
                print("LINEAR_BWD_FORK=%s  %s" % (flag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:]), flush=True)
    else:
        one(int(sys.argv[1]) if len(sys.argv) > 1 else 7050, int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
