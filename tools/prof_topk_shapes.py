#!/usr/bin/env python3
"""score_topk (kd = 64, k = 50, 8 masked items per query) at the evaluation shapes of the BASELINE configs:
fp16 filter + exact refinement vs the materialised fp32 path (use_filter=False), and agreement of the two.
    python tools/prof_topk_shapes.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import hip_ops  # noqa: E402

SHAPES = [("baby", 19445, 7050), ("baby eval batch", 4096, 7050), ("sports", 35598, 18357),
          ("clothing", 39387, 23033), ("c5 block", 20000, 500000)]


def main():
    global SHAPES
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":   # small problems: where does the filter start to pay?
        SHAPES = [("", nq, nc) for nc in (2048, 7050, 18357, 50000, 200000) for nq in (512, 2048, 4096, 8192, 16384)]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for name, nq, nc in SHAPES:
        Q = (torch.rand(nq, 64, device=dev, generator=g) - 0.5) * 0.2
        C = (torch.rand(nc, 64, device=dev, generator=g) - 0.5) * 0.2 + 0.05    # a common component, as after propagation
        rows = torch.arange(nq, device=dev).repeat_interleave(8)
        cols = torch.randint(0, nc, (nq * 8,), device=dev, generator=g)
        key = torch.unique(rows * nc + cols)
        rows, cols = key // nc, key % nc
        rp = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
        rp[1:] = torch.cumsum(torch.bincount(rows, minlength=nq), 0)
        rp, cl = rp.to(torch.int32), cols.to(torch.int32)
        res = {}
        for mode in ("1", "0"):
            for _ in range(2):
                out = hip_ops.score_topk(Q, C, 50, rp, cl, return_values=True, use_filter=mode == "1")
            torch.cuda.synchronize()
            reps = 3 if nc > 100000 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                out = hip_ops.score_topk(Q, C, 50, rp, cl, return_values=True, use_filter=mode == "1")
            torch.cuda.synchronize()
            res[mode] = ((time.perf_counter() - t0) / reps * 1e3, out)
        same = (res["1"][1][0] == res["0"][1][0]).float().mean().item()
        dv = (res["1"][1][1] - res["0"][1][1]).abs().max().item()
        print("%-16s %6d x %6d : filter %8.3f ms (%5.1f useful TF, %6.2f M users/s) | materialised %8.3f ms | same ids %.5f, "
              "max |dval| %.1e" % (name, nq, nc, res["1"][0], 2.0 * nq * nc * 64 / res["1"][0] / 1e9,
                                    nq / res["1"][0] / 1e3, res["0"][0], same, dv), flush=True)
        del Q, C, out, res
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
