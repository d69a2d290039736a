#!/usr/bin/env python3
"""Within-run A/B of the long-row threshold in bench.py's own harness (3-layer chain, ping-pong buffers; interleaved
A/B/A/B rounds so that clock / box drift cancels): C5 (3 x spmm_raw) and Baby (lightgcn_mean, 3 layers)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import hip_ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for shape, thrs, steps in (("c5", (64, 32, 24), 20), ("baby", (64, 32, 16), 200), ("sports", (64, 32, 16), 200)):
        nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        n = nu + ni
        graphs = {t: hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True, long_row_threshold=t)
                  for t in thrs}
        X0 = torch.rand(n, 64, device=dev) - 0.5
        bufs = [torch.empty_like(X0), torch.empty_like(X0)]

        def step(g):
            if shape == "c5":
                cur = X0
                for layer in range(3):
                    hip_ops.spmm_raw(g, cur, Y=bufs[layer % 2])
                    cur = bufs[layer % 2]
            else:
                with torch.no_grad():
                    hip_ops.lightgcn_mean(g, X0, 3)
        res = {t: [] for t in thrs}
        for rnd in range(4):
            for t in thrs:
                for _ in range(3):
                    step(graphs[t])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step(graphs[t])
                torch.cuda.synchronize()
                res[t].append((time.perf_counter() - t0) / steps / 3 * 1e6)
        for t in thrs:
            print("%s thr %d: us/layer per round %s" % (shape, t, " ".join("%.1f" % x for x in res[t])), flush=True)


if __name__ == "__main__":
    main()
