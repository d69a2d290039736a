#!/usr/bin/env python3
"""SpMM layer time on the cache-resident dataset shapes vs the long-row threshold of the plan (rows above it go to chunk blocks)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import hip_ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for shape in ("baby", "sports", "clothing"):
        nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        n = nu + ni
        x = torch.rand(n, 64, device=dev) - 0.5
        y = torch.empty_like(x)
        for thr in (8, 16, 32, 64, 128, 512, 1 << 30):
            g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True, long_row_threshold=thr)
            for _ in range(20):
                hip_ops.spmm_raw(g, x, Y=y)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(200):
                hip_ops.spmm_raw(g, x, Y=y)
            e.record()
            torch.cuda.synchronize()
            print("%s nnz %d thr %d: %.2f us/layer (long rows %d, chunks %d)" % (shape, g.nnz, thr, s.elapsed_time(e) / 200 * 1e3,
                                                                          g.n_long, g.n_chunks), flush=True)


if __name__ == "__main__":
    main()
