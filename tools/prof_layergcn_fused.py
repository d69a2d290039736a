#!/usr/bin/env python3
"""LayerGCN forward (4 layers, Baby-shaped graph): the fused one-launch-per-layer epilogue vs SpMM + cos-scale kernels."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import _lib, hip_ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for shape in ("baby", "sports", "c5"):
        nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        n = nu + ni
        g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
        E0 = (torch.rand(n, 64, device=dev) - 0.5) * 0.1
        lib = _lib.load()
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def unfused():
            acc = torch.zeros_like(E0)
            cur = E0
            for _ in range(4):
                y = torch.empty_like(E0)
                hip_ops.spmm_raw(g, cur, Y=y)
                out, w = torch.empty_like(E0), torch.empty(n, device=dev)
                _lib.check(lib.mmrec_cos_scale_fwd_f32(P(y), P(E0), P(out), P(w), P(acc), n, 64, S()), "cos")
                cur = out
            return acc

        def fused():
            with torch.no_grad():
                return hip_ops.layergcn_sum(g, E0, 4)
        a, b = unfused(), fused()
        assert torch.equal(a, b), "fused != unfused"
        for name, fn in (("unfused", unfused), ("fused", fused)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50 if shape != "c5" else 10):
                fn()
            torch.cuda.synchronize()
            print("%s %s: %.1f us per 4-layer forward (bit-identical results)" %
                  (shape, name, (time.perf_counter() - t0) / (50 if shape != "c5" else 10) * 1e6), flush=True)


if __name__ == "__main__":
    main()
