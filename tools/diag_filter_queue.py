#!/usr/bin/env python3
"""How many queries of a config-5-size evaluation block the fp16 filter hands to its slow queue, by kind of embeddings."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import _lib, hip_ops, synth  # noqa: E402


def al256(x):
    return (x + 255) & ~255


def main():
    dev = torch.device("cuda:0")
    nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    E0 = torch.empty(n, 64, device=dev)
    torch.nn.init.xavier_uniform_(E0[:nu]), torch.nn.init.xavier_uniform_(E0[nu:])
    x3 = E0
    for _ in range(3):
        y = torch.empty_like(E0)
        hip_ops.spmm_raw(g, x3, Y=y)
        x3 = y
    kinds = {"xavier (untrained ego)": E0, "layer-3 output": x3, "LightGCN mean of 3 layers": hip_ops.lightgcn_mean(g, E0, 3),
             "uniform random": torch.rand(n, 64, device=dev) - 0.5}
    lib = _lib.load()
    nq, k = 20000, 50
    s, e = np.searchsorted(eu, 0), np.searchsorted(eu, nq)
    rp, col = hip_ops.mask_to_csr(np.stack([eu[s:e], ei[s:e]]), nq, dev)
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    for name, E in kinds.items():
        Q, C = E[:nq].contiguous(), E[nu:].contiguous()
        ws = torch.zeros(lib.mmrec_topk_workspace_bytes(nq, ni, 64, k), dtype=torch.uint8, device=dev)
        idx = torch.empty(nq, k, dtype=torch.int64, device=dev)
        n_stages = -(-ni // 64)
        nq_pad = -(-nq // 256) * 256
        off = al256(nq_pad * 128) + al256(n_stages * 64 * 128) + al256(nq_pad * 4)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(lib.mmrec_score_topk_f32(P(Q), P(C), nq, ni, 64, P(rp), P(col), k, P(idx), None, P(ws), 0,
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "topk")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        flagged = int(ws[off:off + 512].view(torch.int32)[67])
        print("%-28s: %5d of %d queries in the slow queue, %.2f ms per 20,000-user block" % (name, flagged, nq, dt * 1e3), flush=True)


if __name__ == "__main__":
    main()
