"""What the feature-sliced multi-GPU layout costs, measured on ONE GPU (round-3 review, item 2).

A rank of a P-GPU feature-sliced run owns 64 / P columns of every table and the WHOLE graph; its propagation layer is
mmrec_spmm_csr_f32 on [N, 64 / P] slices and nothing crosses xGMI.  So the time of ONE slice's layer on one MI355X IS the
P-GPU per-layer time of the layout (all ranks run the same launch on their own columns) and
    implied speed-up at P = t(d = 64) / t(d = 64 / P).
Graphs: the config-5 headline graph (20M nnz, 1.5M rows), its 80 %-pruned training graph (4M nnz) and the item-item kNN
graph (10M nnz over 500K rows).    python tools/dslice_probe.py [--reps 20]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import hip_ops, synth  # noqa: E402


def time_layer(g, n_x, d, reps):
    dev = g.rowptr.device
    X = torch.randn(n_x, d, device=dev) * 0.1
    Y = torch.empty(g.n_rows, d, device=dev)
    for _ in range(3):
        hip_ops.spmm_raw(g, X, Y=Y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hip_ops.spmm_raw(g, X, Y=Y)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    graphs = {"c5_full_20M": (hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True), n)}
    rng = np.random.default_rng(0)
    keep = np.sort(rng.choice(eu.shape[0], eu.shape[0] // 5, replace=False))           # dropout 0.8 (freedom.py:128-143)
    r2, c2, v2 = synth.sym_norm_coo(eu[keep], ei[keep], nu, ni)
    graphs["c5_pruned_4M"] = (hip_ops.CsrGraph.from_coo_host(np.stack([r2, c2]), v2, n, n, dev, symmetric=True), n)
    rows = np.repeat(np.arange(ni), 20)                                                # image + text kNN(10), uncoalesced
    cols = rng.integers(0, ni, rows.shape[0])
    graphs["c5_item_item_10M"] = (hip_ops.CsrGraph.from_coo_host(np.stack([rows, cols]),
                                                                 np.full(rows.shape[0], 0.05, np.float32), ni, ni, dev), ni)
    # the same row structure as the full graph, but every column id folded into a 65,536-row window: the X slice a launch
    # gathers from is 2 MB at d = 8 -- L2 resident.  What a slice launch would take if its gathers hit L2 (the ceiling of
    # any scheme that makes them, e.g. all rows walking the column space in step).
    c_fold = (c % 65536).astype(c.dtype)
    graphs["c5_full_20M_cols_folded_to_64K"] = (hip_ops.CsrGraph.from_coo_host(np.stack([r, c_fold]), v, n, n, dev), n)
    out = {}
    for name, (g, n_x) in graphs.items():
        t = {d: time_layer(g, n_x, d, args.reps) for d in (64, 32, 16, 8)}
        alg = {d: ((8 + 4 * d) * g.nnz + (4 + 4 * d) * g.n_rows) / 1e9 for d in t}
        out[name] = {"nnz": g.nnz, "rows": g.n_rows, "ms_per_layer": {str(d): round(x, 4) for d, x in t.items()},
                     "implied_speedup": {str(64 // d): round(t[64] / t[d], 2) for d in (32, 16, 8)},
                     "algorithmic_GB": {str(d): round(x, 3) for d, x in alg.items()},
                     "algorithmic_TBps": {str(d): round(alg[d] / t[d], 2) for d in t}}
        print(name, json.dumps(out[name]), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
