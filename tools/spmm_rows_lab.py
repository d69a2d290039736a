#!/usr/bin/env python3
"""SpMM short-row lab (round-3 review item 4: "the SpMM the model actually runs"): variant builds of spmm.hip --
rows-per-group (RPG), rows walked together (RB) and X rows prefetched per row (PF) -- on the config-5 graphs the FREEDOM
step launches: the full graph (20M nnz / 1.5M rows: evaluation, headline), its 80 %-pruned training graph (4M nnz: 2.7 per
row) and the item-item kNN graph (10M nnz / 500K rows), plus Amazon-Baby.  Every variant's output is compared with the
library's, bit for bit.

    python tools/spmm_rows_lab.py build          # here (hipcc cross-compiles): tools/probe_libs/libspmm_rows_*.so
    python tools/spmm_rows_lab.py run [reps]     # on the GPU
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probe_libs")
# (name, rpg for > 2^18 rows, RB, PF)
VARIANTS = [("rpg1", 1, 0, 4), ("rpg2", 2, 0, 4), ("base_rpg4", 4, 0, 4), ("rpg8", 8, 0, 4),
            ("rb2_pf4", 2, 2, 4), ("rb2_pf8", 2, 2, 8), ("rb4_pf2", 4, 4, 2), ("rb4_pf4", 4, 4, 4), ("rb4_pf8", 4, 4, 8),
            ("rb8_pf2", 8, 8, 2), ("rb8_pf4", 8, 8, 4), ("waves8", 4, 0, 4), ("waves8_rpg2", 2, 0, 4)]
EXTRA = {"waves8": ["-DMMREC_SPMM_WAVES=8"], "waves8_rpg2": ["-DMMREC_SPMM_WAVES=8"]}


def build():
    os.makedirs(OUT, exist_ok=True)
    src = [os.path.join(ROOT, "mmrec_amd", "csrc", f) for f in ("spmm.hip", "spmm_narrow.hip")]
    procs = []
    for name, rpg, rb, pf in VARIANTS:
        lib = os.path.join(OUT, "libspmm_rows_%s.so" % name)
        procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                       "-munsafe-fp-atomics", "-DMMREC_SPMM_RPG(n)=((n) <= (1 << 18) ? 1 : %d)" % rpg,
                                       "-DMMREC_SPMM_RB=%d" % rb, "-DMMREC_SPMM_PF=%d" % pf] + EXTRA.get(name, []) + src + ["-o", lib]))
    assert all(p.wait() == 0 for p in procs)
    print("built", len(procs))


def graphs(dev):
    import numpy as np
    from mmrec_amd import hip_ops, synth
    out = {}
    nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
    n = nu + ni
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    out["c5_full_20M"] = (hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True), n)
    rng = np.random.default_rng(0)
    keep = np.sort(rng.choice(eu.shape[0], eu.shape[0] // 5, replace=False))
    r2, c2, v2 = synth.sym_norm_coo(eu[keep], ei[keep], nu, ni)
    out["c5_pruned_4M"] = (hip_ops.CsrGraph.from_coo_host(np.stack([r2, c2]), v2, n, n, dev, symmetric=True), n)
    rows = np.repeat(np.arange(ni), 20)
    cols = rng.integers(0, ni, rows.shape[0])
    out["c5_item_item_10M"] = (hip_ops.CsrGraph.from_coo_host(np.stack([rows, cols]), np.full(rows.shape[0], 0.05, np.float32),
                                                              ni, ni, dev), ni)
    return out


def run(reps):
    import torch
    dev = torch.device("cuda:0")
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    gs = graphs(dev)
    from mmrec_amd import hip_ops
    for gname, (g, n_x) in gs.items():
        x = torch.rand(n_x, 64, device=dev) - 0.5
        ref = torch.empty(g.n_rows, 64, device=dev)
        hip_ops.spmm_raw(g, x, Y=ref)
        line = []
        for name, rpg, rb, pf in VARIANTS:
            lib = ctypes.CDLL(os.path.join(OUT, "libspmm_rows_%s.so" % name))
            fn = lib.mmrec_spmm_csr_f32
            fn.restype = ctypes.c_int32
            fn.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            y = torch.full((g.n_rows, 64), float("nan"), device=dev)

            def call():
                rc = fn(P(g.rowptr), P(g.colidx), P(g.vals), P(x), P(y), None, None, None, g.n_rows, 64, 1.0, 0.0, 1.0,
                        g.long_row_threshold, P(g.long_rows), P(g.long_chunk_ptr), g.n_long, g.n_chunks,
                        P(g.partials_for(64)), None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            for _ in range(3):
                call()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                call()
            e.record()
            torch.cuda.synchronize()
            same = bool(torch.equal(y, ref))
            line.append("%s %.4f ms%s" % (name, s.elapsed_time(e) / reps, "" if same else " (BITS DIFFER)"))
        print("%s (nnz %d, rows %d, %.1f per row; long rows %d): %s" % (gname, g.nnz, g.n_rows, g.nnz / g.n_rows, g.n_long,
                                                                          " | ".join(line)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
