#!/usr/bin/env python3
"""A/B of LDS staging of the most popular X rows in the CSR SpMM (tools/spmm_lds_hot_lab.hip), config-5 graph, d = 64.

    python tools/spmm_lds_hot_lab.py build        # here (no GPU): tools/probe_libs/libspmm_lds_hot_lab.so
    python tools/spmm_lds_hot_lab.py run          # on the GPU box -> stdout (kept as profiles/r05_spmm_lds_hot_lab.log)

Reports, per H in {0, 128, 256, 512}: share of the short-row gathers served from LDS, ms per launch (HIP events, median of 5
windows x 10 launches), bitwise equality with H = 0, and the library's own launch on the same graph for scale."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probe_libs")
LIB = os.path.join(OUT, "libspmm_lds_hot_lab.so")


def build():
    from mmrec_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB,
                           os.path.join(ROOT, "tools", "spmm_lds_hot_lab.hip")])
    print("built", LIB)


def run():
    import torch
    from mmrec_amd import hip_ops, synth
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(LIB)
    nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = hip_ops.CsrGraph.from_coo_device(torch.from_numpy(r.astype(np.int32)).to(dev), torch.from_numpy(c.astype(np.int32)).to(dev),
                                         torch.from_numpy(v).to(dev), n, n, symmetric=True)
    rp = g.rowptr_host.astype(np.int64)
    long_t = g.long_row_threshold
    deg = np.diff(rp)
    short = deg <= long_t
    nnz_short = int(deg[short].sum())
    col = g.colidx.cpu().numpy()
    in_short = np.repeat(short, deg)
    counts = np.bincount(col[in_short], minlength=n)
    order = np.argsort(-counts, kind="stable")
    print("graph: %d rows, nnz %d; short rows (<= %d nonzeros): %d with %d nonzeros (%.1f %% of all)" %
          (n, rp[-1], long_t, int(short.sum()), nnz_short, 100.0 * nnz_short / rp[-1]))
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand(n, 64, device=dev, generator=gen) - 0.5
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def timed(fn, reps=10, windows=5):
        for _ in range(3):
            fn()
        per = []
        for _ in range(windows):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            per.append(a.elapsed_time(b) / reps)
        return float(np.median(per)), float(min(per)), float(max(per))

    Yl = torch.empty_like(X)
    med, lo, hi = timed(lambda: hip_ops.spmm_raw(g, X, Y=Yl))
    print("library launch (rows + chunk blocks + long-row reduce, all %d nonzeros): %.3f ms (min %.3f max %.3f)" % (rp[-1], med, lo, hi))
    base = None
    for H, grid in ((0, 512), (128, 512), (256, 512), (512, 256), (0, 256)):
        hot = order[:max(H, 1)].astype(np.int32)
        enc = col.copy()
        share = 0.0
        if H:
            slot = np.full(n, -1, np.int64)
            slot[hot[:H]] = np.arange(H)
            hit = slot[col] >= 0
            enc = np.where(hit, -(slot[col] + 1), col).astype(np.int32)
            share = float((hit & in_short).sum()) / nnz_short
        enc_d, hot_d = torch.from_numpy(enc).to(dev), torch.from_numpy(hot).to(dev)
        Y = torch.zeros_like(X)

        def launch():
            rc = lib.lab_spmm_rows(H, P(g.rowptr), P(enc_d), P(g.vals), P(X), P(hot_d), P(Y), n, long_t, grid, stream)
            assert rc == 0, rc
        med, lo, hi = timed(launch)
        torch.cuda.synchronize()
        same = ""
        if H == 0 and base is None:
            base = Y.clone()
            sel = torch.from_numpy(np.nonzero(short)[0]).to(dev)
            same = "  == library rows bitwise: %s" % bool(torch.equal(Y[sel], Yl[sel]))
        else:
            same = "  == H=0 bitwise: %s" % bool(torch.equal(Y, base))
        print("lab H = %3d (LDS %3d KB, grid %d): %.3f ms (min %.3f max %.3f)  gathers from LDS: %.1f %% of the short-row gathers%s" %
              (H, H * 256 // 1024, grid, med, lo, hi, 100 * share, same))


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
