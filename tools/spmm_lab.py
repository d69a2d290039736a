#!/usr/bin/env python3
"""SpMM cache-policy lab on the c5 graph: the library's kernel and variant builds of the same source
(-DMMREC_SPMM_LAB=mask: bit0 nontemporal colidx/vals loads, bit1 nontemporal Y stores, bit2 nontemporal X gathers)
on the whole graph, on the user rows only (gathers from the 128 MB item table) and on the item rows only (gathers
from the 256 MB user table).

    python tools/spmm_lab.py build          # here (hipcc cross-compiles): tools/probe_libs/libspmm_lab<mask>.so
    python tools/spmm_lab.py run [reps]     # on the GPU
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probe_libs")
MASKS = (0, 1, 2, 3, 4, 7)


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "mmrec_amd", "csrc", "spmm.hip")
    for m in MASKS:
        lib = os.path.join(OUT, "libspmm_lab%d.so" % m)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
               "-DMMREC_SPMM_LAB=%d" % m, src, "-o", lib]
        subprocess.run(cmd, check=True)
        print("built", lib)


def run(reps):
    import numpy as np
    import torch
    from mmrec_amd import hip_ops, synth
    dev = torch.device("cuda:0")
    nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    blocks = {"all": g, "users": g.row_block(0, nu), "items": g.row_block(nu, n)}
    x = torch.rand(n, 64, device=dev) - 0.5
    y = torch.empty_like(x)
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    print("graph: nnz %d, rows %d; long rows %d, chunks %d" % (g.nnz, n, g.n_long, g.n_chunks))
    for m in MASKS:
        lib = ctypes.CDLL(os.path.join(OUT, "libspmm_lab%d.so" % m))
        fn = lib.mmrec_spmm_csr_f32
        fn.restype = ctypes.c_int32
        fn.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_float, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        line = []
        for name, b in blocks.items():
            def call():
                rc = fn(P(b.rowptr), P(b.colidx), P(b.vals), P(x), P(y), None, None, None, b.n_rows, 64, 1.0, 0.0, 1.0,
                        b.long_row_threshold, P(b.long_rows), P(b.long_chunk_ptr), b.n_long, b.n_chunks,
                        P(b.partials_for(64)), None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            for _ in range(3):
                call()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                call()
            e.record()
            torch.cuda.synchronize()
            line.append("%s %.3f ms" % (name, s.elapsed_time(e) / reps))
        print("lab mask %d: %s" % (m, " | ".join(line)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
