#!/usr/bin/env python3
"""SpMM plan sweep: long-row threshold x chunk size x rows per group on the Baby / Sports / C5 graphs, variant builds of
spmm.hip (their own plan functions), timed with HIP events.
    python tools/spmm_sweep.py build ; python tools/spmm_sweep.py run"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probe_libs")
VARIANTS = [(chunk, rpg) for chunk in (256, 512, 1024) for rpg in (1, 2, 4)]


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "mmrec_amd", "csrc", "spmm.hip")
    procs = []
    for chunk, rpg in VARIANTS:
        lib = os.path.join(OUT, "libspmm_c%d_r%d.so" % (chunk, rpg))
        procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
                                       "-DMMREC_SPMM_CHUNK=%d" % chunk, "-DMMREC_SPMM_RPG(n)=%d" % rpg, src, "-o", lib]))
    assert all(p.wait() == 0 for p in procs)
    print("built", len(procs))


def run():
    import numpy as np
    import torch
    from mmrec_amd import hip_ops, synth
    dev = torch.device("cuda:0")
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    HP = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for shape in ("baby", "sports", "c5"):
        nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        n = nu + ni
        g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
        rp = g.rowptr_host
        x = torch.rand(n, 64, device=dev) - 0.5
        y = torch.empty_like(x)
        reps = 20 if shape == "c5" else 200
        for chunk, rpg in VARIANTS:
            lib = ctypes.CDLL(os.path.join(OUT, "libspmm_c%d_r%d.so" % (chunk, rpg)))
            fn = lib.mmrec_spmm_csr_f32
            fn.restype = ctypes.c_int32
            fn.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            out = []
            for thr in (16, 32, 64):
                nl, nc = ctypes.c_int32(0), ctypes.c_int32(0)
                lib.mmrec_spmm_plan_count(HP(rp), n, thr, ctypes.byref(nl), ctypes.byref(nc))
                lr, cp = np.empty(max(nl.value, 1), np.int32), np.empty(nl.value + 1, np.int32)
                lib.mmrec_spmm_plan_fill(HP(rp), n, thr, HP(lr), HP(cp))
                lrd, cpd = torch.from_numpy(lr).to(dev), torch.from_numpy(cp).to(dev)
                part = torch.empty(max(nc.value, 1) * 64, device=dev)

                def call(a, b):
                    rc = fn(P(g.rowptr), P(g.colidx), P(g.vals), P(a), P(b), None, None, None, n, 64, 1.0, 0.0, 1.0, thr,
                            P(lrd), P(cpd), nl.value, nc.value, P(part), None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                    assert rc == 0, rc
                for _ in range(5):
                    call(x, y)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    call(x, y)
                e.record()
                torch.cuda.synchronize()
                out.append("thr %d: %.1f us" % (thr, s.elapsed_time(e) / reps * 1e3))
            print("%s chunk %d rows/group %d | %s" % (shape, chunk, rpg, " | ".join(out)), flush=True)


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
