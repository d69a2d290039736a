#!/usr/bin/env python3
"""Number of candidate ranges of the fp16 top-K filter (grid.y of the pass kernels; 32 group maxima per range and query):
fewer ranges = fewer, longer workgroups (one resident wave of workgroups at Baby size) but a looser bound (more survivors).
    python tools/prof_topk_ranges.py build ; python tools/prof_topk_ranges.py run"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "probe_libs")
VARIANTS = [(4, 4), (6, 6), (8, 8), (4, 8), (6, 12), (8, 16), (12, 16), (16, 16)]


def build():
    sys.path.insert(0, ROOT)
    from mmrec_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.SOURCES if s != "topk_filter.hip"]
    b.build(verbose=False)
    for lo, hi in VARIANTS:
        o = os.path.join(OUT, "tfr_%d_%d.o" % (lo, hi))
        subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DMMREC_TF_MINR=%d" % lo, "-DMMREC_TF_MAXR=%d" % hi, "-c",
                                                        os.path.join(b.CSRC, "topk_filter.hip"), "-o", o])
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libmmrec_ranges_%d_%d.so" % (lo, hi))] + objs + [o])
        os.remove(o)
    print("built", len(VARIANTS))


def run_one():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from mmrec_amd import hip_ops, synth
    dev = torch.device("cuda:0")
    out = []
    for shape in ("baby", "sports", "clothing"):
        nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, nu + ni, nu + ni, dev, symmetric=True)
        E0 = torch.empty(nu + ni, 64, device=dev)
        torch.nn.init.xavier_uniform_(E0[:nu]), torch.nn.init.xavier_uniform_(E0[nu:])
        E = hip_ops.lightgcn_mean(g, E0, 2)                      # what a model ranks with
        U, I = E[:nu].contiguous(), E[nu:].contiguous()
        rp, col = hip_ops.mask_to_csr(np.stack([eu, ei]), nu, dev)
        for _ in range(3):
            hip_ops.score_topk(U, I, 50, rp, col)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            hip_ops.score_topk(U, I, 50, rp, col)
        torch.cuda.synchronize()
        out.append("%s %.4f ms" % (shape, (time.perf_counter() - t0) / 20 * 1e3))
    print(" | ".join(out))


def run():
    for rnd in range(2):
        for lo, hi in VARIANTS:
            env = dict(os.environ, MMREC_HIP_LIB=os.path.join(OUT, "libmmrec_ranges_%d_%d.so" % (lo, hi)))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
            print("ranges %2d..%2d  %s" % (lo, hi, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]),
                  flush=True)


if __name__ == "__main__":
    {"build": build, "run": run, "one": run_one}[sys.argv[1]]()
