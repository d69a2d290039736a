#!/usr/bin/env python3
"""A/B of build-time variants of the fp16 top-K filter (topk_filter.hip) as whole libraries (MMREC_HIP_LIB selects one).

    python tools/prof_topk_variants.py build      # here (no GPU): variant libraries into tools/probe_libs/
    python tools/prof_topk_variants.py run        # on the GPU: parity of every variant vs the materialised fp32 path + ms per call

Round 3 used this harness for the pass-1 stage stride
(profiles/r03_topk_pass1_stride_ab.log) and for the one-workgroup-per-CU pass kernel that was built, measured and dropped
(profiles/r03_topk_wide_kernel_ab.log; the kernel itself is in git history: commit "top-K filter: v_max3 group maxima ...")."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "probe_libs")
VARIANTS = {          # name -> defines
    "current": [],                                # the tree as it is (pass 1 on every second stage from 131,072 candidates on)
    "final16": ["-DMMREC_TF_FINAL_ROWS=16"],      # final kernel: 16 instead of 8 candidate rows in flight per lane (116 instead of 68 VGPRs)
    "final12": ["-DMMREC_TF_FINAL_ROWS=12"],
    "final4": ["-DMMREC_TF_FINAL_ROWS=4"],        # fewer rows in flight, fewer registers, more resident waves
}


def build():
    sys.path.insert(0, ROOT)
    from mmrec_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    b.build(verbose=False)
    objs = [os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.SOURCES if s != "topk_filter.hip"]
    extra = b.EXTRA_FLAGS.get("topk_filter.hip", [])
    jobs = [(name, os.path.join(b.CSRC, "topk_filter.hip"), defs + extra) for name, defs in VARIANTS.items()]
    # (the round-2 source no longer links against the current topk.hip: ABI 7 added entry points it does not define)
    for name, src, defs in jobs:
        o = os.path.join(OUT, "tfw_%s.o" % name)
        subprocess.check_call([b._hipcc()] + b.FLAGS + defs + ["-c", src, "-o", o])
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libmmrec_tfw_%s.so" % name)] + objs + [o])
        os.remove(o)
        print("built", name, flush=True)


def cases(dev):
    import numpy as np
    import torch
    from mmrec_amd import hip_ops, synth
    out = []
    for shape in ("baby", "sports", "clothing"):
        nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, nu + ni, nu + ni, dev, symmetric=True)
        E0 = torch.empty(nu + ni, 64, device=dev)
        torch.nn.init.xavier_uniform_(E0[:nu]), torch.nn.init.xavier_uniform_(E0[nu:])
        E = hip_ops.lightgcn_mean(g, E0, 2)                      # what a model ranks with
        rp, col = hip_ops.mask_to_csr(np.stack([eu, ei]), nu, dev)
        out.append((shape, E[:nu].contiguous(), E[nu:].contiguous(), rp, col))
    if os.environ.get("TFW_C5", "1") != "0":                     # what bench.py's c5_full_eval ranks: E = A^3 X0 at config-5 size
        nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        g = hip_ops.CsrGraph.from_coo_device(torch.from_numpy(r.astype(np.int32)).to(dev), torch.from_numpy(c.astype(np.int32)).to(dev),
                                             torch.from_numpy(v).to(dev), nu + ni, nu + ni, symmetric=True)
        E = torch.rand(nu + ni, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) - 0.5
        for _ in range(3):
            E = hip_ops.spmm(g, E)
        nq = 65536
        sel = eu < nq                                            # edges are sorted by user
        rp, col = hip_ops.mask_to_csr(np.stack([eu[sel], ei[sel]]), nq, dev)
        out.append(("c5prop_65536x500000", E[:nq].contiguous(), E[nu:].contiguous(), rp, col))
        del g
    gen = torch.Generator(device=dev).manual_seed(9)
    for nq, nc, kd in ((65536, 500000, 64), (4096, 7050, 64), (20000, 40000, 64), (333, 33000, 64),
                       (65536, 500000, 128), (19445, 7050, 128)):
        common = torch.randn(kd, device=dev, generator=gen) * 0.05
        Q = torch.randn(nq, kd, device=dev, generator=gen) * 0.03 + common
        C = torch.randn(nc, kd, device=dev, generator=gen) * 0.03 + common
        mrow = np.repeat(np.arange(nq), 8)
        mcol = np.random.default_rng(2).integers(0, nc, nq * 8)
        key = np.unique(mrow.astype(np.int64) * nc + mcol)
        rp, col = hip_ops.mask_to_csr(np.stack([key // nc, key % nc]), nq, dev)
        out.append(("%dx%dx%d" % (nq, nc, kd), Q, C, rp, col))
    only = os.environ.get("TFW_ONLY")                            # e.g. TFW_ONLY=c5prop under rocprofv3
    return [c for c in out if only is None or c[0].startswith(only)]


def run_one():
    import torch
    sys.path.insert(0, ROOT)
    from mmrec_amd import hip_ops
    dev = torch.device("cuda:0")
    res = []
    for name, Q, C, rp, col in cases(dev):
        idx, val = hip_ops.score_topk(Q, C, 50, rp, col, return_values=True)
        # parity: the materialised fp32 path on a sample of queries (ids up to near-ties: compare the score VALUES)
        n = min(Q.shape[0], 4096)
        sel = torch.randperm(Q.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(1))[:n].sort()[0]
        rps = rp.long()
        lens = (rps[sel + 1] - rps[sel])
        srp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        srp[1:] = torch.cumsum(lens, 0).to(torch.int32)
        src = torch.repeat_interleave(rps[sel], lens) + (torch.arange(int(lens.sum()), device=dev) -
                                                          torch.repeat_interleave(srp[:-1].long(), lens))
        scol = col[src].contiguous() if src.numel() else col[:1]
        ridx, rval = hip_ops.score_topk(Q[sel].contiguous(), C, 50, srp, scol, return_values=True, use_filter=False)
        same_ids = float((idx[sel] == ridx).all(dim=1).float().mean())
        unit = float(rval.abs().max())
        dv = float((val[sel] - rval).abs().max()) / max(unit, 1e-30)
        for _ in range(2):
            hip_ops.score_topk(Q, C, 50, rp, col)
        torch.cuda.synchronize()
        reps = 5 if Q.shape[0] * C.shape[0] > 1e9 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            hip_ops.score_topk(Q, C, 50, rp, col)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        res.append("%s %.3f ms (ids %.4f, dval %.1e)" % (name, ms, same_ids, dv))
    print(" | ".join(res))


def run():
    names = list(VARIANTS)
    for rnd in range(2):
        for name in names:
            env = dict(os.environ, MMREC_HIP_LIB=os.path.join(OUT, "libmmrec_tfw_%s.so" % name))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
            print("%-14s %s" % (name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED: " + r.stderr[-600:]),
                  flush=True)


if __name__ == "__main__":
    {"build": build, "run": run, "one": run_one}[sys.argv[1]]()
