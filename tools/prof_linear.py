#!/usr/bin/env python3
"""Times the modal projection (mmrec_linear_*: Y = X W^T + b, out = 64) over the item counts of the
named configurations; prints TFLOP/s against the 157.3 TFLOP/s fp32-MFMA peak and GB/s of X."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import hip_ops  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    shapes = [("baby", 7050, 4096), ("sports", 18357, 4096), ("clothing", 23033, 4096),
              ("baby-text", 7050, 384), ("clothing-text", 23033, 384), ("vbpr-baby", 7050, 4480),
              ("c5-shard", 62500, 4096), ("c5", 500000, 4096)]
    shapes += [(a, int(a[1:]), 4096) for a in sys.argv[1:] if a[0] == "n" and a[1:].isdigit()]     # n12000: 12,000 rows x 4096
    for name, n, f in shapes:
        if sys.argv[1:] and name not in sys.argv[1:]:        # python tools/prof_linear.py baby c5 : only these shapes
            continue
        X = torch.rand(n, f, device=dev, generator=gen)
        W = (torch.rand(64, f, device=dev, generator=gen) - 0.5).requires_grad_()
        b = torch.zeros(64, device=dev, requires_grad=True)
        G = torch.rand(n, 64, device=dev, generator=gen) - 0.5
        reps = 50 if n < 100000 else 10
        with torch.no_grad():
            t_f = timed(lambda: hip_ops.linear(X, W, b), reps)
        Xg = X.requires_grad_()

        def fb():
            Xg.grad = None
            hip_ops.linear(Xg, W, b).backward(G)
        t_fb = timed(fb, reps)
        hip_ops.LINEAR_F16X3 = False                       # A/B: the fp32-MFMA kernels (forward, dW, dX)
        with torch.no_grad():
            t_f32 = timed(lambda: hip_ops.linear(X, W, b), reps)
        t_fb32 = timed(fb, reps)
        hip_ops.LINEAR_F16X3 = True
        fl = 2.0 * n * f * 64
        print("%-14s n=%7d F=%4d  fwd %8.1f us  %6.1f TF/s (%4.1f%% of 157.3)  X %5.2f TB/s | fwd+bwd %8.1f us  %6.1f TF/s"
              " || fp32-MFMA kernels: fwd %8.1f us, fwd+bwd %8.1f us"
              % (name, n, f, t_f * 1e6, fl / t_f / 1e12, fl / t_f / 157.3e10, n * f * 4 / t_f / 1e12,
                 t_fb * 1e6, 3 * fl / t_fb / 1e12, t_f32 * 1e6, t_fb32 * 1e6), flush=True)
        del X, W, b, G, Xg


VARIANTS = {"nt0": "-DMMREC_BWD_NT=0", "nt1_dw_reads": "-DMMREC_BWD_NT=1", "nt2_dx_stores": "-DMMREC_BWD_NT=2",
            "nt3_both": "-DMMREC_BWD_NT=3"}          # builds of gemm.hip with extra flags


def build_variants(specs=()):
    """python tools/prof_linear.py build-variants [name=-DFLAG[,-DFLAG] ...]   (here, no GPU): tools/probe_libs/libmmrec_bwd_<name>.so
    (default: the MMREC_BWD_NT set above)"""
    variants = dict(sp.split("=", 1) for sp in specs) if specs else VARIANTS
    import subprocess
    from mmrec_amd import build as b
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_libs")
    os.makedirs(out, exist_ok=True)
    b.build(verbose=False)
    objs = [os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.SOURCES if s != "gemm.hip"]
    for name, flags in variants.items():
        o = os.path.join(out, "gemm_%s.o" % name)
        subprocess.check_call([b._hipcc()] + b.FLAGS + flags.split(",") + ["-c", os.path.join(b.CSRC, "gemm.hip"), "-o", o])
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(out, "libmmrec_bwd_%s.so" % name)] + objs + [o])
        os.remove(o)
        print("built", name, flush=True)


def run_variants(shapes):
    """python tools/prof_linear.py run-variants [shapes...]   (GPU): every variant library in its own process, twice"""
    import subprocess
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_libs")
    names = os.environ["MMREC_VARIANTS"].split(",") if os.environ.get("MMREC_VARIANTS") else list(VARIANTS)   # (built by build-variants name=flags)
    for rnd in range(2):
        for name in names:
            env = dict(os.environ, MMREC_HIP_LIB=os.path.join(out, "libmmrec_bwd_%s.so" % name))
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + shapes, env=env, capture_output=True, text=True)
            for line in r.stdout.strip().splitlines():
                print("%-14s %s" % (name, line), flush=True)
            if r.returncode:
                print(name, "FAILED", r.stderr[-400:])


if __name__ == "__main__":
    if sys.argv[1:2] == ["build-variants"]:
        build_variants(sys.argv[2:])
    elif sys.argv[1:2] == ["run-variants"]:
        run_variants(sys.argv[2:])
    else:
        main()
